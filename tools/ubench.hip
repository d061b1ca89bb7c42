// Microbenchmarks that inform the kernel design (DESIGN.md cites the numbers):
// streaming bandwidth, random 16-B table reads, random global atomics, LDS atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

__global__ void k_copy(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) b[i]=a[i];
}
__global__ void k_read(const ulonglong2* __restrict__ a, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0; for(;i<n;i+=s){ ulonglong2 v=a[i]; acc+=v.x^v.y; }
  if(acc==0x1234567) out[0]=acc;
}
// random 16B loads: n lookups, table of `slots` 16-B entries
template<int ILP>
__global__ void k_rand16(const ulonglong2* __restrict__ tab, uint64_t slots_mask, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0;
  for(; i + (ILP-1)*s < n; i += ILP*s){
    ulonglong2 v[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++) v[j]=tab[mix64(i+j*s)&slots_mask];
    #pragma unroll
    for(int j=0;j<ILP;j++) acc+=v[j].x^v[j].y;
  }
  if(acc==0x1234567) out[0]=acc;
}
template<int ILP>
__global__ void k_rand8(const uint64_t* __restrict__ tab, uint64_t slots_mask, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0;
  for(; i + (ILP-1)*s < n; i += ILP*s){
    uint64_t v[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++) v[j]=tab[mix64(i+j*s)&slots_mask];
    #pragma unroll
    for(int j=0;j<ILP;j++) acc+=v[j];
  }
  if(acc==0x1234567) out[0]=acc;
}
__global__ void k_atomic_f64(double* tab, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) unsafeAtomicAdd(&tab[mix64(i)&mask], 1.0);
}
__global__ void k_atomic_u64(unsigned long long* tab, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) atomicAdd(&tab[mix64(i)&mask], 1ULL);
}
__global__ void k_atomic_u32(unsigned* tab, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) atomicAdd(&tab[mix64(i)&mask], 1u);
}
__global__ void k_atomic_min(unsigned long long* tab, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) atomicMin(&tab[mix64(i)&mask], (unsigned long long)i);
}
__global__ void k_scatter8(uint64_t* tab, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) tab[mix64(i)&mask]=i;
}
// LDS atomics: each WG has 8192 doubles (64KB) and does iters random adds per thread
__global__ void k_lds_f64(double* out, int iters){
  __shared__ double t[8192];
  for(int i=threadIdx.x;i<8192;i+=blockDim.x) t[i]=0;
  __syncthreads();
  uint64_t x = blockIdx.x*1315423911u + threadIdx.x;
  for(int i=0;i<iters;i++){ x=mix64(x+i); unsafeAtomicAdd(&t[x&8191], 1.0); }
  __syncthreads();
  if(threadIdx.x==0) out[blockIdx.x]=t[5];
}
__global__ void k_lds_u64(unsigned long long* out, int iters){
  __shared__ unsigned long long t[8192];
  for(int i=threadIdx.x;i<8192;i+=blockDim.x) t[i]=0;
  __syncthreads();
  uint64_t x = blockIdx.x*1315423911u + threadIdx.x;
  for(int i=0;i<iters;i++){ x=mix64(x+i); atomicAdd(&t[x&8191], 1ULL); }
  __syncthreads();
  if(threadIdx.x==0) out[blockIdx.x]=t[5];
}
__global__ void k_lds_cas(unsigned long long* out, int iters){
  __shared__ unsigned long long t[8192];
  for(int i=threadIdx.x;i<8192;i+=blockDim.x) t[i]=~0ULL;
  __syncthreads();
  uint64_t x = blockIdx.x*1315423911u + threadIdx.x; unsigned long long acc=0;
  for(int i=0;i<iters;i++){ x=mix64(x+i); acc+=atomicCAS(&t[x&8191], ~0ULL, x); }
  __syncthreads();
  if(threadIdx.x==0) out[blockIdx.x]=t[5]+acc;
}

template<class F> float timeit(F f, int reps=5){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float best=1e30f;
  for(int r=0;r<reps;r++){ hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
  return best;
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d L2=%d MB clock=%d MHz lds/block=%zu\n", p.name,p.multiProcessorCount,p.l2CacheSize>>20,p.clockRate/1000,p.sharedMemPerBlock);
  const size_t GB=1ull<<30;
  char *A,*B; CK(hipMalloc(&A,2*GB)); CK(hipMalloc(&B,2*GB)); CK(hipMemset(A,1,2*GB)); CK(hipMemset(B,0,2*GB));
  uint64_t* out; CK(hipMalloc(&out,1<<20));
  int grids[]={2048,4096,8192};
  for(int g:grids){
    size_t n=GB/16; float ms=timeit([&]{k_copy<<<g,256>>>((ulonglong2*)A,(ulonglong2*)B,n);});
    printf("copy 1GiB grid=%d: %.3f ms  %.1f GB/s (r+w)\n",g,ms,2.0*GB/ms/1e6);
    n=2*GB/16; ms=timeit([&]{k_read<<<g,256>>>((ulonglong2*)A,out,n);});
    printf("read 2GiB grid=%d: %.3f ms  %.1f GB/s\n",g,ms,2.0*GB/ms/1e6);
  }
  size_t nl=1ull<<27; // 134M lookups
  for(size_t mb: {1,2,4,8,16,32,64,128,256,1024}){
    uint64_t slots=(mb<<20)/16;
    float m1=timeit([&]{k_rand16<1><<<8192,256>>>((ulonglong2*)A,slots-1,out,nl);});
    float m4=timeit([&]{k_rand16<4><<<8192,256>>>((ulonglong2*)A,slots-1,out,nl);});
    float m8=timeit([&]{k_rand16<8><<<4096,256>>>((ulonglong2*)A,slots-1,out,nl);});
    printf("rand16 table=%4zu MB: ilp1 %.3f ms %.1f G/s | ilp4 %.3f ms %.1f G/s | ilp8 %.3f ms %.1f G/s\n",mb,m1,nl/m1/1e6,m4,nl/m4/1e6,m8,nl/m8/1e6);
  }
  for(size_t mb: {4,8,16,32,64}){
    uint64_t slots=(mb<<20)/8;
    float m4=timeit([&]{k_rand8<4><<<8192,256>>>((uint64_t*)A,slots-1,out,nl);});
    printf("rand8  table=%4zu MB: ilp4 %.3f ms %.1f G/s\n",mb,m4,nl/m4/1e6);
  }
  for(size_t mb: {1,8,16,32,128}){
    uint64_t slots=(mb<<20)/8;
    float a=timeit([&]{k_atomic_f64<<<8192,256>>>((double*)B,slots-1,nl);});
    float b=timeit([&]{k_atomic_u64<<<8192,256>>>((unsigned long long*)B,slots-1,nl);});
    float c=timeit([&]{k_atomic_min<<<8192,256>>>((unsigned long long*)B,slots-1,nl);});
    float d=timeit([&]{k_scatter8<<<8192,256>>>((uint64_t*)B,slots-1,nl);});
    float e=timeit([&]{k_atomic_u32<<<8192,256>>>((unsigned*)B,slots*2-1,nl);});
    printf("atomics region=%4zu MB: addf64 %.3f ms %.2f G/s | addu64 %.3f ms %.2f G/s | minu64 %.3f ms %.2f G/s | addu32 %.3f ms %.2f G/s | scatter8 %.3f ms %.2f G/s\n",mb,a,nl/a/1e6,b,nl/b/1e6,c,nl/c/1e6,e,nl/e/1e6,d,nl/d/1e6);
  }
  {
    int iters=4096; int g=2048; double ops=(double)g*256*iters;
    float a=timeit([&]{k_lds_f64<<<g,256>>>((double*)out,iters);});
    float b=timeit([&]{k_lds_u64<<<g,256>>>((unsigned long long*)out,iters);});
    float c=timeit([&]{k_lds_cas<<<g,256>>>((unsigned long long*)out,iters);});
    printf("LDS atomics (64KB table/WG, 2 WG/CU): addf64 %.3f ms %.1f G/s | addu64 %.3f ms %.1f G/s | cas64 %.3f ms %.1f G/s\n",a,ops/a/1e6,b,ops/b/1e6,c,ops/c/1e6);
  }
  return 0;
}
