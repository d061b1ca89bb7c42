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

// ---- composite probe: exactly the memory work of the dense hash-join probe
// (hash_join.rs:207-253 on a direct-address table) and nothing else: stream 8-B probe keys, ONE random 4-B
// load per key into a table of `slots` entries (3.8 MiB for 1e6 build keys: resident in one XCD's L2),
// stream the (left_idx u64, right_idx u32) pair out at the key's own position.  No ballot, no look-back,
// no barrier, no compaction: every key hits.  MODE selects how the table is read:
//   0 plain load (allocates a 128-B line in the CU's L1 per miss), 1 non-temporal, 2 sc1 (agent scope: L1 bypassed)
template<int ILP, int MODE>
__global__ void k_probe_composite(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ heads, uint64_t mask,
                                  uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x*ILP+threadIdx.x; const size_t s=(size_t)gridDim.x*blockDim.x*ILP;
  for(; i + (size_t)(ILP-1)*blockDim.x < n; i += s){
    uint64_t k[ILP]; uint32_t h[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++) k[j]=__builtin_nontemporal_load(keys+i+(size_t)j*blockDim.x);
    #pragma unroll
    for(int j=0;j<ILP;j++){
      const uint32_t* p=heads+(k[j]&mask);
      h[j] = MODE==0 ? *p : (MODE==1 ? __builtin_nontemporal_load(p) : __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT));
    }
    #pragma unroll
    for(int j=0;j<ILP;j++){
      __builtin_nontemporal_store((uint64_t)h[j], left+i+(size_t)j*blockDim.x);
      __builtin_nontemporal_store((uint32_t)(i+(size_t)j*blockDim.x), right+i+(size_t)j*blockDim.x);
    }
  }
}
// the lookups alone (indices computed, nothing streamed): what the three load flavours sustain into a 4-B table
template<int ILP, int MODE>
__global__ void k_rand4(const uint32_t* __restrict__ tab, uint64_t mask, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0;
  for(; i + (ILP-1)*s < n; i += ILP*s){
    uint32_t v[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++){
      const uint32_t* p=tab+(mix64(i+j*s)&mask);
      v[j] = MODE==0 ? *p : (MODE==1 ? __builtin_nontemporal_load(p) : __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT));
    }
    #pragma unroll
    for(int j=0;j<ILP;j++) acc+=v[j];
  }
  if(acc==0x1234567) out[0]=acc;
}
__global__ void k_fill_keys(uint64_t* keys, uint64_t mask, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) keys[i]=mix64(i*0x9E3779B97F4A7C15ull+1)&mask;
}

template<class F> float timeit(F f, int reps=5){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float best=1e30f;
  for(int r=0;r<reps;r++){ hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
  return best;
}
int probe_composite(){
  const size_t n=100000000; // C3: 1e8 probe keys against 1e6 build keys
  uint64_t *keys,*left; uint32_t *right,*heads; uint64_t* out;
  CK(hipMalloc(&keys,8*n)); CK(hipMalloc(&left,8*n)); CK(hipMalloc(&right,4*n)); CK(hipMalloc(&out,4096));
  printf("composite probe: %zu keys (8 B in) -> one random 4-B table load each -> (u64,u32) pair out, all hit; algorithmic bytes 8 nP + 12 M = %.1f GB\n", n, 20.0*n/1e9);
  for(size_t slots: {(size_t)1<<17,(size_t)1<<20,(size_t)1<<21,(size_t)1<<23}){ // 0.5 / 4 / 8 / 32 MiB tables
    CK(hipMalloc(&heads,4*slots)); CK(hipMemset(heads,1,4*slots));
    k_fill_keys<<<4096,256>>>(keys,slots-1,n); CK(hipDeviceSynchronize());
    const char* mname[3]={"plain","nt","sc1"};
    float best=1e30f;
    #define RUN(ILP,MODE,G) { float ms=timeit([&]{k_probe_composite<ILP,MODE><<<G,256>>>(keys,heads,slots-1,left,right,n);}); \
      printf("  table %5.1f MiB  ilp %2d  %-5s grid %5d: %.3f ms  %.0f Gkeys/s  %.0f GB/s algorithmic (%.3f of 8 TB/s)\n", 4.0*slots/1048576, ILP, mname[MODE], G, ms, n/ms/1e6, 20.0*n/ms/1e6, 20.0*n/ms/1e6/8000); if(ms<best)best=ms; }
    RUN(4,0,8192) RUN(8,0,4096) RUN(8,0,8192) RUN(16,0,2048) RUN(16,0,4096)
    RUN(8,1,4096) RUN(16,1,4096) RUN(8,2,4096) RUN(16,2,4096) RUN(16,2,2048)
    #undef RUN
    for(int mode=0;mode<3;mode++){
      float ms = mode==0 ? timeit([&]{k_rand4<8,0><<<4096,256>>>(heads,slots-1,out,n);}) :
                 mode==1 ? timeit([&]{k_rand4<8,1><<<4096,256>>>(heads,slots-1,out,n);}) :
                           timeit([&]{k_rand4<8,2><<<4096,256>>>(heads,slots-1,out,n);});
      printf("  table %5.1f MiB  lookups alone (%s): %.3f ms  %.0f G/s\n", 4.0*slots/1048576, mname[mode], ms, n/ms/1e6);
    }
    printf("  table %5.1f MiB  best composite %.3f ms = %.3f of the HBM roofline on 8 nP + 12 M bytes\n", 4.0*slots/1048576, best, 20.0*n/best/1e6/8000);
    CK(hipFree(heads));
  }
  { // the streams alone: keys in, pairs out, no lookup
    float ms=timeit([&]{k_probe_composite<8,0><<<4096,256>>>(keys,(const uint32_t*)left,0,left,right,n);});
    printf("  streams alone (every key -> slot 0): %.3f ms  %.0f GB/s\n", ms, 20.0*n/ms/1e6);
  }
  return 0;
}


typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
// ---- width sweep: does the streaming rate depend on the bytes a lane loads per instruction? (mode 'w')
template<class T, int NT> __global__ void k_read_w(const T* __restrict__ a, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0;
  for(; i + 3*s < n; i += 4*s){
    T v0 = NT ? __builtin_nontemporal_load(a+i) : a[i], v1 = NT ? __builtin_nontemporal_load(a+i+s) : a[i+s];
    T v2 = NT ? __builtin_nontemporal_load(a+i+2*s) : a[i+2*s], v3 = NT ? __builtin_nontemporal_load(a+i+3*s) : a[i+3*s];
    acc += (uint64_t)v0 ^ (uint64_t)v1 ^ (uint64_t)v2 ^ (uint64_t)v3;
  }
  if(acc==0x1234567) out[0]=acc;
}
// one workgroup streams ONE contiguous region (the bucket pass's shape): rows of 8 B + 4 B in two columns, or 16 B in one
template<int MODE> __global__ __launch_bounds__(512) void k_region(const uint64_t* __restrict__ v, const uint32_t* __restrict__ w, const ulonglong2* __restrict__ r16,
                                                              uint64_t* out, size_t rows_per_wg){
  const size_t lo = blockIdx.x * rows_per_wg, hi = lo + rows_per_wg;
  uint64_t acc = 0;
  if(MODE!=2) for(size_t i = lo + threadIdx.x; i + 3*512 < hi; i += 4*512){
    if(MODE==0){ // 8 + 4 bytes per row, two loads
      #pragma unroll
      for(int u=0;u<4;u++){ acc += __builtin_nontemporal_load(v+i+u*512) ^ __builtin_nontemporal_load(w+i+u*512); }
    } else if(MODE==1){ // 16-byte rows, one load
      #pragma unroll
      for(int u=0;u<4;u++){ u64x2_t t = __builtin_nontemporal_load((const u64x2_t*)r16+i+u*512); acc += t.x ^ t.y; }
    }
  }
  if(MODE==2){ // 8 + 4 bytes per row, a lane takes TWO consecutive rows: one 16-byte and one 8-byte load per pair
    for(size_t q = threadIdx.x; q + 512 < rows_per_wg/2; q += 2*512){
      #pragma unroll
      for(int u=0;u<2;u++){ const size_t p = lo + 2*(q + u*512);
        u64x2_t t = __builtin_nontemporal_load((const u64x2_t*)(v+p)); uint64_t ww = __builtin_nontemporal_load((const uint64_t*)(w+p)); acc += t.x ^ t.y ^ ww; }
    }
  }
  if(acc==0x1234567) out[0]=acc;
}
int width_sweep(){
  const size_t GB=1ull<<30; char *A; uint64_t* out; CK(hipMalloc(&A,10*GB)); CK(hipMemset(A,1,10*GB)); CK(hipMalloc(&out,4096));
  for(int nt=0;nt<2;nt++) for(int g: {2048,8192}){
    float m4 = nt? timeit([&]{k_read_w<uint32_t,1><<<g,256>>>((uint32_t*)A,out,2*GB/4);}) : timeit([&]{k_read_w<uint32_t,0><<<g,256>>>((uint32_t*)A,out,2*GB/4);});
    float m8 = nt? timeit([&]{k_read_w<uint64_t,1><<<g,256>>>((uint64_t*)A,out,2*GB/8);}) : timeit([&]{k_read_w<uint64_t,0><<<g,256>>>((uint64_t*)A,out,2*GB/8);});
    printf("stream read 2 GiB %s grid %d: 4 B/lane %.3f ms %.0f GB/s | 8 B/lane %.3f ms %.0f GB/s\n", nt?"nt":"plain", g, m4, 2.0*GB/m4/1e6, m8, 2.0*GB/m8/1e6);
  }
  // the bucket pass's shape: 2442 regions of 204800 rows, one 512-thread workgroup each
  const size_t rows = 204800, nwg = 2442, n = rows*nwg;
  float a = timeit([&]{k_region<0><<<nwg,512>>>((const uint64_t*)A,(const uint32_t*)(A+5*GB),nullptr,out,rows);});
  float b = timeit([&]{k_region<1><<<nwg,512>>>(nullptr,nullptr,(const ulonglong2*)A,out,rows);});
  float c = timeit([&]{k_region<2><<<nwg,512>>>((const uint64_t*)A,(const uint32_t*)(A+5*GB),nullptr,out,rows);});
  printf("regions (2442 x 204800 rows, 512 threads, 4 rows per trip): 8+4 B columns %.3f ms %.0f GB/s | 16 B rows %.3f ms %.0f GB/s | 8+4 B, two rows per lane %.3f ms %.0f GB/s\n",
         a, 12.0*n/a/1e6, b, 16.0*n/b/1e6, c, 12.0*n/c/1e6);
  return 0;
}

// the composite probe with the key stream taken off the VGPR return path: keys land in LDS by LDS-DMA (global_load_lds_dwordx4:
// 128 keys per wave-instruction, lane l gets keys 2l, 2l+1), each lane then looks its two keys up and writes the pairs with one
// 16-byte and one 8-byte store.  NG such loads in flight per wave.  (mode 'g')
template<int NG>
__global__ __launch_bounds__(256) void k_probe_glds(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ heads, uint64_t mask,
                                                    uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n){
  __shared__ __attribute__((aligned(16))) uint64_t sk[4][NG][128];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const size_t per = (size_t)NG * 128, stride = (size_t)gridDim.x * 4 * per;
  for(size_t c = ((size_t)blockIdx.x * 4 + w) * per; c + per <= n; c += stride){
    #pragma unroll
    for(int g=0; g<NG; g++)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(keys + c + (size_t)g*128 + 2*l),
                                       (void __attribute__((address_space(3)))*)&sk[w][g][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    asm volatile("" ::: "memory");
    uint64_t k[2*NG]; uint32_t h[2*NG];
    #pragma unroll
    for(int g=0; g<NG; g++){ k[2*g] = sk[w][g][2*l]; k[2*g+1] = sk[w][g][2*l+1]; }
    #pragma unroll
    for(int j=0;j<2*NG;j++) h[j] = heads[k[j]&mask];
    #pragma unroll
    for(int g=0; g<NG; g++){
      const size_t r = c + (size_t)g*128 + 2*l;
      u64x2_t lv; lv.x = h[2*g]; lv.y = h[2*g+1];
      __builtin_nontemporal_store(lv, (u64x2_t*)(left + r));
      __builtin_nontemporal_store(((uint64_t)(uint32_t)(r+1) << 32) | (uint32_t)r, (uint64_t*)(right + r));
    }
  }
}
int probe_glds(){
  const size_t n=100000000;
  uint64_t *keys,*left; uint32_t *right,*heads;
  CK(hipMalloc(&keys,8*n)); CK(hipMalloc(&left,8*n)); CK(hipMalloc(&right,4*n));
  for(size_t slots: {(size_t)1<<17,(size_t)1<<20}){
    CK(hipMalloc(&heads,4*slots)); CK(hipMemset(heads,1,4*slots));
    k_fill_keys<<<4096,256>>>(keys,slots-1,n); CK(hipDeviceSynchronize());
    #define RUNG(NG,G) { float ms=timeit([&]{k_probe_glds<NG><<<G,256>>>(keys,heads,slots-1,left,right,n);}); \
      printf("  table %5.1f MiB  LDS-DMA keys, %d x 128 keys in flight per wave, grid %5d: %.3f ms  %.3f of 8 TB/s on 20 B/key\n", 4.0*slots/1048576, NG, G, ms, 20.0*n/ms/1e6/8000); }
    RUNG(2,4096) RUNG(4,2048) RUNG(4,4096) RUNG(8,1024) RUNG(8,2048) RUNG(8,4096)
    float ms=timeit([&]{k_probe_composite<16,0><<<4096,256>>>(keys,heads,slots-1,left,right,n);});
    printf("  table %5.1f MiB  register composite ilp16: %.3f ms  %.3f\n", 4.0*slots/1048576, ms, 20.0*n/ms/1e6/8000);
    CK(hipFree(heads));
  }
  return 0;
}

int main(int argc, char** argv){
  if(argc>1 && argv[1][0]=='g') return probe_glds();
  if(argc>1 && argv[1][0]=='p') return probe_composite();
  if(argc>1 && argv[1][0]=='w') return width_sweep();
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
