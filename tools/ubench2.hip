// Round-5 micro-benchmarks for the dense hash-join probe (C3: 1e8 probe keys x 1e6 build keys, every key hits).
// The r02 counters (profiles/r02_probe_pmc_ta.txt) say the probe's time is the SUM of the L1 miss latencies / (64
// outstanding misses per CU x 256 CUs).  Two levers follow and are measured here before anything is built into the product:
//   (a) a smaller direct-address table (3-byte or 20-bit entries instead of 4 bytes: 3.0 / 2.5 MiB instead of 4 MiB), so
//       that the table stays in its XCD's 4 MiB L2 next to the streams -> fewer lookups pay the beyond-L2 latency;
//   (b) the key stream pre-fetched into L2 through the SCALAR cache (s_load_dword per 128-byte line, a few us ahead):
//       the vector key loads then hold their L1 miss slot for an L2-hit latency, not an HBM latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));

// FMT 0: 4-byte entries; 1: 3-byte entries (unaligned 4-byte load, low 24 bits); 2: 20-bit entries (unaligned load, shift, mask)
template<int FMT> __device__ __forceinline__ uint32_t tab_load(const uint8_t* __restrict__ tab, uint64_t e){
  if(FMT==0) return ((const uint32_t*)tab)[e];
  if(FMT==1) return *(const u32u*)(tab + 3u*(uint32_t)e) & 0xFFFFFFu;
  const uint32_t e32 = (uint32_t)e; return (*(const u32u*)(tab + ((e32*5u)>>1)) >> ((e32&1u)<<2)) & 0xFFFFFu;
}
template<int ILP, int FMT>
__global__ __launch_bounds__(256) void k_comp(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab,
                                  uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; const size_t S=(size_t)gridDim.x*blockDim.x;
  for(; i + (size_t)(ILP-1)*S < n; i += ILP*S){
    uint64_t k[ILP]; uint32_t h[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++) k[j]=__builtin_nontemporal_load(keys+i+(size_t)j*S);
    #pragma unroll
    for(int j=0;j<ILP;j++) h[j]=tab_load<FMT>(tab,k[j]);
    #pragma unroll
    for(int j=0;j<ILP;j++){
      __builtin_nontemporal_store((uint64_t)h[j], left+i+(size_t)j*S);
      __builtin_nontemporal_store((uint32_t)(i+(size_t)j*S), right+i+(size_t)j*S);
    }
  }
}
// WHAT: 0 = keys + lookups + stores, 1 = keys + lookups (no stores), 2 = lookups + stores (keys computed), 3 = keys from a
// 1 MiB window (L2-hot) + lookups + stores
template<int ILP, int WHAT>
__global__ __launch_bounds__(256) void k_parts(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab, uint64_t slots,
                                  uint64_t* __restrict__ left, uint32_t* __restrict__ right, uint64_t* out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; const size_t S=(size_t)gridDim.x*blockDim.x;
  uint64_t acc=0;
  for(; i + (size_t)(ILP-1)*S < n; i += ILP*S){
    uint64_t k[ILP]; uint32_t h[ILP];
    #pragma unroll
    for(int j=0;j<ILP;j++){
      const size_t r=i+(size_t)j*S;
      k[j] = WHAT==2 ? (mix64(r)%slots) : WHAT==3 ? __builtin_nontemporal_load(keys+(r&((1u<<17)-1))) : __builtin_nontemporal_load(keys+r);
    }
    #pragma unroll
    for(int j=0;j<ILP;j++) h[j]=tab_load<0>(tab,k[j]);
    #pragma unroll
    for(int j=0;j<ILP;j++){
      if(WHAT==1){ acc+=h[j]; continue; }
      __builtin_nontemporal_store((uint64_t)h[j], left+i+(size_t)j*S);
      __builtin_nontemporal_store((uint32_t)(i+(size_t)j*S), right+i+(size_t)j*S);
    }
  }
  if(WHAT==1 && acc==0x1234567) out[0]=acc;
}

// wave-contiguous chunks of NL x 128 keys (lane l takes keys 2l, 2l+1 of each 128-key piece: one 16-byte load), the chunk
// D trips ahead pre-fetched into L2 by scalar loads, one per GRAN bytes.  D = 0: no prefetch (the baseline of this shape).
template<int NL, int D, int GRAN, int FMT>
__global__ __launch_bounds__(256) void k_pf(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab,
                                            uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n, uint32_t* sink){
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  constexpr size_t per = (size_t)NL * 128;
  const size_t nchunks = n / per, gw = (size_t)blockIdx.x * 4 + w, nw = (size_t)gridDim.x * 4;
  uint32_t d = 0;
  for(size_t c = gw; c < nchunks; c += nw){
    if(D > 0){
      const size_t pc = c + (size_t)D * nw;
      if(pc < nchunks){
        const uint64_t* pb = keys + pc * per;
        const uint64_t pbs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)pb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)pb);
        #pragma unroll
        for(int j = 0; j < (int)(per * 8 / GRAN); j++)
          asm volatile("s_load_dword %0, %1, %2" : "+s"(d) : "s"(pbs), "n"(j * GRAN));
      }
    }
    const size_t r0 = c * per + 2 * l;
    u64x2_t k[NL]; uint32_t h[2*NL];
    #pragma unroll
    for(int g=0; g<NL; g++) k[g] = __builtin_nontemporal_load((const u64x2_t*)(keys + r0 + (size_t)g*128));
    #pragma unroll
    for(int g=0; g<NL; g++){ h[2*g] = tab_load<FMT>(tab,k[g].x); h[2*g+1] = tab_load<FMT>(tab,k[g].y); }
    #pragma unroll
    for(int g=0; g<NL; g++){
      const size_t r = r0 + (size_t)g*128;
      u64x2_t lv; lv.x = h[2*g]; lv.y = h[2*g+1];
      __builtin_nontemporal_store(lv, (u64x2_t*)(left + r));
      __builtin_nontemporal_store(((uint64_t)(uint32_t)(r+1) << 32) | (uint32_t)r, (uint64_t*)(right + r));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if(d == 0x12345679u) sink[0] = d;
}
// a pure stream read with and without the scalar prefetch (does the prefetch cost / gain anything on its own?)
template<int NL, int D>
__global__ __launch_bounds__(256) void k_stream_pf(const uint64_t* __restrict__ keys, size_t n, uint64_t* sink){
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  constexpr size_t per = (size_t)NL * 128;
  const size_t nchunks = n / per, gw = (size_t)blockIdx.x * 4 + w, nw = (size_t)gridDim.x * 4;
  uint32_t d = 0; uint64_t acc = 0;
  for(size_t c = gw; c < nchunks; c += nw){
    if(D > 0){
      const size_t pc = c + (size_t)D * nw;
      if(pc < nchunks){
        const uint64_t* pb = keys + pc * per;
        const uint64_t pbs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)pb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)pb);
        #pragma unroll
        for(int j = 0; j < (int)(per * 8 / 128); j++)
          asm volatile("s_load_dword %0, %1, %2" : "+s"(d) : "s"(pbs), "n"(j * 128));
      }
    }
    const size_t r0 = c * per + 2 * l;
    #pragma unroll
    for(int g=0; g<NL; g++){ u64x2_t k = __builtin_nontemporal_load((const u64x2_t*)(keys + r0 + (size_t)g*128)); acc += k.x ^ k.y; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if(acc == 0x12345679u + d) sink[0] = acc;
}


// keys by LDS-DMA (global_load_lds_dwordx4: 128 keys per wave-instruction), NG pieces in flight per wave, packed table
template<int NG, int FMT>
__global__ __launch_bounds__(256) void k_glds(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab,
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
    for(int j=0;j<2*NG;j++) h[j] = tab_load<FMT>(tab,k[j]);
    #pragma unroll
    for(int g=0; g<NG; g++){
      const size_t r = c + (size_t)g*128 + 2*l;
      u64x2_t lv; lv.x = h[2*g]; lv.y = h[2*g+1];
      __builtin_nontemporal_store(lv, (u64x2_t*)(left + r));
      __builtin_nontemporal_store(((uint64_t)(uint32_t)(r+1) << 32) | (uint32_t)r, (uint64_t*)(right + r));
    }
  }
}
// chunk form with one key per lane and piece (8-byte loads, 64 keys per piece): does the 16-byte key load matter?
template<int NL, int FMT, int OCC>
__global__ __launch_bounds__(256, OCC) void k_chunk8(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab,
                                            uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n){
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  constexpr size_t per = (size_t)NL * 64;
  const size_t nchunks = n / per, gw = (size_t)blockIdx.x * 4 + w, nw = (size_t)gridDim.x * 4;
  for(size_t c = gw; c < nchunks; c += nw){
    const size_t r0 = c * per + l;
    uint64_t k[NL]; uint32_t h[NL];
    #pragma unroll
    for(int g=0; g<NL; g++) k[g] = __builtin_nontemporal_load(keys + r0 + (size_t)g*64);
    #pragma unroll
    for(int g=0; g<NL; g++) h[g] = tab_load<FMT>(tab,k[g]);
    #pragma unroll
    for(int g=0; g<NL; g++){
      const size_t r = r0 + (size_t)g*64;
      __builtin_nontemporal_store((uint64_t)h[g], left + r);
      __builtin_nontemporal_store((uint32_t)r, right + r);
    }
  }
}
// chunk form, 16-byte key loads, occupancy hint and plain (not nt) stores as options
template<int NL, int FMT, int OCC, int NTS>
__global__ __launch_bounds__(256, OCC) void k_chunk16(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tab,
                                            uint64_t* __restrict__ left, uint32_t* __restrict__ right, size_t n){
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  constexpr size_t per = (size_t)NL * 128;
  const size_t nchunks = n / per, gw = (size_t)blockIdx.x * 4 + w, nw = (size_t)gridDim.x * 4;
  for(size_t c = gw; c < nchunks; c += nw){
    const size_t r0 = c * per + 2 * l;
    u64x2_t k[NL]; uint32_t h[2*NL];
    #pragma unroll
    for(int g=0; g<NL; g++) k[g] = __builtin_nontemporal_load((const u64x2_t*)(keys + r0 + (size_t)g*128));
    #pragma unroll
    for(int g=0; g<NL; g++){ h[2*g] = tab_load<FMT>(tab,k[g].x); h[2*g+1] = tab_load<FMT>(tab,k[g].y); }
    #pragma unroll
    for(int g=0; g<NL; g++){
      const size_t r = r0 + (size_t)g*128;
      u64x2_t lv; lv.x = h[2*g]; lv.y = h[2*g+1];
      const uint64_t rv = ((uint64_t)(uint32_t)(r+1) << 32) | (uint32_t)r;
      if(NTS){ __builtin_nontemporal_store(lv, (u64x2_t*)(left + r)); __builtin_nontemporal_store(rv, (uint64_t*)(right + r)); }
      else { *(u64x2_t*)(left + r) = lv; *(uint64_t*)(right + r) = rv; }
    }
  }
}

__global__ void k_fill_keys(uint64_t* keys, uint64_t slots, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) keys[i]=mix64(i*0x9E3779B97F4A7C15ull+1)%slots;
}
template<class F> float timeit(F f, int reps=7){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  float best=1e30f;
  for(int r=0;r<reps;r++){ hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
  return best;
}
#define P(name, ms) printf("  %-64s %.3f ms  %.3f of 8 TB/s on 20 B/key\n", name, ms, 20.0*n/(ms)/1e6/8000)

int main(int argc, char** argv){
  const size_t n=100000000;
  uint64_t *keys,*left,*out; uint32_t *right; uint8_t* tab; uint32_t* sink;
  CK(hipMalloc(&keys,8*n+4096)); CK(hipMalloc(&left,8*n+4096)); CK(hipMalloc(&right,4*n+4096)); CK(hipMalloc(&out,4096)); CK(hipMalloc(&sink,4096));
  CK(hipMalloc(&tab,(64<<20)+64)); CK(hipMemset(tab,1,(64<<20)+64));
  const char which = argc>1 ? argv[1][0] : 'a';
  for(size_t slots: {(size_t)1000000, (size_t)250000, (size_t)2000000}){
    k_fill_keys<<<4096,256>>>(keys,slots,n); CK(hipDeviceSynchronize());
    printf("== %zu build keys: table 4 B %.2f MiB | 3 B %.2f MiB | 20 bit %.2f MiB\n", slots, 4.0*slots/1048576, 3.0*slots/1048576, 2.5*slots/1048576);
    float ms;
    if(which=='a' || which=='f'){
      ms=timeit([&]{k_comp<16,0><<<4096,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 4096, 4-byte entries", ms);
      ms=timeit([&]{k_comp<16,0><<<2048,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 2048, 4-byte entries", ms);
      ms=timeit([&]{k_comp<16,1><<<4096,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 4096, 3-byte entries", ms);
      ms=timeit([&]{k_comp<16,1><<<2048,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 2048, 3-byte entries", ms);
      ms=timeit([&]{k_comp<16,2><<<4096,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 4096, 20-bit entries", ms);
      ms=timeit([&]{k_comp<16,2><<<2048,256>>>(keys,tab,left,right,n);}); P("strided ilp16 grid 2048, 20-bit entries", ms);
      ms=timeit([&]{k_comp<8,2><<<4096,256>>>(keys,tab,left,right,n);}); P("strided ilp8 grid 4096, 20-bit entries", ms);
      ms=timeit([&]{k_comp<8,2><<<8192,256>>>(keys,tab,left,right,n);}); P("strided ilp8 grid 8192, 20-bit entries", ms);
    }

    if(which=='b'){
      #define RC16(NL,FMT,OCC,NTS,G) ms=timeit([&]{k_chunk16<NL,FMT,OCC,NTS><<<G,256>>>(keys,tab,left,right,n);}); \
        { char nm[128]; snprintf(nm,128,"chunk16 %dx128 keys/wave fmt %d occ %d nt-stores %d grid %d",NL,FMT,OCC,NTS,G); P(nm,ms); }
      #define RC8(NL,FMT,OCC,G) ms=timeit([&]{k_chunk8<NL,FMT,OCC><<<G,256>>>(keys,tab,left,right,n);}); \
        { char nm[128]; snprintf(nm,128,"chunk8 %dx64 keys/wave fmt %d occ %d grid %d",NL,FMT,OCC,G); P(nm,ms); }
      #define RG(NG,FMT,G) ms=timeit([&]{k_glds<NG,FMT><<<G,256>>>(keys,tab,left,right,n);}); \
        { char nm[128]; snprintf(nm,128,"LDS-DMA keys %dx128 per wave fmt %d grid %d",NG,FMT,G); P(nm,ms); }
      RC16(4,2,1,1,1024) RC16(4,2,1,1,1536) RC16(4,2,1,1,2048) RC16(4,2,1,1,3072) RC16(4,2,1,1,4096) RC16(4,2,1,1,8192)
      RC16(4,2,2,1,2048) RC16(4,2,4,1,2048) RC16(4,2,6,1,2048) RC16(4,2,8,1,2048) RC16(4,2,8,1,4096)
      RC16(3,2,1,1,2048) RC16(3,2,1,1,4096) RC16(6,2,1,1,2048) RC16(6,2,1,1,1024) RC16(2,2,8,1,4096) RC16(2,2,8,1,8192)
      RC16(4,2,1,0,2048) RC16(4,1,1,1,2048) RC16(4,1,1,1,4096) RC16(6,1,1,1,2048) RC16(4,0,1,1,2048)
      RC8(8,2,1,2048) RC8(8,2,1,4096) RC8(16,2,1,2048) RC8(16,2,1,1024) RC8(8,1,1,2048) RC8(8,0,1,2048) RC8(16,0,1,2048)
      RG(4,2,2048) RG(8,2,1024) RG(8,2,2048) RG(4,2,4096) RG(8,1,1024) RG(8,0,1024)
    }
    if(which=='a' || which=='p'){
      ms=timeit([&]{k_parts<16,0><<<4096,256>>>(keys,tab,slots,left,right,out,n);}); P("parts: keys + lookups + stores", ms);
      ms=timeit([&]{k_parts<16,1><<<4096,256>>>(keys,tab,slots,left,right,out,n);}); P("parts: keys + lookups (no stores)", ms);
      ms=timeit([&]{k_parts<16,2><<<4096,256>>>(keys,tab,slots,left,right,out,n);}); P("parts: lookups + stores (keys computed)", ms);
      ms=timeit([&]{k_parts<16,3><<<4096,256>>>(keys,tab,slots,left,right,out,n);}); P("parts: keys from a 1 MiB window + lookups + stores", ms);
    }
    if(which=='a' || which=='s'){
      #define RPF(NL,D,GRAN,FMT,G) ms=timeit([&]{k_pf<NL,D,GRAN,FMT><<<G,256>>>(keys,tab,left,right,n,sink);}); \
        { char nm[128]; snprintf(nm,128,"chunks %dx128 keys/wave, prefetch %d trips ahead per %d B, fmt %d, grid %d",NL,D,GRAN,FMT,G); P(nm,ms); }
      RPF(1,0,128,0,2048) RPF(1,1,128,0,2048) RPF(1,2,128,0,2048) RPF(1,4,128,0,2048) RPF(1,2,64,0,2048)
      RPF(2,0,128,0,2048) RPF(2,1,128,0,2048) RPF(2,2,128,0,2048) RPF(2,1,64,0,2048)
      RPF(4,0,128,0,2048) RPF(4,1,128,0,2048) RPF(4,1,128,0,1024) RPF(4,1,128,0,1536)
      RPF(8,0,128,0,1024) RPF(8,1,128,0,1024)
      RPF(1,2,128,2,2048) RPF(2,0,128,2,2048) RPF(2,1,128,2,2048) RPF(2,2,128,2,2048) RPF(4,0,128,2,2048) RPF(4,1,128,2,2048) RPF(4,1,128,2,1024) RPF(8,0,128,2,1024) RPF(8,1,128,2,1024)
      RPF(2,1,128,1,2048) RPF(4,1,128,1,2048) RPF(4,1,128,1,1024)
    }
  }
  if(which=='a' || which=='r'){
    float ms;
    ms=timeit([&]{k_stream_pf<2,0><<<2048,256>>>(keys,n,out);}); printf("stream read 0.8 GB, 2x128 chunks, no prefetch: %.3f ms %.0f GB/s\n", ms, 8.0*n/ms/1e6);
    ms=timeit([&]{k_stream_pf<2,1><<<2048,256>>>(keys,n,out);}); printf("stream read 0.8 GB, 2x128 chunks, prefetch 1 : %.3f ms %.0f GB/s\n", ms, 8.0*n/ms/1e6);
    ms=timeit([&]{k_stream_pf<2,2><<<2048,256>>>(keys,n,out);}); printf("stream read 0.8 GB, 2x128 chunks, prefetch 2 : %.3f ms %.0f GB/s\n", ms, 8.0*n/ms/1e6);
    ms=timeit([&]{k_stream_pf<8,0><<<2048,256>>>(keys,n,out);}); printf("stream read 0.8 GB, 8x128 chunks, no prefetch: %.3f ms %.0f GB/s\n", ms, 8.0*n/ms/1e6);
    ms=timeit([&]{k_stream_pf<8,1><<<2048,256>>>(keys,n,out);}); printf("stream read 0.8 GB, 8x128 chunks, prefetch 1 : %.3f ms %.0f GB/s\n", ms, 8.0*n/ms/1e6);
  }
  return 0;
}
