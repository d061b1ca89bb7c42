// Round-6 micro-benchmarks for the GENERAL-key hash-join probe (C3 with sparse 64-bit keys: 1e8 probe keys x 1e6 build keys).
// The shipped route is three HBM passes (partition -> LDS probe -> un-permute, 5.2 GB moved, 1.83 ms).  Measured here before
// anything is built into the product: ONE pass in which every XCD keeps an eighth of the build side (the keys whose mixed
// hash has its top three bits = the XCD's number) as an open-addressing table resident in ITS OWN 4 MiB L2, every XCD
// streams ALL probe keys (eight readers of one stream: one of them pays HBM, seven the Infinity Cache) and answers only
// the keys of its slice, storing the build row at the probe row's own position (so the output needs no un-permute).
//   x : the one-pass form, divergent lookups (lanes whose key is not the XCD's idle)
//   q : the one-pass form, a wave compacts its slice's keys through LDS and looks them up with full lanes
//   p : the two-pass form's second pass: probe rows pre-partitioned 8 ways (stable), each XCD reads only its list (12 B/row)
//   r : eight XCDs reading one stream (no lookups, no stores): what the replicated key read costs alone
// usage: ubench3 [nprobe] [nbuild] [slots_per_slice]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);}}while(0)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__host__ __device__ __forceinline__ uint32_t h32(uint32_t x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }
__device__ __forceinline__ uint32_t xcc_id(){ uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }
__host__ __device__ __forceinline__ uint64_t build_key(uint32_t i){ return mix64(0x1234567ull + i) | 1ull; } // sparse, (practically) unique
struct Slot { uint32_t klo, khi, row1; }; // 12 bytes
__host__ __device__ __forceinline__ uint32_t home_of(uint64_t h, uint32_t nslots){
#ifdef __HIP_DEVICE_COMPILE__
  return __umulhi((uint32_t)h, nslots);
#else
  return (uint32_t)(((uint64_t)(uint32_t)h * nslots) >> 32);
#endif
}

__global__ void k_gen_probe(uint64_t* pk, size_t n, uint32_t nb){
  for(size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; i<n; i+=(size_t)gridDim.x*blockDim.x) pk[i] = build_key(h32((uint32_t)i) % nb);
}
__global__ void k_build(Slot* tab, uint32_t nslots, uint32_t nb){
  const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=nb) return;
  const uint64_t k = build_key(i), h = mix64(k);
  Slot* t = tab + (size_t)(h>>61)*nslots;
  uint32_t s = home_of(h, nslots);
  while(atomicCAS(&t[s].row1, 0u, i+1u) != 0u){ s++; if(s==nslots) s=0; }
  t[s].klo=(uint32_t)k; t[s].khi=(uint32_t)(k>>32);
}
__device__ __forceinline__ uint32_t lookup(const Slot* __restrict__ t, uint32_t nslots, uint64_t k, uint64_t h){
  uint32_t s = home_of(h, nslots);
  while(true){
    const uint32_t* p = (const uint32_t*)(t+s);
    const uint32_t a=p[0], b=p[1], c=p[2];
    if(!c) return 0xffffffffu;
    if(a==(uint32_t)k && b==(uint32_t)(k>>32)) return c-1u;
    s++; if(s==nslots) s=0;
  }
}

// U lookups with their first slot loads all in flight; the (rarer) continuation of a probe sequence runs per lookup
template<int U>
__device__ __forceinline__ void lookup_batch(const Slot* __restrict__ t, uint32_t nslots, const uint64_t (&k)[U], const bool (&act)[U], uint32_t (&m)[U]){
  uint32_t s[U], a[U], b[U], c[U];
#pragma unroll
  for(int u=0;u<U;u++){ s[u] = act[u] ? home_of(mix64(k[u]), nslots) : 0u; const uint32_t* p=(const uint32_t*)(t+s[u]); a[u]=p[0]; b[u]=p[1]; c[u]=p[2]; }
#pragma unroll
  for(int u=0;u<U;u++){
    uint32_t r = 0xffffffffu;
    if(act[u]){
      uint32_t ss=s[u], aa=a[u], bb=b[u], cc=c[u];
      while(true){
        if(!cc) break;
        if(aa==(uint32_t)k[u] && bb==(uint32_t)(k[u]>>32)){ r=cc-1u; break; }
        ss++; if(ss==nslots) ss=0;
        const uint32_t* p=(const uint32_t*)(t+ss); aa=p[0]; bb=p[1]; cc=p[2];
      }
    }
    m[u]=r;
  }
}
__global__ void k_check(const uint64_t* out8, const uint32_t* out4, size_t n, uint32_t nb, unsigned long long* bad){
  unsigned long long c=0;
  for(size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; i<n; i+=(size_t)gridDim.x*blockDim.x){
    const uint32_t e = h32((uint32_t)i) % nb;
    const uint64_t g = out8 ? out8[i] : out4[i];
    if(g != e) c++;
  }
  if(c) atomicAdd(bad, c);
}

// MODE 0: slice = XCC id; 1: slice = blockIdx & 7; 2: slice = (blockIdx >> 3) & 7 (an XCD sees all eight tables: the control)
__device__ __forceinline__ uint32_t my_slice(int mode){ return mode==0 ? (xcc_id()&7u) : mode==1 ? (blockIdx.x&7u) : ((blockIdx.x>>3)&7u); }

// x: divergent one-pass.  A workgroup takes tiles of 256*U rows off its slice's ticket.
template<int U, int OUT8>
__global__ __launch_bounds__(256) void k_x(const uint64_t* __restrict__ keys, size_t n, const Slot* __restrict__ tab, uint32_t nslots,
                                          uint64_t* __restrict__ out8, uint32_t* __restrict__ out4, unsigned* tickets, int mode){
  const uint32_t sl = my_slice(mode);
  const Slot* t = tab + (size_t)sl*nslots;
  __shared__ unsigned s_t;
  const size_t ntiles = (n + 256*U - 1)/(256*U);
  if(threadIdx.x==0) s_t = atomicAdd(&tickets[sl*32], 1u); // the block's rank among its slice's blocks
  __syncthreads();
  const size_t brank = s_t, bstride = gridDim.x/8; // (an uneven split of blocks over XCDs: ranks >= bstride wrap around)
  for(size_t tile = brank % bstride; tile < ntiles; tile += bstride){
    if(brank >= bstride) break; // (surplus blocks of an XCD idle; its missing blocks are covered below)
    const size_t base = tile*(size_t)(256*U);
    uint64_t k[U];
#pragma unroll
    for(int j=0;j<U;j++){ const size_t r = base + j*256 + threadIdx.x; k[j] = r<n ? __builtin_nontemporal_load(keys+r) : 0; }
    bool act[U]; uint32_t m[U];
#pragma unroll
    for(int j=0;j<U;j++){ const size_t r = base + j*256 + threadIdx.x; act[j] = r<n && (uint32_t)(mix64(k[j])>>61)==sl; }
    lookup_batch<U>(t, nslots, k, act, m);
#pragma unroll
    for(int j=0;j<U;j++){ const size_t r = base + j*256 + threadIdx.x; if(act[j]){ if(OUT8) out8[r] = m[j]; else out4[r] = m[j]; } }
  }
}

// q: a wave compacts the keys of its slice through LDS (U rounds of 64 rows), then looks them up with full lanes.
template<int U, int OUT8, int ILP>
__global__ __launch_bounds__(256) void k_q(const uint64_t* __restrict__ keys, size_t n, const Slot* __restrict__ tab, uint32_t nslots,
                                          uint64_t* __restrict__ out8, uint32_t* __restrict__ out4, unsigned* tickets, int mode){
  const uint32_t sl = my_slice(mode);
  const Slot* t = tab + (size_t)sl*nslots;
  __shared__ unsigned s_t;
  __shared__ uint64_t qk[4][64*U];
  __shared__ uint16_t qr[4][64*U];
  const int w = threadIdx.x>>6, lane = threadIdx.x&63;
  const size_t ntiles = (n + 256*U - 1)/(256*U);
  if(threadIdx.x==0) s_t = atomicAdd(&tickets[sl*32], 1u); // the block's rank among its slice's blocks
  __syncthreads();
  const size_t brank = s_t, bstride = gridDim.x/8; // (an uneven split of blocks over XCDs: ranks >= bstride wrap around)
  for(size_t tile = brank % bstride; tile < ntiles; tile += bstride){
    if(brank >= bstride) break; // (surplus blocks of an XCD idle; its missing blocks are covered below)
    const size_t base = tile*(size_t)(256*U) + (size_t)w*(64*U); // a wave's 64*U rows are consecutive
    uint64_t k[U];
#pragma unroll
    for(int j=0;j<U;j++){ const size_t r = base + j*64 + lane; k[j] = r<n ? __builtin_nontemporal_load(keys+r) : 0; }
    uint32_t cnt = 0;
#pragma unroll
    for(int j=0;j<U;j++){
      const size_t r = base + j*64 + lane;
      const bool mine = r<n && (uint32_t)(mix64(k[j])>>61)==sl;
      const uint64_t bm = __ballot(mine);
      if(mine){
        const uint32_t p = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm>>32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0));
        qk[w][p] = k[j]; qr[w][p] = (uint16_t)(j*64+lane);
      }
      cnt += (uint32_t)__popcll(bm);
    }
    // (wave-private queue: no barrier needed, LDS ops of one wave are ordered)
    for(uint32_t i0 = 0; i0 < cnt; i0 += 64*ILP){
      uint64_t kk[ILP]; uint32_t rr[ILP], m[ILP]; bool act[ILP];
#pragma unroll
      for(int u=0;u<ILP;u++){ const uint32_t i = i0 + u*64 + lane; act[u] = i<cnt; const uint32_t ic = i<cnt ? i : 0; kk[u]=qk[w][ic]; rr[u]=qr[w][ic]; }
      lookup_batch<ILP>(t, nslots, kk, act, m);
#pragma unroll
      for(int u=0;u<ILP;u++){
        const uint32_t i = i0 + u*64 + lane;
        if(i<cnt){ if(OUT8) out8[base+rr[u]] = m[u]; else out4[base+rr[u]] = m[u]; }
      }
    }
  }
}

// p: second pass of the two-pass form: slice lists (key, row) in row order; a workgroup takes tiles of 256*U entries of ITS slice.
template<int U, int OUT8, int WHAT>
__global__ __launch_bounds__(256) void k_p(const uint64_t* __restrict__ pkey, const uint32_t* __restrict__ prow, const size_t* __restrict__ soff,
                                          const Slot* __restrict__ tab, uint32_t nslots,
                                          uint64_t* __restrict__ out8, uint32_t* __restrict__ out4, unsigned* tickets, int mode){
  const uint32_t sl = my_slice(mode);
  const Slot* t = tab + (size_t)sl*nslots;
  __shared__ unsigned s_t;
  const size_t lo = soff[sl], hi = soff[sl+1], n = hi-lo;
  const size_t ntiles = (n + 256*U - 1)/(256*U);
  if(threadIdx.x==0) s_t = atomicAdd(&tickets[sl*32], 1u); // the block's rank among its slice's blocks
  __syncthreads();
  const size_t brank = s_t, bstride = gridDim.x/8; // (an uneven split of blocks over XCDs: ranks >= bstride wrap around)
  for(size_t tile = brank % bstride; tile < ntiles; tile += bstride){
    if(brank >= bstride) break; // (surplus blocks of an XCD idle; its missing blocks are covered below)
    const size_t base = lo + tile*(size_t)(256*U);
    uint64_t k[U]; uint32_t r[U], m[U];
#pragma unroll
    for(int j=0;j<U;j++){ const size_t i = base + j*256 + threadIdx.x; const size_t ic = i<hi ? i : lo; k[j]=__builtin_nontemporal_load(pkey+ic); r[j]=__builtin_nontemporal_load(prow+ic); }
    bool act[U];
#pragma unroll
    for(int j=0;j<U;j++) act[j] = base + j*256 + threadIdx.x < hi;
    if(WHAT!=2) lookup_batch<U>(t, nslots, k, act, m);
    else {
#pragma unroll
      for(int j=0;j<U;j++) m[j] = (uint32_t)k[j];
    }
    if(WHAT==1){
      uint32_t a=0;
#pragma unroll
      for(int j=0;j<U;j++) a += m[j];
      if(a==0x12345u) out4[0]=a;
      continue;
    }
#pragma unroll
    for(int j=0;j<U;j++){ const size_t i = base + j*256 + threadIdx.x; if(i<hi){ if(OUT8) out8[r[j]] = m[j]; else out4[r[j]] = m[j]; } }
  }
}

// r: eight readers of one stream
template<int U>
__global__ __launch_bounds__(256) void k_r(const uint64_t* __restrict__ keys, size_t n, uint64_t* sink, unsigned* tickets, int mode, int readers){
  const uint32_t sl = my_slice(mode);
  __shared__ unsigned s_t;
  const size_t ntiles = (n + 256*U - 1)/(256*U);
  uint64_t acc=0;
  if(threadIdx.x==0) s_t = atomicAdd(&tickets[(readers==8? sl : 0)*32], 1u);
  __syncthreads();
  const size_t brank = s_t, bstride = readers==8 ? gridDim.x/8 : gridDim.x;
  for(size_t tile = brank % bstride; tile < ntiles; tile += bstride){
    if(brank >= bstride) break;
    const size_t base = tile*(size_t)(256*U);
#pragma unroll
    for(int j=0;j<U;j++){ const size_t r = base + j*256 + threadIdx.x; acc += r<n ? __builtin_nontemporal_load(keys+r) : 0; }
  }
  if(acc==0x1234567) sink[0]=acc;
}

int main(int argc, char** argv){
  const size_t n = argc>1 ? (size_t)atof(argv[1]) : 100000000;
  const uint32_t nb = argc>2 ? (uint32_t)atof(argv[2]) : 1000000;
  uint32_t nslots = argc>3 ? (uint32_t)atof(argv[3]) : 0;
  if(!nslots) nslots = (uint32_t)(nb/8/0.6);
  printf("n=%zu nb=%u slots/slice=%u (%.2f MiB per slice, load %.2f)\n", n, nb, nslots, nslots*12.0/1048576, nb/8.0/nslots);
  uint64_t *pk, *out8, *sink; uint32_t* out4; Slot* tab; unsigned* tickets; unsigned long long* bad;
  CK(hipMalloc(&pk, n*8)); CK(hipMalloc(&out8, n*8)); CK(hipMalloc(&out4, n*4)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&tab, (size_t)8*nslots*12)); CK(hipMemset(tab, 0, (size_t)8*nslots*12));
  CK(hipMalloc(&tickets, 8*32*4)); CK(hipMalloc(&bad, 8));
  k_gen_probe<<<4096,256>>>(pk, n, nb);
  k_build<<<(nb+255)/256,256>>>(tab, nslots, nb);
  CK(hipDeviceSynchronize());
  // the two-pass form's input: stable 8-way partition on the host
  std::vector<uint64_t> hk(n); CK(hipMemcpy(hk.data(), pk, n*8, hipMemcpyDeviceToHost));
  std::vector<size_t> soff(9,0);
  for(size_t i=0;i<n;i++) soff[(mix64(hk[i])>>61)+1]++;
  for(int s=0;s<8;s++) soff[s+1]+=soff[s];
  std::vector<uint64_t> hpk(n); std::vector<uint32_t> hpr(n);
  { std::vector<size_t> cur(soff.begin(), soff.begin()+8); for(size_t i=0;i<n;i++){ const int s=(int)(mix64(hk[i])>>61); hpk[cur[s]]=hk[i]; hpr[cur[s]]=(uint32_t)i; cur[s]++; } }
  uint64_t* ppk; uint32_t* ppr; size_t* dsoff;
  CK(hipMalloc(&ppk,n*8)); CK(hipMalloc(&ppr,n*4)); CK(hipMalloc(&dsoff,9*sizeof(size_t)));
  CK(hipMemcpy(ppk,hpk.data(),n*8,hipMemcpyHostToDevice)); CK(hipMemcpy(ppr,hpr.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dsoff,soff.data(),9*sizeof(size_t),hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, bool check8, bool check4){
    float best=1e9, sum=0; const int reps=5;
    for(int it=0; it<reps+2; it++){
      CK(hipMemsetAsync(tickets,0,8*32*4));
      if(it==0){ CK(hipMemsetAsync(out8,0xff,n*8)); CK(hipMemsetAsync(out4,0xff,n*4)); }
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(it>=2){ sum+=ms; if(ms<best) best=ms; }
    }
    CK(hipGetLastError());
    unsigned long long hb=0;
    if(check8||check4){ CK(hipMemset(bad,0,8)); k_check<<<2048,256>>>(check8?out8:nullptr, out4, n, nb, bad); CK(hipMemcpy(&hb,bad,8,hipMemcpyDeviceToHost)); }
    printf("%-44s min %.3f ms  avg %.3f ms  %s\n", name, best, sum/reps, (check8||check4) ? (hb? "MISMATCH":"ok") : "");
    fflush(stdout);
  };
  char nm[128];
  for(int mode=0; mode<3; mode++){
    const char* mn = mode==0?"xcc":mode==1?"b&7":"ctrl";
    for(int grid : {2048, 4096}){
      snprintf(nm,128,"r stream x8 readers U8 [%s g%d]",mn,grid); run(nm,[&]{ k_r<8><<<grid,256>>>(pk,n,sink,tickets,mode,8); },false,false);
    }
    snprintf(nm,128,"r stream x1 reader U8 [%s g2048]",mn); run(nm,[&]{ k_r<8><<<2048,256>>>(pk,n,sink,tickets,mode,1); },false,false);
    for(int grid : {2048, 8192}){
      snprintf(nm,128,"x divergent U8 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_x<8,1><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },true,false);
      snprintf(nm,128,"x divergent U8 out4 [%s g%d]",mn,grid); run(nm,[&]{ k_x<8,0><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },false,true);
      snprintf(nm,128,"x divergent U16 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_x<16,1><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },true,false);
      snprintf(nm,128,"q compacted U16 ILP2 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_q<16,1,2><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },true,false);
      snprintf(nm,128,"q compacted U16 ILP2 out4 [%s g%d]",mn,grid); run(nm,[&]{ k_q<16,0,2><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },false,true);
      snprintf(nm,128,"q compacted U16 ILP4 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_q<16,1,4><<<grid,256>>>(pk,n,tab,nslots,out8,out4,tickets,mode); },true,false);
      snprintf(nm,128,"p partitioned U8 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_p<8,1,0><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },true,false);
      snprintf(nm,128,"p partitioned U8 out4 [%s g%d]",mn,grid); run(nm,[&]{ k_p<8,0,0><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },false,true);
      snprintf(nm,128,"p partitioned U8 lookups only [%s g%d]",mn,grid); run(nm,[&]{ k_p<8,0,1><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },false,false);
      snprintf(nm,128,"p partitioned U8 out4 stores only [%s g%d]",mn,grid); run(nm,[&]{ k_p<8,0,2><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },false,false);
      snprintf(nm,128,"p partitioned U8 out8 stores only [%s g%d]",mn,grid); run(nm,[&]{ k_p<8,1,2><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },false,false);
      snprintf(nm,128,"p partitioned U4 out4 [%s g%d]",mn,grid); run(nm,[&]{ k_p<4,0,0><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },false,true);
      snprintf(nm,128,"p partitioned U16 out8 [%s g%d]",mn,grid); run(nm,[&]{ k_p<16,1,0><<<grid,256>>>(ppk,ppr,dsoff,tab,nslots,out8,out4,tickets,mode); },true,false);
    }
  }
  return 0;
}
