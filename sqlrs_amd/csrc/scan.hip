// scan.hip — single-pass exclusive prefix sum (decoupled look-back) over u32 counts.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

// Tile = 1024 * CH elements; wave w owns 256 * CH contiguous elements read as CH coalesced chunks of
// 256 (one uint4 per lane).  HBM traffic: 4 B read + 4/8 B written per element.
// CH = 4 (4096-element tiles) for short inputs; CH = 16 for long ones: a tile takes its index from ONE atomic counter, which
// sustains ~88 tickets/us — 1e8 counts in 4096-element tiles are 24 414 tickets = 0.28 ms of ticketing for 0.2 ms of memory
// traffic (the count -> scan -> fill join of duplicate build keys: scans 0.58 -> ms; 6.3 M entries: 38 -> us)
template <int CH>
__global__ __launch_bounds__(BLOCK) void scan_u32_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                         uint64_t *__restrict__ out64,
                                                         uint32_t *__restrict__ out32,
                                                         uint64_t *desc, unsigned *ticket,
                                                         uint64_t *total, int64_t num_tiles) {
  __shared__ int64_t s_tile;
  __shared__ uint64_t s_wave[WAVES_PER_BLOCK];
  __shared__ uint64_t s_excl;
  // one ticket per LB_TILES_PER_TICKET consecutive tiles (a single atomic counter sustains only
  // ~88 tickets/us on MI355X)
  if (threadIdx.x == 0) s_tile = (int64_t)atomicAdd(ticket, 1u) * LB_TILES_PER_TICKET;
  __syncthreads();
  const int64_t tile0 = s_tile;
  const int lane = lane_id(), w = wave_id();
  for (int sub = 0; sub < LB_TILES_PER_TICKET; sub++) {
  const int64_t tile = tile0 + sub;
  if (tile >= num_tiles) break;
  const int64_t wbase = tile * (1024 * CH) + (int64_t)w * (256 * CH);
  uint32_t v[CH][4];
  uint32_t lane_excl[CH]; // exclusive prefix of this lane's uint4 within the wave's 256 * CH
  uint32_t carry = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    int64_t i = wbase + c * 256 + lane * 4;
    if (i + 3 < n) {
      uint4 x = *reinterpret_cast<const uint4 *>(in + i);
      v[c][0] = x.x; v[c][1] = x.y; v[c][2] = x.z; v[c][3] = x.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) v[c][k] = (i + k < n) ? in[i + k] : 0u;
    }
    uint32_t s = v[c][0] + v[c][1] + v[c][2] + v[c][3];
    uint32_t inc = wave_iscan_u32(s);
    lane_excl[c] = carry + inc - s;
    carry += (uint32_t)__shfl((int)inc, 63, 64);
  }
  if (lane == 0) s_wave[w] = carry;
  __syncthreads();
  if (w == 0) {
    uint64_t agg = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    uint64_t excl = lookback_wave(desc, tile, agg);
    if (lane == 0) {
      s_excl = excl;
      if (tile == num_tiles - 1) *total = excl + agg;
    }
  }
  __syncthreads();
  uint64_t base = s_excl;
  for (int k = 0; k < w; k++) base += s_wave[k];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    int64_t i = wbase + c * 256 + lane * 4;
    uint64_t p = base + lane_excl[c];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i + k < n) {
        if (out64) out64[i + k] = p;
        if (out32) out32[i + k] = (uint32_t)p;
      }
      p += v[c][k];
    }
  }
  __syncthreads(); // s_wave / s_excl are reused by the next tile
  } // sub
}

void exclusive_scan_u32(Ctx *ctx, const uint32_t *in, int64_t n, uint64_t *out64, uint32_t *out32,
                        uint64_t *total) {
  if (n <= 0) {
    SQ_HIP(hipMemsetAsync(total, 0, 8, ctx->stream));
    return;
  }
  ProfScope ps(ctx, "scan_u32");
  const bool big = n >= (1 << 20); // (>= 64 tiles of 16 384)
  int64_t tiles = ceil_div(n, big ? 16384 : 4096);
  BufP desc = ctx->alloc_zero(8 * (size_t)tiles + 8);
  unsigned *ticket = (unsigned *)(desc->as<uint64_t>() + tiles);
  const dim3 g((unsigned)ceil_div(tiles, LB_TILES_PER_TICKET)), b(BLOCK);
  if (big) scan_u32_kernel<16><<<g, b, 0, ctx->stream>>>(in, n, out64, out32, desc->as<uint64_t>(), ticket, total, tiles);
  else scan_u32_kernel<4><<<g, b, 0, ctx->stream>>>(in, n, out64, out32, desc->as<uint64_t>(), ticket, total, tiles);
  SQ_HIP(hipGetLastError());
}

} // namespace sq
