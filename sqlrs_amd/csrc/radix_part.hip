// radix_part.hip — LDS-staged hash partitioning of rows (key, up to two 8-byte values, row id,
// validity flags) into P buckets, one or two levels of <= 512-way multi-split.
//
// Direct scattered stores run at the random-access rate of the memory system (~90 G stores/s
// on MI355X, profiles/r01_ubench_mi355x.txt) no matter how the lines fill up later, so each
// tile of 6144 rows is first sorted by bucket inside LDS and then written as bucket-contiguous
// runs: consecutive lanes store consecutive addresses (256 digits: runs of ~24 rows).
//
// Per level:  hist (8 B/row read: keys) -> exclusive scan of the [segment][digit][tile] count
// matrix = start of every run -> scatter (all columns read once, written once).
// Level 2 splits every level-1 bucket again (MSD order), giving up to 65536 buckets.
// When the key range is known the row id rides in the key word (KeyPack, radix_part.hpp):
// 16 instead of 20 bytes per row.
#include <cstdio>
#include <cstdlib>

#include "device_utils.hpp"
#include "radix_part.hpp"
#include "radix_part_kernels.hpp" // every kernel of the partition levels (namespace sq)

namespace sq {

namespace {

// Geometry of one level.  Either planned on the host from the segment boundaries (plan_level +
// upload_level + rp_make_tiles_kernel) or built on the device from the chunk table of a chunked first
// level (rp_chunk_plan_kernel): the kernels only see the device arrays.
struct Level {
  std::vector<int64_t> seg_mat, seg_start; // per segment (seg_start has nseg + 1 entries); host plan only
  std::vector<uint32_t> seg_tiles, seg_tile_base;
  int64_t mat_entries = 0;
  uint32_t num_tiles = 0, nseg = 0;
  // the four arrays on the device: ONE upload per level (seven small copies per level before)
  std::vector<uint8_t> blob; // host source of the upload (must outlive it)
  BufP dev, tiles;
  const int64_t *d_seg_start = nullptr, *d_seg_mat = nullptr;
  const uint32_t *d_seg_tiles = nullptr, *d_seg_tile_base = nullptr;
};

// per-segment geometry only (<= 513 entries); the Tile descriptors themselves are filled on the
// device (rp_make_tiles_kernel): building and uploading 81 K of them on the host left the GPU
// idle for 0.2 ms per level
Level plan_level(const std::vector<int64_t> &seg_start, uint32_t digits, int RP_TILE) {
  Level L;
  L.seg_start = seg_start;
  size_t nseg = seg_start.size() - 1;
  L.nseg = (uint32_t)nseg;
  for (size_t s = 0; s < nseg; s++) {
    int64_t len = seg_start[s + 1] - seg_start[s];
    uint32_t nt = (uint32_t)ceil_div(len, RP_TILE);
    L.seg_mat.push_back(L.mat_entries);
    L.seg_tiles.push_back(nt);
    L.seg_tile_base.push_back(L.num_tiles);
    L.num_tiles += nt;
    L.mat_entries += (int64_t)nt * digits;
  }
  return L;
}

// device layout of the four per-segment arrays inside one block
struct LevelLayout {
  size_t o_start, o_mat, o_tiles, o_base, total;
  explicit LevelLayout(size_t nseg) {
    o_start = 0;
    o_mat = o_start + 8 * (nseg + 1);
    o_tiles = o_mat + 8 * nseg;
    o_base = o_tiles + round_up(4 * nseg, 8);
    total = o_base + round_up(4 * nseg, 8);
  }
};
void bind_level(Level &L, const LevelLayout &lay) {
  const uint8_t *d = L.dev->as<uint8_t>();
  L.d_seg_start = (const int64_t *)(d + lay.o_start);
  L.d_seg_mat = (const int64_t *)(d + lay.o_mat);
  L.d_seg_tiles = (const uint32_t *)(d + lay.o_tiles);
  L.d_seg_tile_base = (const uint32_t *)(d + lay.o_base);
}

void upload_level(Ctx *ctx, Level &L) {
  const size_t nseg = L.seg_tiles.size();
  const LevelLayout lay(nseg);
  L.blob.resize(lay.total);
  std::memcpy(L.blob.data() + lay.o_start, L.seg_start.data(), 8 * (nseg + 1));
  std::memcpy(L.blob.data() + lay.o_mat, L.seg_mat.data(), 8 * nseg);
  std::memcpy(L.blob.data() + lay.o_tiles, L.seg_tiles.data(), 4 * nseg);
  std::memcpy(L.blob.data() + lay.o_base, L.seg_tile_base.data(), 4 * nseg);
  L.dev = ctx->alloc(lay.total);
  SQ_HIP(hipMemcpyAsync(L.dev->p, L.blob.data(), lay.total, hipMemcpyHostToDevice, ctx->stream));
  bind_level(L, lay);
}

} // namespace

bool partition_rows(Ctx *ctx, const PartitionInput &in, uint32_t P_wanted, PartitionedRows *out) {
  const int64_t n = in.n;
  if (n <= 0 || n > 0xffffffffll || in.nv > 2) return false;
  out->bend_host.clear();
  out->slim = PartitionedRows::Slim();
  // digits per level: one level up to 256 buckets, else P = d1 * 2^p2_bits
  uint32_t p2_bits = 0, d1 = std::max(1u, P_wanted);
  if (P_wanted > 512) { // one level handles up to 512 digits (runs of >= 8 rows per tile)
    // split the digits evenly between the two levels: 2^p2_bits second-level digits with
    // 2^p2_bits >= sqrt(P) (runs get longer as a level's digit count drops)
    const char *p2_e = hook("SQLRS_RP_P2BITS"); // tuning hook, read per call (in-process A/B)
    const int p2_env = p2_e ? std::atoi(p2_e) : 0;
    p2_bits = 5;
    while (p2_bits < 8 && (1u << (2 * p2_bits)) < P_wanted) p2_bits++;
    if (p2_env >= 4 && p2_env <= 9) p2_bits = (uint32_t)p2_env;
    d1 = (uint32_t)ceil_div(P_wanted, 1u << p2_bits);
    if (d1 > 512) {
      p2_bits = 8;
      d1 = (uint32_t)ceil_div(P_wanted, 256);
    }
    if (d1 > 256 && p2_bits == 8) return false;
    // Very large batches that take the chunked first level: one bit more for level 1 (512 digits) and one less for
    // level 2.  Level 2 gains more from its longer runs than level 1 loses (C5 with sparse keys, same process:
    // 7.93 + 6.67 -> 7.80 + 6.26 ms); the bucket layout of the result does not depend on the split.  The arena
    // slack of 512 digits must still be a fraction of the input (see `chunked` below).
    if (!p2_e && p2_bits == 8 && 2 * d1 <= 512 && !in.key_validity && !in.val_validity[0] && !in.val_validity[1] &&
        2ull * std::min<uint64_t>((uint64_t)ceil_div(n, 6144), (uint64_t)ctx->num_cus) * (2 * d1) * 6144ull <= 2 * (uint64_t)n) {
      p2_bits = 7;
      d1 = (uint32_t)ceil_div(P_wanted, 1u << p2_bits);
    }
  }
  const uint32_t P = d1 << p2_bits;
  const bool flags = in.key_validity || in.val_validity[0] || in.val_validity[1];
  const int nv = in.nv;
  int rowbits = 1;
  while (rowbits < 32 && (1ll << rowbits) < n) rowbits++;
  const bool pack = !flags && nv <= 1 && in.pack.kbits != 0 && in.pack.kbits + rowbits <= 64;
  const KeyPack kp = pack ? in.pack : KeyPack();
  out->pack = kp;
  out->rec = nullptr;
  const int WG = 512;
  // rows per thread: 12 -> 6144-row tiles, 8 -> 4096-row tiles (two value columns); one
  // workgroup per CU either way (the staging area is ~140 KiB)
  const char *rows_e = hook("SQLRS_RP_ROWS"); // tuning only, read per call (in-process A/B)
  const int rows_env = rows_e ? std::atoi(rows_e) : 0;
  // 16 = 8192-row tiles (packed rows with one value column only: 128 KiB of staging, 256 VGPRs, no spills): the
  // default for very large batches — a third fewer barrier rounds per row (C5, one process: level 1 5.50 -> 5.36 ms,
  // level 2 3.66 -> 3.61 ms); smaller batches keep 6144-row tiles (less arena slack, more tiles per workgroup)
  // (two-level partitions only — the counting single level is slower with them, C4: 1.55 -> 1.96 ms — and not with the
  //  predicate on a column of its own: that instantiation needs more than 256 VGPRs and spills)
  // Round 5: the slim first level runs 768-thread workgroups over 6144-row tiles by default (twelve waves per CU instead of
  // eight, 159 VGPRs, no spill: C5 level 1 5.33 (8192-row tiles, 512 threads) / 5.39 (6144, 512) -> 4.95 ms in one process,
  // step 10.45 -> 10.10; 1024 threads x 6 rows: the same 4.98 with 3 spilled registers).  8192-row tiles only on request.
  const bool big16 = pack && nv == 1 && rows_env == 16;
  const int ROWS = nv > 1 ? 8 : (rows_env == 6 ? 6 : ((rows_env == 8 && pack) ? 8 : (big16 ? 16 : 12)));
  const int RP_TILE = WG * ROWS;
  const size_t lds = (size_t)RP_TILE * (pack ? 8 * (1 + nv) : 8 * (1 + nv) + 4 + 2 + 1) + (size_t)WG * (4 + 4 + 8);

  // (staggering the columns' start offsets inside their 2 MiB aligned blocks — same row, same HBM channel?
  //  — changed nothing; the 10-15 % spread of these kernels between processes follows physical placement)
  auto staggered = [&](size_t bytes, int) { return ctx->alloc(bytes); };
  // final level of a packed partition with one value column: 16-byte {key|row word, value} records instead of
  // two 8-byte columns — one store per row here, one load per row in the bucket pass, and a (tile, digit) run
  // of ~24 rows covers three cache lines instead of 2 x 1.5 (C5: level 2 4.93 -> 4.26 ms, bucket pass 1.81 ->
  // 1.67 ms in one process)
  const char *rec_e = hook("SQLRS_RP_REC"); // read per call: 0 = column form (in-process A/B, tools/ab_in_process.py)
  const bool use_rec = pack && nv == 1 && (ROWS == 12 || ROWS == 16) && !(rec_e && std::atoi(rec_e) == 0);
  struct Cols {
    BufP k, v0, v1, idx, fl, rec;
  };
  auto alloc_cols = [&](int64_t rows, bool final_level, Cols &c, RpOut &ro) {
    const size_t np = (size_t)rows + 1024; // + the sink rows of rp_scatter_kernel (workgroups of up to 1024 threads)
    c = Cols();
    if (final_level && use_rec) {
      c.rec = ctx->alloc(16 * np);
    } else {
      c.k = staggered(8 * np, 0);
      c.v0 = nv >= 1 ? staggered(8 * np, 1) : nullptr;
    }
    c.v1 = nv >= 2 ? staggered(8 * np, 2) : nullptr;
    c.idx = pack ? nullptr : staggered(4 * np, 3);
    c.fl = flags ? ctx->alloc(np) : nullptr;
    ro.key = c.k ? c.k->as<uint64_t>() : nullptr;
    ro.v0 = c.v0 ? c.v0->as<uint64_t>() : nullptr;
    ro.v1 = c.v1 ? c.v1->as<uint64_t>() : nullptr;
    ro.idx = c.idx ? c.idx->as<uint32_t>() : nullptr;
    ro.flags = c.fl ? c.fl->as<uint8_t>() : nullptr;
    ro.rec = c.rec ? (u64x2 *)c.rec->p : nullptr;
  };
  auto publish = [&](const Cols &c) {
    out->key = c.k; out->v0 = c.v0; out->v1 = c.v1; out->idx = c.idx; out->flags = c.fl; out->rec = c.rec;
  };
  // one level = hist + scan + scatter over the tiles of `L` (sink = first row behind the output columns)
  struct SlimLaunch { // level 2 of the slim form: rp_scatter_slim_kernel instead of rp_scatter_kernel
    SlimIn in;
    SlimOut out;
    uint32_t kshift, rbits;
    BufP total; // (out) the scan's grand total, device u64
  };
  // `before_scatter` (optional): called with the scanned offsets once they are queued and before the scatter is — what only needs
  // the offsets (the bucket starts and their way to the host) then runs ahead of the level's long kernel
  auto exec_level = [&](int level, uint32_t digits, const Level &L, const RpIn &rin, const RpOut &rout, int64_t sink,
                        BufP *offs_out, BufP premat = nullptr, SlimLaunch *slim = nullptr,
                        const std::function<void(const BufP &)> *before_scatter = nullptr) {
    const int64_t entries = std::max<int64_t>(L.mat_entries, 1);
    BufP mat = premat ? premat : ctx->alloc(4 * (size_t)entries);
    BufP offs = ctx->alloc(4 * (size_t)entries);
    BufP total = ctx->alloc(8);
    if (slim) slim->total = total;
    unsigned nt = L.num_tiles;
    const Tile *tp = (const Tile *)L.tiles->p;
    if (nt && !premat) {
      ProfScope ps(ctx, in.build_side ? "rp_hist_build" : "rp_hist");
#define SQ_RH1(R, PL) rp_hist_kernel<512, R, PL><<<dim3(nt), dim3(512), 0, ctx->stream>>>(rin.key, rin.key_validity, rin.flags, tp, P, p2_bits, level, digits, mat->as<uint32_t>(), kp)
#define SQ_RH(R) do { if (!rin.key_validity && !rin.flags) SQ_RH1(R, true); else SQ_RH1(R, false); } while (0)
      if (ROWS == 12) SQ_RH(12); else if (ROWS == 16) SQ_RH(16); else if (ROWS == 8) SQ_RH(8); else SQ_RH(6);
#undef SQ_RH
#undef SQ_RH1
      SQ_HIP(hipGetLastError());
    }
    // tile-major counts -> digit-major, scan, digit-major offsets -> tile-major for the scatter
    BufP mat_dm = ctx->alloc(4 * (size_t)entries);
    BufP offs_tm = ctx->alloc(4 * (size_t)entries);
    const size_t tlds = (size_t)RP_TB * (digits + 1) * 4;
    const unsigned tblocks = (unsigned)ceil_div(nt, RP_TB);
    if (nt)
      rp_transpose_kernel<false><<<dim3(tblocks), dim3(256), tlds, ctx->stream>>>(
          mat->as<uint32_t>(), mat_dm->as<uint32_t>(), tp, nt, digits);
    exclusive_scan_u32(ctx, mat_dm->as<uint32_t>(), L.mat_entries, nullptr, offs->as<uint32_t>(),
                       total->as<uint64_t>());
    if (nt)
      rp_transpose_kernel<true><<<dim3(tblocks), dim3(256), tlds, ctx->stream>>>(
          offs->as<uint32_t>(), offs_tm->as<uint32_t>(), tp, nt, digits);
    SQ_HIP(hipGetLastError());
    if (before_scatter) (*before_scatter)(offs);
    if (nt) {
      ProfScope ps(ctx, in.build_side ? "rp_scatter_build" : "rp_scatter");
      // one workgroup per CU slot; contiguous tile ranges (8 per workgroup at least)
      uint32_t wgs = std::min<uint32_t>(nt, (uint32_t)ctx->num_cus * ((ROWS == 6 || (ROWS == 8 && pack)) ? 2 : 1));
      uint32_t tpw = (uint32_t)ceil_div(nt, wgs);
      wgs = (uint32_t)ceil_div(nt, tpw);
      dim3 g(wgs), b((unsigned)WG);
      const int mode = level == 1 ? (flags ? RP_L1_NULL : RP_L1) : (flags ? RP_LN_FLAG : RP_LN);
      if (slim) {
        const size_t slds = (size_t)RP_TILE * 13 + 1024 + (size_t)WG * 16 + 512 + 128;
        // (1024-thread workgroups over the same 8192-row tile, eight rows per thread — sixteen waves per CU as in the bucket pass —
        //  were measured in round 5: 128 VGPRs, 17 spilled, level 2 3.00 -> 3.21 ms in one process; not kept)
        if (ROWS == 16) {
          auto kfn = rp_scatter_slim_kernel<512, 16>;
          allow_big_lds(ctx, kfn);
          kfn<<<g, b, slds, ctx->stream>>>(slim->in, slim->out, tp, p2_bits, digits, offs_tm->as<uint32_t>(), nt, tpw, sink,
                                          slim->kshift, slim->rbits);
        } else if (hook("SQLRS_RP_L2_WG") && std::atoi(hook("SQLRS_RP_L2_WG")) == 768) { // A/B hook, read per call
          auto kfn = rp_scatter_slim_kernel<768, 8>;
          allow_big_lds(ctx, kfn);
          kfn<<<g, dim3(768), slds + 256 * 16, ctx->stream>>>(slim->in, slim->out, tp, p2_bits, digits, offs_tm->as<uint32_t>(), nt, tpw, sink,
                                                            slim->kshift, slim->rbits);
        } else {
          auto kfn = rp_scatter_slim_kernel<512, 12>;
          allow_big_lds(ctx, kfn);
          kfn<<<g, b, slds, ctx->stream>>>(slim->in, slim->out, tp, p2_bits, digits, offs_tm->as<uint32_t>(), nt, tpw, sink,
                                          slim->kshift, slim->rbits);
        }
        SQ_HIP(hipGetLastError());
        slim->total = total;
        *offs_out = offs;
        return;
      }
#define SQ_RP1(NV, R, M, PK)                                                                                  \
  do {                                                                                                        \
    constexpr bool can_rec = NV == 1 && PK && (R == 12 || R == 16);                                           \
    auto kfn = rp_scatter_kernel<NV, 512, R, M, PK>;                                                          \
    if (can_rec && rout.rec) kfn = rp_scatter_kernel<NV, 512, R, M, PK, can_rec>;                             \
    allow_big_lds(ctx, kfn);                                                                                  \
    kfn<<<g, b, lds, ctx->stream>>>(rin, rout, tp, P, p2_bits, level, digits, offs_tm->as<uint32_t>(), nt, tpw, \
                                    sink, kp);                                                                  \
  } while (0)
#define SQ_RP(NV, R)                                                                                          \
  do {                                                                                                        \
    if (mode == RP_L1) { if (pack) SQ_RP1(NV, R, RP_L1, true); else SQ_RP1(NV, R, RP_L1, false); }            \
    else if (mode == RP_L1_NULL) SQ_RP1(NV, R, RP_L1_NULL, false);                                            \
    else if (mode == RP_LN) { if (pack) SQ_RP1(NV, R, RP_LN, true); else SQ_RP1(NV, R, RP_LN, false); }       \
    else SQ_RP1(NV, R, RP_LN_FLAG, false);                                                                    \
  } while (0)
      if (nv == 2) SQ_RP(2, 8);
      else if (ROWS == 6) { if (nv == 0) SQ_RP(0, 6); else SQ_RP(1, 6); }
      else if (ROWS == 8) { if (nv == 0) SQ_RP(0, 8); else SQ_RP(1, 8); }
      else if (ROWS == 16) { if (pack && mode == RP_LN) SQ_RP1(1, 16, RP_LN, true); else SQ_RP1(1, 16, RP_L1, true); }
      // (768 threads x 8 rows for the counting level 2 of hashed partitions: measured in round 5, 6.08 vs 6.08 ms — not kept)
      else { if (nv == 0) SQ_RP(0, 12); else SQ_RP(1, 12); }
#undef SQ_RP1
#undef SQ_RP
      SQ_HIP(hipGetLastError());
    }
    *offs_out = offs;
  };

  // host-planned level over `seg_start` segments (the host vectors of `L` are the source of async
  // uploads: `L` must outlive the bucket_starts() synchronisation that follows)
  auto run_level = [&](int level, uint32_t digits, const std::vector<int64_t> &seg_start, const RpIn &rin,
                       const RpOut &rout, BufP *offs_out, Level *L) {
    *L = plan_level(seg_start, digits, RP_TILE);
    L->tiles = ctx->alloc(sizeof(Tile) * (size_t)std::max<uint32_t>(L->num_tiles, 1));
    upload_level(ctx, *L);
    rp_make_tiles_kernel<<<dim3((unsigned)L->seg_tiles.size()), dim3(256), 0, ctx->stream>>>(
        L->d_seg_start, L->d_seg_mat, L->d_seg_tiles, L->d_seg_tile_base, RP_TILE, (Tile *)L->tiles->p);
    SQ_HIP(hipGetLastError());
    exec_level(level, digits, *L, rin, rout, n, offs_out);
  };

  // `host` receives a copy (the synchronisation that keeps L's upload sources alive pays for it)
  auto bucket_starts = [&](const Level &L, const BufP &offs, uint32_t digits, int64_t rows,
                           std::vector<uint32_t> *host) -> BufP {
    uint32_t nseg = L.nseg;
    int64_t total = (int64_t)nseg * digits + 1;
    BufP bs = ctx->alloc(4 * (size_t)total);
    rp_bucket_starts_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream>>>(
        offs->as<uint32_t>(), L.d_seg_mat, L.d_seg_tiles, L.d_seg_start, digits, nseg, rows, bs->as<uint32_t>());
    SQ_HIP(hipGetLastError());
    if (4 * (size_t)total <= ctx->pinned_bytes) { // (pinned staging buffer: a copy into pageable memory costs tens of us of host time)
      const uint32_t *h = (const uint32_t *)ctx->fetch(bs->p, 4 * (size_t)total);
      host->assign(h, h + total);
      return bs;
    }
    host->resize((size_t)total);
    SQ_HIP(hipMemcpyAsync(host->data(), bs->p, 4 * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return bs;
  };

  // The bucket starts of a LAST level need its scanned offsets, not its rows: kernel + copy to the host are queued AHEAD of the
  // level's scatter (exec_level's `before_scatter`) and the host waits for the copy's event behind the launches that follow — the
  // bucket pass's work list is planned while the GPU is busy (round trip + planning were a 55-70 us hole in front of the bucket
  // pass of every C5 step).  SQLRS_RP_EARLY_STARTS=0 (read per call): fetched behind the level as before.
  struct EarlyStarts {
    BufP bs;
    int64_t total = 0;
    std::function<void(const BufP &)> queue;
  };
  auto early_starts_for = [&](const Level &L, uint32_t digits, int64_t rows, EarlyStarts &es) {
    es.total = (int64_t)L.nseg * digits + 1;
    const char *es_e = hook("SQLRS_RP_EARLY_STARTS");
    const bool off = es_e && es_e[0] == '0';
    es.queue = [&L, &es, digits, rows, off, ctx](const BufP &offs) {
      if (off) return;
      BufP bs = ctx->alloc(4 * (size_t)es.total);
      rp_bucket_starts_kernel<<<dim3((unsigned)ceil_div(es.total, 256)), dim3(256), 0, ctx->stream>>>(
          offs->as<uint32_t>(), L.d_seg_mat, L.d_seg_tiles, L.d_seg_start, digits, L.nseg, rows, bs->as<uint32_t>());
      SQ_HIP(hipGetLastError());
      if (ctx->fetch_early(bs->p, 4 * (size_t)es.total)) es.bs = bs;
    };
  };
  // (the event also covers the level's uploads, queued ahead of it: their host sources may go)
  auto finish_starts = [&](EarlyStarts &es, const Level &L, const BufP &offs, uint32_t digits, int64_t rows, std::vector<uint32_t> *host) -> BufP {
    if (!es.bs) return bucket_starts(L, offs, digits, rows, host);
    const uint32_t *h = (const uint32_t *)ctx->fetch_early_wait();
    host->assign(h, h + es.total);
    return es.bs;
  };

  // ---- claimed single level (no histogram pass, optional fused row filter): one-level range partitions of packed
  // rows; see rp_claim_scatter_kernel.  Hashed buckets keep the counting level (their bucket pass takes a sentinel
  // row for a key of its own).
  const char *claim_e = hook("SQLRS_RP_CLAIM"); // test / tuning hook, read per call: 0 = never, 1 = whatever the batch size
  const int claim_env = claim_e ? std::atoi(claim_e) : -1;
  const bool claimable = p2_bits == 0 && pack && kp.dense && nv <= 1 && ROWS == 12 && P <= (uint32_t)WG && P >= 2 &&
                         claim_env != 0 && (claim_env == 1 || n >= (1ll << 22));
  if (claimable) {
    const uint32_t tiles1c = (uint32_t)ceil_div(n, RP_TILE);
    uint32_t wgs = std::min<uint32_t>(tiles1c, (uint32_t)ctx->num_cus);
    if (const char *wg_e = hook("SQLRS_RP_CHUNK_WGS")) // test hook, read per call: fewer workgroups = longer tile ranges per workgroup
      wgs = std::max(1u, std::min<uint32_t>(wgs, (uint32_t)std::atoi(wg_e)));
    const uint32_t tpw = (uint32_t)ceil_div(tiles1c, std::max(wgs, 1u));
    wgs = (uint32_t)ceil_div(tiles1c, std::max(tpw, 1u));
    // block size: the holes (about wgs * B / 2 per bucket) stay near 3 % of the rows; >= 16 rows = 256 bytes
    uint32_t B = 16;
    while (B < 256 && (uint64_t)(2 * B) * 16 * wgs * P <= (uint64_t)n) B *= 2;
    const uint32_t slack = 4096 + wgs * B;
    const uint64_t slots_max = (uint64_t)n + (uint64_t)n / 8 + (uint64_t)P * ((uint64_t)slack + B) + 64;
    const uint64_t pool_rows = slots_max + (uint64_t)1024 * wgs; // + one sink per workgroup (of up to 1024 threads)
    if (pool_rows <= 0xffffffffull) {
      BufP est = ctx->alloc_zero(4 * ((size_t)P + 1));
      // region start | region end | cursor | {kept rows (u64), overflow flag (u64)}: one buffer, one fetch
      const size_t plan_words = 3 * (size_t)P + (P & 1), plan_bytes = 4 * plan_words + 16;
      BufP plan = ctx->alloc(plan_bytes);
      SQ_HIP(hipMemsetAsync(plan->as<uint8_t>() + 4 * plan_words, 0, 16, ctx->stream));
      uint32_t *rstart = plan->as<uint32_t>(), *rend = rstart + P, *cursor = rend + P;
      uint64_t *cst = (uint64_t *)(plan->as<uint32_t>() + plan_words);
      {
        ProfScope ps(ctx, "rp_sample_hist");
        // an eighth of every tile, or less of it when a bucket still gets >= 4096 sampled rows (+-5 % at three sigma against
        // the regions' 12.5 % head room): C4's 2e8 rows read 50 MB instead of 200 (0.054 -> see DESIGN.md)
        uint32_t sdiv = 8;
        while (sdiv < 32 && (int64_t)P * 4096 * (2 * sdiv) <= n) sdiv *= 2;
        const int64_t samples = (int64_t)tiles1c * (RP_TILE / sdiv);
        // (four blocks per CU: every block ends with up to P global atomics — 4096 blocks spent more time on those than
        //  on reading the sample)
        const unsigned sblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(samples, 256 * 4 * 2), 4 * (int64_t)ctx->num_cus));
        rp_sample_hist_kernel<<<dim3(sblocks), dim3(256), 0, ctx->stream>>>(in.keys, in.filter, n, (uint32_t)RP_TILE, tiles1c, P, kp, sdiv,
                                                                           est->as<uint32_t>());
        rp_region_plan_kernel<<<dim3(1), dim3(512), 0, ctx->stream>>>(est->as<uint32_t>(), P, n, slack, B, rstart, rend, cursor);
        SQ_HIP(hipGetLastError());
      }
      // slim form (rp_claim_scatter_slim_kernel): 12 bytes per row out of the level; SQLRS_RP_SLIM=0 (read per call) = the
      // 16-byte form (tests, A/B)
      const char *cslim_e = hook("SQLRS_RP_SLIM");
      const bool claim_slim = nv == 1 && kp.rbits + SLIM_LOCAL_BITS + 7 <= 32 && !(cslim_e && std::atoi(cslim_e) == 0);
      Cols cc;
      PartitionedRows::Slim csl;
      if (claim_slim) {
        uint32_t log_b = 0;
        while ((1u << log_b) < B) log_b++;
#ifdef SLIM_AOS
        csl.buf0 = ctx->alloc(sizeof(SlimRec) * (size_t)pool_rows);
        csl.rows.rec = csl.buf0->as<SlimRec>();
#else
        csl.buf0 = ctx->alloc(8 * (size_t)pool_rows);
        csl.buf1 = ctx->alloc(4 * (size_t)pool_rows);
        csl.rows.v = csl.buf0->as<uint64_t>();
        csl.rows.w = csl.buf1->as<uint32_t>();
#endif
        csl.blk_bt = ctx->alloc(4 * (size_t)((slots_max >> log_b) + 2));
        csl.log_b = log_b;
        csl.tile = (uint32_t)RP_TILE;
        csl.on = true;
      } else if (use_rec) cc.rec = ctx->alloc(16 * (size_t)pool_rows);
      else {
        cc.k = ctx->alloc(8 * (size_t)pool_rows);
        cc.v0 = nv >= 1 ? ctx->alloc(8 * (size_t)pool_rows) : nullptr;
      }
      ClaimOut co;
      co.key = cc.k ? cc.k->as<uint64_t>() : nullptr;
      co.v0 = cc.v0 ? cc.v0->as<uint64_t>() : nullptr;
      co.rec = cc.rec ? (u64x2 *)cc.rec->p : nullptr;
      co.cursor = cursor;
      co.rend = rend;
      co.kept = (unsigned long long *)cst;
      co.flag = (unsigned int *)(cst + 1);
      co.B = B;
      const int psrc = !in.filter.col ? -1 : ((nv >= 1 && (const void *)in.filter.col == in.vals[0]) ? 1 : 3);
      const size_t clds = (size_t)RP_TILE * 8 * (1 + nv) + (size_t)WG * (4 + 4 + 8 + 8);
      {
        ProfScope ps(ctx, in.filter.col ? "rp_claim_scatter_filter" : "rp_claim_scatter");
        const uint64_t *k = in.keys, *a0 = (const uint64_t *)in.vals[0];
        const int64_t sink = (int64_t)slots_max;
#define SQ_CL1(NV, PS, RC)                                                                                          \
  do {                                                                                                              \
    auto kfn = rp_claim_scatter_kernel<NV, 512, 12, PS, RC>;                                                        \
    allow_big_lds(ctx, kfn);                                                                                        \
    kfn<<<dim3(wgs), dim3(512), clds, ctx->stream>>>(k, a0, in.filter, n, co, P, tiles1c, tpw, sink, kp);           \
  } while (0)
#define SQ_CL(NV, RC)                                                                                               \
  do {                                                                                                              \
    if (psrc < 0) SQ_CL1(NV, -1, RC);                                                                               \
    else if (psrc == 1 && NV >= 1) SQ_CL1(NV, (NV >= 1 ? 1 : 3), RC);                                               \
    else SQ_CL1(NV, 3, RC);                                                                                         \
  } while (0)
        if (claim_slim) {
          SlimClaimOut so;
          so.rows = csl.rows;
          so.cursor = cursor;
          so.rend = rend;
          so.blk_bt = csl.blk_bt->as<uint32_t>();
          so.flag = co.flag;
          so.kept = co.kept;
          so.B = B;
          so.log_b = csl.log_b;
          const char *sd_e = hook("SQLRS_RP_SLIM_DELTA"); // test hook, read per call: blocks abandoned after fewer tiles
          so.max_delta = sd_e ? (uint32_t)std::max(1, std::min(std::atoi(sd_e), 127)) : 127u;
          const size_t slds = (size_t)RP_TILE * (8 + 4 + 2) + (size_t)WG * (4 + 4 + 8 + 8 + 4);
          // 768-thread workgroups (twelve waves on the same 6144-row tile, 159 VGPRs, no spill) by default: C4 scatter 1.49 -> 1.43 ms,
          // with the WHERE fused 1.31 -> 1.14 (one process, three rounds); SQLRS_RP_CLAIM_WG=512 (read per call) = the eight-wave form
          const char *cwg_e = hook("SQLRS_RP_CLAIM_WG");
          const bool wg768 = psrc != 3 && !(cwg_e && std::atoi(cwg_e) == 512); // (own predicate column: 4 spilled registers at 768)
          const size_t slds768 = (size_t)RP_TILE * (8 + 4 + 2) + (size_t)768 * (4 + 4 + 8 + 8 + 4);
#define SQ_CS(PS)                                                                                                   \
  do {                                                                                                              \
    if (wg768) {                                                                                                    \
      auto kfn = rp_claim_scatter_slim_kernel<768, 8, PS>;                                                          \
      allow_big_lds(ctx, kfn);                                                                                      \
      kfn<<<dim3(wgs), dim3(768), slds768, ctx->stream>>>(k, a0, in.filter, n, so, P, tiles1c, tpw, sink, kp);      \
    } else {                                                                                                        \
      auto kfn = rp_claim_scatter_slim_kernel<512, 12, PS>;                                                         \
      allow_big_lds(ctx, kfn);                                                                                      \
      kfn<<<dim3(wgs), dim3(512), slds, ctx->stream>>>(k, a0, in.filter, n, so, P, tiles1c, tpw, sink, kp);         \
    }                                                                                                               \
  } while (0)
          if (psrc < 0) SQ_CS(-1);
          else if (psrc == 1) SQ_CS(1);
          else SQ_CS(3);
#undef SQ_CS
        } else if (nv == 0) SQ_CL(0, false);
        else if (use_rec) SQ_CL(1, true);
        else SQ_CL(1, false);
#undef SQ_CL
#undef SQ_CL1
        SQ_HIP(hipGetLastError());
      }
      // one round trip: region starts, cursors (= region fill) and the two counters
      const uint32_t *hp = (const uint32_t *)ctx->fetch(plan->p, plan_bytes); // (pinned staging buffer: <= 6.2 KiB)
      uint64_t hc[2];
      std::memcpy(hc, hp + plan_words, 16);
      if (!(uint32_t)hc[1]) {
        out->n = (int64_t)hc[0];
        out->P = P;
        publish(cc);
        if (claim_slim) out->slim = csl;
        out->bstart = plan; // (region starts; device consumers of contiguous buckets never see a claimed partition)
        out->bstart_host.assign(hp, hp + P);
        out->bstart_host.push_back((uint32_t)slots_max);
        out->bend_host.assign(hp + 2 * (size_t)P, hp + 3 * (size_t)P);
        return true;
      }
      // a region overflowed (the sample misjudged a bucket): the counting level below redoes the batch
    }
  }
  if (in.filter.col && p2_bits == 0) return false; // a single level evaluates a row filter only in its claimed form

  // ---- chunked first level (no histogram pass, optional fused row filter): two-level partitions of
  // batches large enough that the slack of the arenas (workgroups x (digits + 1) chunks) is a fraction of the input
  static const int chunk_env = [] { // test / tuning hook: 1 = whenever two levels are needed, 0 = never
    const char *e = hook("SQLRS_RP_CHUNKED");
    return e ? std::atoi(e) : -1;
  }();
  // (3072-row tiles with two workgroups per CU were measured slower for this level too: 6.4 vs 5.7 ms; so were 1024-thread
  //  workgroups with 8 rows per thread over the same 8192-row tiles — twice the waves per CU, but 128 VGPRs and 32 spilled:
  //  6.65 vs 5.68 ms, round 4)
  const uint32_t tiles1 = (uint32_t)ceil_div(n, RP_TILE);
  uint32_t cwgs = std::min<uint32_t>(tiles1, (uint32_t)ctx->num_cus);
  if (const char *wg_e = hook("SQLRS_RP_CHUNK_WGS")) // test hook, read per call: fewer workgroups = longer tile ranges per workgroup
    cwgs = std::max(1u, std::min<uint32_t>(cwgs, (uint32_t)std::atoi(wg_e)));
  const uint32_t ctpw = (uint32_t)ceil_div(tiles1, std::max(cwgs, 1u));
  cwgs = (uint32_t)ceil_div(tiles1, std::max(ctpw, 1u));
  const uint64_t spare_chunks = (uint64_t)cwgs * (d1 + 1); // chunks that may stay partly filled or unused
  static const int ct_env = [] { // tuning hook: tiles per chunk (1, 4, 8, 16 measured alike: 1 = smallest reservation)
    const char *e = hook("SQLRS_RP_CHUNK_TILES");
    return e ? std::max(1, std::min(64, std::atoi(e))) : 1;
  }();
  const uint64_t CAP = (uint64_t)RP_TILE * (uint64_t)ct_env;
  bool chunked = p2_bits != 0 && !flags && chunk_env != 0 && d1 <= (uint32_t)WG &&
                 (chunk_env == 1 || spare_chunks * CAP <= (uint64_t)n); // (the slack of the arenas is a fraction of the input)
  if (in.filter.col && !chunked) return false; // only the chunked first level evaluates a row filter
  if (chunked) {
    // Slim records (12 instead of 16 bytes per row through level 1, level 2 and the bucket pass; see the section
    // "slim records" above): dense packed rows with one value column whose chunk histograms fit LDS, bucket tables of
    // <= 4096 slots.  SQLRS_RP_SLIM=0 (read per call) keeps the 16-byte form (in-process A/B, tests).
    const char *slim_e = hook("SQLRS_RP_SLIM");
    const bool slim_on = pack && kp.dense && nv == 1 && (ROWS == 12 || ROWS == 16) && ct_env == 1 && (size_t)P * 4 <= 24 * 1024 &&
                         kp.rbits + SLIM_LOCAL_BITS + 7 <= 32 && kp.rbits + p2_bits + SLIM_LOCAL_BITS <= 32 &&
                         !(slim_e && std::atoi(slim_e) == 0) && !(hook("SQLRS_RP_H2") && std::atoi(hook("SQLRS_RP_H2")) == 0);
    // arena mode: a workgroup fills at most ceil(its rows / CAP) chunks completely and leaves <= d1 partly filled
    // (slim: + the chunks closed early because their next run would be more than SLIM_RUNS - 1 tiles after their first)
    const char *sd_e = hook("SQLRS_RP_SLIM_DELTA"); // test hook, read per call: early closes at test sizes
    const uint32_t slim_delta = sd_e ? (uint32_t)std::max(1, std::min<int>(std::atoi(sd_e), (int)SLIM_RUNS - 1)) : SLIM_RUNS - 1;
    const uint64_t arena = (uint64_t)ceil_div((int64_t)ctpw, (int64_t)ct_env) + d1 + 1 +
                           (slim_on ? (uint64_t)d1 * (uint64_t)ceil_div((int64_t)ctpw, (int64_t)slim_delta + 1) : 0);
    const uint64_t max_chunks = arena * cwgs;
    if (max_chunks * (CAP + RP_CHUNK_SKEW) + WG * (uint64_t)cwgs > 0xffffffffull) { // Tile::start is 64-bit, rows index u32 math
      if (in.filter.col) return false;
      chunked = false;
    }
    if (chunked && slim_on) {
      const size_t pool_rows = (size_t)max_chunks * (CAP + RP_CHUNK_SKEW) + (size_t)1024 * cwgs; // + one sink per workgroup (of up to 1024 threads)
      const uint32_t digits2 = 1u << p2_bits;
      auto slim_alloc = [&](size_t rows, BufP &b0, BufP &b1) {
        SlimRowsView v;
#ifdef SLIM_AOS
        b0 = ctx->alloc(sizeof(SlimRec) * rows);
        v.rec = b0->as<SlimRec>();
#else
        b0 = ctx->alloc(8 * rows);
        b1 = ctx->alloc(4 * rows);
        v.v = b0->as<uint64_t>();
        v.w = b1->as<uint32_t>();
#endif
        return v;
      };
      BufP cb0, cb1;
      BufP clen = ctx->alloc_zero(4 * (size_t)max_chunks);
      BufP cdig = ctx->alloc(4 * (size_t)max_chunks), cbase = ctx->alloc(4 * (size_t)max_chunks);
      BufP cstart = ctx->alloc(2 * (size_t)max_chunks * SLIM_RUNS);
      SQ_HIP(hipMemsetAsync(cstart->p, 0xff, 2 * (size_t)max_chunks * SLIM_RUNS, ctx->stream));
      BufP ctr = ctx->alloc_zero(8);
      BufP chist = ctx->alloc(4 * (size_t)max_chunks * digits2);
      SlimChunkOut so;
      so.rows = slim_alloc(pool_rows, cb0, cb1);
      so.chunk_len = clen->as<uint32_t>();
      so.chunk_dig = cdig->as<uint32_t>();
      so.chunk_base = cbase->as<uint32_t>();
      so.cstart = cstart->as<uint16_t>();
      so.counter = ctr->as<unsigned int>();
      so.max_chunks = (uint32_t)max_chunks;
      so.cap = (uint32_t)CAP;
      so.arena = (uint32_t)arena;
      so.hist = chist->as<uint32_t>();
      so.kshift = kp.rbits + p2_bits;
      so.max_delta = slim_delta;
      {
        const char *cc_e = hook("SQLRS_RP_CONC"); // tuning / test hook, read per call: 1 .. 8 eighths of a tile's row slots, 9 = never
        so.conc_eighths = cc_e ? (uint32_t)std::max(1, std::min(std::atoi(cc_e), 9)) : 3u;
      }
      const int psrc = !in.filter.col ? -1 : ((const void *)in.filter.col == in.vals[0] ? 1 : 3);
      const size_t clds = (size_t)RP_TILE * 14 + (size_t)WG * (4 + 4 + 8 + 8 + 4 + 4) + (size_t)P * 4;
      {
        ProfScope ps(ctx, in.filter.col ? "rp_chunk_scatter_filter" : "rp_chunk_scatter");
        const int64_t sink = (int64_t)max_chunks * (CAP + RP_CHUNK_SKEW);
#define SQ_SL1(R, PS)                                                                                                \
  do {                                                                                                              \
    auto kfn = rp_chunk_scatter_slim_kernel<512, R, PS>;                                                            \
    allow_big_lds(ctx, kfn);                                                                                        \
    kfn<<<dim3(cwgs), dim3(512), clds, ctx->stream>>>(in.keys, (const uint64_t *)in.vals[0], in.filter, n, so, P, p2_bits, d1, \
                                                      tiles1, ctpw, sink, kp);                                      \
  } while (0)
#define SQ_SL(R) do { if (psrc < 0) SQ_SL1(R, -1); else if (psrc == 1) SQ_SL1(R, 1); else SQ_SL1(R, 3); } while (0)
        const char *l1wg_e = hook("SQLRS_RP_L1_WG"); // A/B hook, read per call: 512 = the eight-wave form, 1024 x 6 rows
        const int l1wg = l1wg_e ? std::atoi(l1wg_e) : 768;
        if (ROWS == 12 && psrc != 3 && l1wg == 1024) {
          const size_t clds1k = (size_t)RP_TILE * 14 + (size_t)1024 * (4 + 4 + 8 + 8 + 4 + 4) + (size_t)P * 4;
          if (psrc < 0) {
            auto kfn = rp_chunk_scatter_slim_kernel<1024, 6, -1>;
            allow_big_lds(ctx, kfn);
            kfn<<<dim3(cwgs), dim3(1024), clds1k, ctx->stream>>>(in.keys, (const uint64_t *)in.vals[0], in.filter, n, so, P, p2_bits, d1, tiles1, ctpw, sink, kp);
          } else {
            auto kfn = rp_chunk_scatter_slim_kernel<1024, 6, 1>;
            allow_big_lds(ctx, kfn);
            kfn<<<dim3(cwgs), dim3(1024), clds1k, ctx->stream>>>(in.keys, (const uint64_t *)in.vals[0], in.filter, n, so, P, p2_bits, d1, tiles1, ctpw, sink, kp);
          }
        } else if (ROWS == 12 && psrc != 3 && l1wg == 768) {
          const size_t clds768 = (size_t)RP_TILE * 14 + (size_t)768 * (4 + 4 + 8 + 8 + 4 + 4) + (size_t)P * 4;
          if (psrc < 0) {
            auto kfn = rp_chunk_scatter_slim_kernel<768, 8, -1>;
            allow_big_lds(ctx, kfn);
            kfn<<<dim3(cwgs), dim3(768), clds768, ctx->stream>>>(in.keys, (const uint64_t *)in.vals[0], in.filter, n, so, P, p2_bits, d1, tiles1, ctpw, sink, kp);
          } else {
            auto kfn = rp_chunk_scatter_slim_kernel<768, 8, 1>;
            allow_big_lds(ctx, kfn);
            kfn<<<dim3(cwgs), dim3(768), clds768, ctx->stream>>>(in.keys, (const uint64_t *)in.vals[0], in.filter, n, so, P, p2_bits, d1, tiles1, ctpw, sink, kp);
          }
        } else if (ROWS == 16) SQ_SL(16); else SQ_SL(12);
#undef SQ_SL
#undef SQ_SL1
        SQ_HIP(hipGetLastError());
      }
      // level-2 geometry from the chunk table, on the device (as below)
      Level L2;
      L2.nseg = d1;
      const LevelLayout lay(d1);
      L2.dev = ctx->alloc(lay.total);
      bind_level(L2, lay);
      L2.tiles = ctx->alloc(sizeof(Tile) * (size_t)max_chunks);
      BufP totals = ctx->alloc(24);
      BufP tile_chunk = ctx->alloc(4 * (size_t)max_chunks);
      BufP plan = ctx->alloc(sizeof(ChunkPlan));
      SQ_HIP(hipMemsetAsync(plan->p, 0, sizeof(ChunkPlan), ctx->stream));
      const unsigned pblocks = (unsigned)std::min<uint64_t>(ceil_div((int64_t)max_chunks, 256 * 8), 128);
      rp_chunk_count_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(so.chunk_len, so.chunk_dig, so.counter, so.max_chunks,
                                                                        so.max_chunks, (uint32_t)RP_TILE, plan->as<ChunkPlan>());
      rp_chunk_prefix_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(
          plan->as<ChunkPlan>(), so.counter, d1, digits2, (int64_t *)L2.d_seg_start, (int64_t *)L2.d_seg_mat,
          (uint32_t *)L2.d_seg_tiles, (uint32_t *)L2.d_seg_tile_base, totals->as<uint64_t>());
      rp_chunk_assign_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(
          so.chunk_len, so.chunk_dig, so.counter, so.max_chunks, so.max_chunks, digits2, (uint32_t)CAP, (uint32_t)RP_TILE,
          L2.d_seg_tiles, L2.d_seg_tile_base, plan->as<ChunkPlan>(), (Tile *)L2.tiles->p, tile_chunk->as<uint32_t>());
      SQ_HIP(hipGetLastError());
      const uint64_t *ht = (const uint64_t *)ctx->fetch(totals->p, 24);
      const uint64_t ntiles = ht[0], kept = ht[1], overflow = ht[2];
      if (overflow) return false; // (impossible by the chunk bound; never trusted blindly)
      L2.num_tiles = (uint32_t)ntiles;
      L2.mat_entries = (int64_t)ntiles * digits2;
      out->n = (int64_t)kept;
      out->P = P;
      if (kept == 0) { // nothing passed the filter: the caller sees an empty partition
        out->bstart_host.assign((size_t)P + 1, 0u);
        out->bstart = nullptr;
        return true;
      }
      const size_t np = (size_t)kept + 1024; // + the sink rows (workgroups of up to 1024 threads)
      PartitionedRows::Slim &sl = out->slim;
      sl = PartitionedRows::Slim();
      sl.rows = slim_alloc(np, sl.buf0, sl.buf1);
      SlimLaunch sln;
      sln.in.rows = so.rows;
      sln.in.tile_chunk = tile_chunk->as<uint32_t>();
      sln.in.cstart = so.cstart;
      sln.out.rows = sl.rows;
      sln.kshift = so.kshift;
      sln.rbits = kp.rbits;
      BufP offs2;
      BufP premat = ctx->alloc(4 * (size_t)L2.mat_entries);
      rp_hist_from_chunks_kernel<<<dim3((unsigned)ceil_div(L2.mat_entries, 256)), dim3(256), 0, ctx->stream>>>(
          so.hist, tile_chunk->as<uint32_t>(), L2.mat_entries, digits2, premat->as<uint32_t>());
      SQ_HIP(hipGetLastError());
      EarlyStarts es;
      early_starts_for(L2, digits2, (int64_t)kept, es);
      exec_level(2, digits2, L2, RpIn(), RpOut(), (int64_t)kept, &offs2, premat, &sln, &es.queue);
      // the non-empty runs of every bucket with the base tile of the chunk they came from
      sl.nzstart = ctx->alloc(4 * (size_t)L2.mat_entries);
      sl.nzbt = ctx->alloc(4 * (size_t)L2.mat_entries);
      sl.nzcount = ctx->alloc(4 * (size_t)P);
      sl.bcol = ctx->alloc(4 * (size_t)P);
      ProfScope ps_runs(ctx, "rp_slim_runs");
      rp_slim_runs_kernel<<<dim3(P), dim3(256), 0, ctx->stream>>>(
          offs2->as<uint32_t>(), L2.mat_entries, sln.total->as<uint64_t>(), L2.d_seg_mat, L2.d_seg_tiles, L2.d_seg_tile_base,
          tile_chunk->as<uint32_t>(), so.chunk_base, digits2, sl.nzstart->as<uint32_t>(), sl.nzbt->as<uint32_t>(),
          sl.nzcount->as<uint32_t>(), sl.bcol->as<uint32_t>());
      SQ_HIP(hipGetLastError());
      if (hook("SQLRS_RP_TRACE")) // placement experiments (tools/placement_log.py): where the big buffers sit
        std::fprintf(stderr, "[rp_slim] keys %p vals %p chunks %p %p rows %p %p kept %llu tiles %u\n", (const void *)in.keys, in.vals[0],
                     cb0 ? cb0->p : nullptr, cb1 ? cb1->p : nullptr, sl.buf0 ? sl.buf0->p : nullptr, sl.buf1 ? sl.buf1->p : nullptr,
                     (unsigned long long)kept, L2.num_tiles);
      sl.tile = (uint32_t)RP_TILE;
      sl.on = true;
      out->key = out->v0 = out->v1 = out->idx = out->flags = out->rec = nullptr;
      out->bstart = finish_starts(es, L2, offs2, digits2, (int64_t)kept, &out->bstart_host);
      return true;
    }
    if (chunked) {
      const size_t pool_rows = (size_t)max_chunks * (CAP + RP_CHUNK_SKEW) + (size_t)1024 * cwgs; // + one sink per workgroup (of up to 1024 threads)
      BufP ck = staggered(8 * pool_rows, 0), c0 = nv >= 1 ? staggered(8 * pool_rows, 1) : nullptr,
           c1 = nv >= 2 ? staggered(8 * pool_rows, 2) : nullptr, ci = pack ? nullptr : staggered(4 * pool_rows, 3);
      BufP clen = ctx->alloc_zero(4 * (size_t)max_chunks);
      BufP cdig = ctx->alloc(4 * (size_t)max_chunks);
      BufP ctr = ctx->alloc_zero(8);
      ChunkOut co;
      co.arena = (uint32_t)arena;
      co.key = ck->as<uint64_t>();
      co.v0 = c0 ? c0->as<uint64_t>() : nullptr;
      co.v1 = c1 ? c1->as<uint64_t>() : nullptr;
      co.idx = ci ? ci->as<uint32_t>() : nullptr;
      co.chunk_len = clen->as<uint32_t>();
      co.chunk_dig = cdig->as<uint32_t>();
      co.counter = ctr->as<unsigned int>();
      co.base_chunks = (uint32_t)max_chunks;
      co.max_chunks = (uint32_t)max_chunks;
      co.cap = (uint32_t)CAP;
      const int psrc = !in.filter.col ? -1 : ((nv >= 1 && (const void *)in.filter.col == in.vals[0]) ? 1 : 3);
      // chunk histograms of the next level counted by this kernel (H2): every bucket needs a 4-byte counter in LDS
      const char *h2_e = hook("SQLRS_RP_H2"); // A/B hook, read per call: 0 = level 2 runs its own histogram pass
      // (round 6: unpacked rows of hashed partitions too, on the 512-thread tile — its 22-byte staging rows leave 16 KiB: P <= ~3000
      //  buckets, C4 over general keys; 65 536 buckets — the sparse-key C5 — keep the histogram pass.  SQLRS_RP_H2_UNPACKED=0: off)
      const char *h2u_e = hook("SQLRS_RP_H2_UNPACKED");
      const bool h2_unpacked = !pack && nv == 1 && ROWS == 12 && psrc != 3 && !(h2u_e && std::atoi(h2u_e) == 0) &&
                               (size_t)RP_TILE * (8 * 2 + 4 + 2) + (size_t)WG * (4 + 4 + 8 + 8) + (size_t)WG * 8 + (size_t)P * 4 <= 159 * 1024;
      const bool h2 = (pack || h2_unpacked) && nv == 1 && (ROWS == 12 || ROWS == 16) && ct_env == 1 && (size_t)P * 4 <= 24 * 1024 && !(h2_e && std::atoi(h2_e) == 0);
      BufP chist = h2 ? ctx->alloc(4 * (size_t)max_chunks * ((size_t)1 << p2_bits)) : nullptr;
      co.hist = chist ? chist->as<uint32_t>() : nullptr;
      const size_t clds = (size_t)RP_TILE * (pack ? 8 * (1 + nv) : 8 * (1 + nv) + 4 + 2) + (size_t)WG * (4 + 4 + 8 + 8) +
                          (h2 ? (size_t)WG * 8 + (size_t)P * 4 : 0);
      {
        ProfScope ps(ctx, in.filter.col ? "rp_chunk_scatter_filter" : "rp_chunk_scatter");
        const uint64_t *k = in.keys, *a0 = (const uint64_t *)in.vals[0], *a1 = (const uint64_t *)in.vals[1];
        const int64_t sink = (int64_t)max_chunks * (CAP + RP_CHUNK_SKEW);
#define SQ_CS1(NV, R, PK, PS)                                                                                       \
  do {                                                                                                              \
    auto kfn = rp_chunk_scatter_kernel<NV, 512, R, PK, PS>;                                                         \
    if (NV == 1 && PK && (R == 12 || R == 16) && h2) kfn = rp_chunk_scatter_kernel<NV, 512, R, PK, PS, (NV == 1 && PK && (R == 12 || R == 16))>; \
    allow_big_lds(ctx, kfn);                                                                                        \
    kfn<<<dim3(cwgs), dim3(512), clds, ctx->stream>>>(k, a0, a1, in.filter, n, co, P, p2_bits, d1, tiles1, ctpw,   \
                                                      sink, kp);                                                    \
  } while (0)
#define SQ_CS(NV, R, PK)                                                                                            \
  do {                                                                                                              \
    if (psrc < 0) SQ_CS1(NV, R, PK, -1);                                                                            \
    else if (psrc == 1 && NV >= 1) SQ_CS1(NV, R, PK, (NV >= 1 ? 1 : 3));                                            \
    else SQ_CS1(NV, R, PK, 3);                                                                                      \
  } while (0)
        if (nv == 2) SQ_CS(2, 8, false);
        else if (ROWS == 16) SQ_CS(1, 16, true);
        else if (ROWS != 12) { chunked = false; } // tuning shapes (SQLRS_RP_ROWS) keep the counting first level
        else if (nv == 0) { if (pack) SQ_CS(0, 12, true); else SQ_CS(0, 12, false); }
        // (768 threads x 8 rows for the unpacked rows of hashed partitions too: sparse-key C5 level 1 6.91 -> 6.81 ms in one process;
        //  SQLRS_RP_L1G_WG=512, read per call, = the eight-wave form; a predicate on a column of its own keeps it: 14 spilled registers)
        else if (!pack && h2) { // unpacked rows + the next level's counts: the 512-thread tile (the 768-thread one has no room for them)
#define SQ_CH(PS)                                                                                                   \
  do {                                                                                                              \
    auto kfn = rp_chunk_scatter_kernel<1, 512, 12, false, PS, true>;                                                \
    allow_big_lds(ctx, kfn, 159 * 1024);                                                                            \
    kfn<<<dim3(cwgs), dim3(512), clds, ctx->stream>>>(k, a0, a1, in.filter, n, co, P, p2_bits, d1, tiles1, ctpw, sink, kp); \
  } while (0)
          if (psrc < 0) SQ_CH(-1); else SQ_CH(1);
#undef SQ_CH
        }
        else if (!pack && psrc != 3 && !(hook("SQLRS_RP_L1G_WG") && std::atoi(hook("SQLRS_RP_L1G_WG")) == 512)) {
          const size_t clds768 = (size_t)RP_TILE * (8 * (1 + nv) + 4 + 2) + (size_t)768 * (4 + 4 + 8 + 8) + (clds - ((size_t)RP_TILE * (8 * (1 + nv) + 4 + 2) + (size_t)WG * (4 + 4 + 8 + 8)));
#define SQ_CG(PS)                                                                                                   \
  do {                                                                                                              \
    auto kfn = rp_chunk_scatter_kernel<1, 768, 8, false, PS>;                                                       \
    allow_big_lds(ctx, kfn);                                                                                        \
    kfn<<<dim3(cwgs), dim3(768), clds768, ctx->stream>>>(k, a0, a1, in.filter, n, co, P, p2_bits, d1, tiles1, ctpw, sink, kp); \
  } while (0)
          if (psrc < 0) SQ_CG(-1); else if (psrc == 1) SQ_CG(1); else SQ_CG(3);
#undef SQ_CG
        }
        else { if (pack) SQ_CS(1, 12, true); else SQ_CS(1, 12, false); }
#undef SQ_CS
#undef SQ_CS1
        SQ_HIP(hipGetLastError());
      }
      if (!chunked && in.filter.col) return false;
      if (chunked) {
        // level-2 geometry from the chunk table, on the device; the host only needs three numbers
        const uint32_t digits2 = 1u << p2_bits;
        Level L2;
        L2.nseg = d1;
        const LevelLayout lay(d1);
        L2.dev = ctx->alloc(lay.total);
        bind_level(L2, lay);
        L2.tiles = ctx->alloc(sizeof(Tile) * (size_t)max_chunks * (size_t)ct_env);
        BufP totals = ctx->alloc(24);
        BufP tile_chunk = h2 ? ctx->alloc(4 * (size_t)max_chunks) : nullptr;
        BufP plan = ctx->alloc(sizeof(ChunkPlan));
        SQ_HIP(hipMemsetAsync(plan->p, 0, sizeof(ChunkPlan), ctx->stream));
        const unsigned pblocks = (unsigned)std::min<uint64_t>(ceil_div((int64_t)max_chunks, 256 * 8), 128);
        rp_chunk_count_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(co.chunk_len, co.chunk_dig, co.counter, co.base_chunks,
                                                                          co.max_chunks, (uint32_t)RP_TILE, plan->as<ChunkPlan>());
        rp_chunk_prefix_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(
            plan->as<ChunkPlan>(), co.counter, d1, digits2, (int64_t *)L2.d_seg_start, (int64_t *)L2.d_seg_mat,
            (uint32_t *)L2.d_seg_tiles, (uint32_t *)L2.d_seg_tile_base, totals->as<uint64_t>());
        rp_chunk_assign_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(
            co.chunk_len, co.chunk_dig, co.counter, co.base_chunks, co.max_chunks, digits2, (uint32_t)CAP, (uint32_t)RP_TILE,
            L2.d_seg_tiles, L2.d_seg_tile_base, plan->as<ChunkPlan>(), (Tile *)L2.tiles->p,
            tile_chunk ? tile_chunk->as<uint32_t>() : nullptr);
        SQ_HIP(hipGetLastError());
        const uint64_t *ht = (const uint64_t *)ctx->fetch(totals->p, 24);
        const uint64_t ntiles = ht[0], kept = ht[1], overflow = ht[2];
        if (overflow) return false; // (impossible by the chunk bound; never trusted blindly)
        L2.num_tiles = (uint32_t)ntiles;
        L2.mat_entries = (int64_t)ntiles * digits2;
        out->n = (int64_t)kept;
        out->P = P;
        RpIn rin2;
        rin2.key = ck->as<uint64_t>();
        rin2.v0 = co.v0;
        rin2.v1 = co.v1;
        rin2.idx = co.idx;
        rin2.flags = nullptr;
        rin2.key_validity = rin2.v0_validity = rin2.v1_validity = nullptr;
        Cols c2;
        RpOut rout2;
        alloc_cols((int64_t)kept, true, c2, rout2);
        BufP offs2, premat;
        if (h2 && L2.mat_entries) { // the level's count matrix is already known: no histogram pass over the chunks
          premat = ctx->alloc(4 * (size_t)L2.mat_entries);
          rp_hist_from_chunks_kernel<<<dim3((unsigned)ceil_div(L2.mat_entries, 256)), dim3(256), 0, ctx->stream>>>(
              co.hist, tile_chunk->as<uint32_t>(), L2.mat_entries, digits2, premat->as<uint32_t>());
          SQ_HIP(hipGetLastError());
        }
        EarlyStarts es;
        early_starts_for(L2, digits2, (int64_t)kept, es);
        exec_level(2, digits2, L2, rin2, rout2, (int64_t)kept, &offs2, premat, nullptr, &es.queue);
        publish(c2);
        out->bstart = finish_starts(es, L2, offs2, digits2, (int64_t)kept, &out->bstart_host);
        return true;
      }
    }
  }

  // ---- level 1
  Cols c1;
  RpOut rout;
  alloc_cols(n, p2_bits == 0, c1, rout);
  RpIn rin;
  rin.key = in.keys;
  rin.v0 = (const uint64_t *)in.vals[0];
  rin.v1 = (const uint64_t *)in.vals[1];
  rin.idx = nullptr;
  rin.flags = nullptr;
  rin.key_validity = in.key_validity;
  rin.v0_validity = in.val_validity[0];
  rin.v1_validity = in.val_validity[1];
  BufP offs1;
  Level L1;
  run_level(1, d1, {0, n}, rin, rout, &offs1, &L1);
  std::vector<uint32_t> hs;
  BufP bs1 = bucket_starts(L1, offs1, d1, n, &hs);
  out->n = n;
  out->P = P;
  if (p2_bits == 0) {
    publish(c1);
    out->bstart = bs1;
    out->bstart_host = std::move(hs);
    return true;
  }
  // ---- level 2: every level-1 bucket is one segment
  std::vector<int64_t> seg(hs.begin(), hs.end());
  Cols c2;
  RpOut rout2;
  alloc_cols(n, true, c2, rout2);
  RpIn rin2;
  rin2.key = rout.key;
  rin2.v0 = rout.v0;
  rin2.v1 = rout.v1;
  rin2.idx = rout.idx;
  rin2.flags = rout.flags;
  rin2.key_validity = rin2.v0_validity = rin2.v1_validity = nullptr;
  BufP offs2;
  Level L2;
  run_level(2, 1u << p2_bits, seg, rin2, rout2, &offs2, &L2);
  publish(c2);
  out->bstart = bucket_starts(L2, offs2, 1u << p2_bits, n, &out->bstart_host);
  return true;
}

} // namespace sq
