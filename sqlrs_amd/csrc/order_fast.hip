// order_fast.hip — ORDER BY on ONE fixed-width key without NULLs, carrying one 8-byte column
// (order.rs:15-67 for the common `ORDER BY k` shape; everything else takes the general path of ops.hip).
//
// The general path is an LSD radix sort of (u64 key, u32 row id) pairs, 8 bits per pass over HBM, followed
// by one gather per column — and a random gather of 8-byte elements fetches a 128-byte line per element
// (12.8 GB for 1e8 rows).  Here the rows themselves travel, and only the TOP bits are sorted through HBM:
//
//   0. min / max of the keys' order-preserving image  -> off = image - min < 2^kbits (kbits <= 32)
//   1. <= 2 stable 8-bit multi-split passes over HBM on the TOP (kbits - rbits <= 16) bits of `off`; a row is
//      the word  off << 32 | row id  plus its carried column (16 B per row and pass; the first pass reads
//      the raw column and builds the word in registers)                                     [LSD order]
//      => rows are grouped by their top bits, in input order inside a group
//   2. group boundaries (a row whose predecessor has other top bits opens its group and closes the previous one)
//   3. one workgroup per group: the group (<= FIN_CAP rows) is sorted on the remaining rbits INSIDE LDS
//      (<= 2 stable 8-bit passes, same ballot ranking as the HBM passes), then key column, carried column
//      and row id (the permutation for any further column) leave in final order.
//
// HBM traffic for 1e8 rows, 31 key bits, one carried column: 0.8 (min/max) + 2 x (0.8 + 3.2) + 0.8
// (boundaries) + 3.6 = 13.2 GB, against 9.6 GB of sort passes + 12.8 GB of gather fetches before.
// Stability (ties in input order, like the general path) holds because every pass is stable.
// A group larger than FIN_CAP (heavily repeated keys with many low bits) sends the call through the same
// passes with rbits = 0 (all key bits sorted in HBM, <= 4 passes: no limit on a group there).
// Keys with MORE than 32 varying bits (random doubles, 63-bit ids) cannot ride in that word: `order_wide`
// below replaces the bits by splitters from a sorted sample and keeps the shape (two passes + in-LDS finish).
#include <atomic>
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

#include "order_kernels.hpp" // every kernel of the route (namespace sq)

namespace sq {

// the wide route; false = not taken (nothing produced that the caller may use).  `imin`, `range`: EXACT extremes of the
// key image.  want_perm: the row ids travel as the payload and `carry_out` stays empty (the caller gathers that column
// like the others)
static std::atomic<bool> g_order_lb_off{false}; // a look-back spin ran out once: the counting forms for the rest of the process
template <int KIND>
static bool order_wide(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, uint64_t imin, uint64_t range,
                       DCol *key_out, DCol *carry_out, BufP *perm_out, bool want_perm) {
  if (const char *e = hook("SQLRS_ORDER_WIDE")) // (A/B hook, read per call: 0 = the general path)
    if (e[0] == '0') return false;
  int top = 9;
  while (top < 16 && (n >> top) > 2048) top++;
  const uint32_t G = 1u << top, nk1 = G >> 8;
  int per_group = OWK_SAMPLES;
  if (const char *e = hook("SQLRS_ORDER_SAMPLES")) per_group = std::max(1, std::min(256, std::atoi(e))); // (A/B hook, read per call)
  const int64_t S = (int64_t)G * per_group;
  if (n < S) return false;
  const int kb = range ? 64 - __builtin_clzll(range) : 1;
  const bool pay_rows = want_perm, has_pay = pay_rows || carry != nullptr;
  // 0. splitters
  BufP ss = ctx->alloc(8 * (size_t)S), ssv = ctx->alloc(4 * (size_t)S), sub = ctx->alloc(8 * ((size_t)G + 1));
  BufP topfirst = ctx->alloc(4 * 256);
  {
    ProfScope ps(ctx, "order_knots");
    owk_sample_kernel<KIND><<<dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, ctx->stream>>>(key.values, n, desc, imin, S, n / S, ss->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    radix_sort_pairs(ctx, ss->as<uint64_t>(), ssv->as<uint32_t>(), S, 0, kb, true);
    owk_knots_kernel<<<dim3((unsigned)ceil_div((int64_t)G, 256)), dim3(256), 0, ctx->stream>>>(ss->as<uint64_t>(), G, (uint32_t)per_group, sub->as<uint64_t>());
    owk_topfirst_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(sub->as<uint64_t>(), G, nk1, topfirst->as<uint32_t>());
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *subp = sub->as<uint64_t>();
  const uint32_t *tfp = topfirst->as<uint32_t>();
  // 1. pass 1: top-level splitters, raw column -> (word, payload) columns
  const int64_t nblocks = ceil_div(n, OW_TILE), ntmax = nblocks + 256;
  // (with a payload both passes write {word, payload} records: a (tile, digit) run is one piece instead of one per column.
  //  SQLRS_ORDER_WIDE_REC1=0, read per call: pass 1 writes two columns)
  const char *rec1_e = hook("SQLRS_ORDER_WIDE_REC1");
  const bool rec1 = has_pay && !(rec1_e && rec1_e[0] == '0');
  BufP w1 = ctx->alloc((rec1 ? 16 : 8) * (size_t)n), p1 = has_pay && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr;
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks)), total = ctx->alloc(8);
  const uint64_t *psrc = (carry && !pay_rows) ? carry->v<uint64_t>() : nullptr;
  dim3 g1((unsigned)nblocks), g2((unsigned)ntmax), b(OW_WG);
  const char *two_e = hook("SQLRS_ORDER_TWO"); // (read per call: 0 = word and payload side by side in LDS, two workgroups per CU)
  const bool two = !(two_e && two_e[0] == '0');
  // The first pass in its look-back form (the narrow route's, see ow_scatter_kernel): its 256 segment sizes from one persistent
  // launch, the tiles chained — no count matrix, no scan.  The second pass keeps its counting form: its digit is a search in the
  // row's own segment's splitters, so nothing ahead of the first pass can count it.  SQLRS_ORDER_LB=0 (read per call) / a spin
  // that ran out: the counting form.
  static thread_local bool wide_lb_skip = false;
  const char *lb_e = hook("SQLRS_ORDER_LB"), *lbf_e = hook("SQLRS_ORDER_LB_TEST_FAIL");
  const bool lb1 = rec1 && two && n < (1ll << 30) && !g_order_lb_off.load() && !wide_lb_skip && !(lb_e && lb_e[0] == '0');
  BufP ghb1, lbdesc1;
  if (lb1) {
    ProfScope ps(ctx, "order_split");
    ghb1 = ctx->alloc(4 * 260);
    lbdesc1 = ctx->alloc(4 * 256 * (size_t)nblocks);
    SQ_HIP(hipMemsetAsync(ghb1->p, 0, 4 * 260, ctx->stream));
    SQ_HIP(hipMemsetAsync(lbdesc1->p, 0, 4 * 256 * (size_t)nblocks, ctx->stream));
    const unsigned gblocks = (unsigned)std::min<int64_t>(nblocks, 4 * (int64_t)ctx->num_cus);
    owk_ghist_kernel<KIND><<<dim3(gblocks), b, 0, ctx->stream>>>(key.values, n, desc, imin, nblocks, ghb1->as<uint32_t>(), subp, nk1, tfp);
    owk_scatter_kernel<KIND, 1, 1, true, false, true, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, nullptr,
                                                                                        w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp,
                                                                                        ghb1->as<uint32_t>(), lbdesc1->as<uint32_t>(),
                                                                                        ghb1->as<uint32_t>() + 256);
    SQ_HIP(hipGetLastError());
  } else {
    ProfScope ps(ctx, "order_split");
    owk_hist_kernel<KIND, 1><<<g1, b, 0, ctx->stream>>>(key.values, n, desc, imin, nblocks, hist->as<uint32_t>(), nullptr, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(), total->as<uint64_t>());
    if (rec1 && two)
      owk_scatter_kernel<KIND, 1, 1, true, false, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                                  w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    else if (rec1)
      owk_scatter_kernel<KIND, 1, 1, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                     w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    else if (has_pay)
      owk_scatter_kernel<KIND, 1, 1, false><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                      w1->as<uint64_t>(), p1->as<uint64_t>(), nullptr, subp, nk1, tfp);
    else
      owk_scatter_kernel<KIND, 1, 0, false><<<g1, b, 0, ctx->stream>>>(key.values, nullptr, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                      w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
  }
  // 2. pass 2 over segment-aligned tiles: the segment's own 256 splitters; records out when there is a payload
  BufP firsttile = ctx->alloc(4 * 257), segstart = ctx->alloc(8 * 257), tiles2 = ctx->alloc(sizeof(OwkTile) * (size_t)ntmax);
  BufP hist2 = ctx->alloc(4 * (size_t)(256 * ntmax)), offs2 = ctx->alloc(4 * (size_t)(256 * ntmax));
  BufP out2 = ctx->alloc((has_pay ? 16 : 8) * (size_t)n);
  {
    ProfScope ps(ctx, "order_split");
    if (lb1) ow_tile_plan_gh_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(ghb1->as<uint32_t>(), n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    else ow_tile_plan_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(offs->as<uint32_t>(), nblocks, n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    owk_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(firsttile->as<uint32_t>(), segstart->as<int64_t>(),
                                                                                             (uint32_t)ntmax, (OwkTile *)tiles2->p);
    const OwkTile *tp = (const OwkTile *)tiles2->p;
    if (rec1) owk_hist_kernel<KIND, 2, true><<<g2, b, 0, ctx->stream>>>(w1->p, n, desc, imin, ntmax, hist2->as<uint32_t>(), tp, subp, nk1, tfp);
    else owk_hist_kernel<KIND, 2><<<g2, b, 0, ctx->stream>>>(w1->p, n, desc, imin, ntmax, hist2->as<uint32_t>(), tp, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist2->as<uint32_t>(), 256 * ntmax, nullptr, offs2->as<uint32_t>(), total->as<uint64_t>());
    if (rec1 && two)
      owk_scatter_kernel<KIND, 2, 1, true, true, true><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                                 out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (rec1)
      owk_scatter_kernel<KIND, 2, 1, true, true><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                           out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (has_pay && two)
      owk_scatter_kernel<KIND, 2, 1, true, false, true><<<g2, b, 0, ctx->stream>>>(w1->p, p1->as<uint64_t>(), n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                                  out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (has_pay)
      owk_scatter_kernel<KIND, 2, 1, true><<<g2, b, 0, ctx->stream>>>(w1->p, p1->as<uint64_t>(), n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                     out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else
      owk_scatter_kernel<KIND, 2, 0, false><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                      out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
  }
  // 3. groups
  BufP gstart = ctx->alloc(4 * (size_t)65536), gend = ctx->alloc(4 * ((size_t)65536 + 4)); // [G]: largest group, [G + 1]: pure chunks, [G + 2]: look-back spin ran out
  BufP pure_items = ctx->alloc(8 * ((size_t)ceil_div(n, (int64_t)OWK_PURE_CHUNK) + G + 1));
  SQ_HIP(hipMemsetAsync(gend->as<uint32_t>() + G, 0, 12, ctx->stream));
  if (lb1) SQ_HIP(hipMemcpyAsync(gend->as<uint32_t>() + G + 2, ghb1->as<uint32_t>() + 256, 4, hipMemcpyDeviceToDevice, ctx->stream));
  {
    ProfScope ps(ctx, "order_groups");
    owk_group_table_kernel<<<dim3(nk1), dim3(256), 0, ctx->stream>>>(offs2->as<uint32_t>(), firsttile->as<uint32_t>(), segstart->as<int64_t>(), subp, G,
                                                                   gstart->as<uint32_t>(), gend->as<uint32_t>(), (uint2 *)pure_items->p);
    SQ_HIP(hipGetLastError());
  }
  const uint32_t *gh = (const uint32_t *)ctx->fetch(gend->as<uint32_t>() + G, 12);
  const uint32_t max_group = gh[0], pure_chunks = gh[1];
  if (lb1 && (gh[2] || (lbf_e && lbf_e[0] == '1'))) { // nothing of this attempt is valid: once more in the counting form
    if (gh[2]) {
      g_order_lb_off.store(true);
      ctx->order_lb_fallbacks++;
    }
    struct Skip {
      Skip() { wide_lb_skip = true; }
      ~Skip() { wide_lb_skip = false; }
    } skip;
    return order_wide<KIND>(ctx, key, desc, carry, n, imin, range, key_out, carry_out, perm_out, want_perm);
  }
  if (hook("SQLRS_ORDER_TRACE"))
    std::fprintf(stderr, "[order_wide] n=%lld key bits=%d groups=%u largest group to sort=%u rows, %u chunks of single-value groups\n",
                 (long long)n, kb, G, max_group, pure_chunks);
  if (max_group > FIN_CAP) return false; // thousands of distinct keys between two neighbouring samples: general path
  // 4. finish
  key_out->dtype = key.dtype;
  key_out->length = n;
  key_out->null_count = 0;
  key_out->own_values = ctx->alloc((KIND == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
  key_out->values = key_out->own_values->p;
  uint32_t *perm = nullptr;
  uint64_t *po = nullptr;
  if (pay_rows) {
    *perm_out = ctx->alloc(4 * (size_t)n);
    perm = (*perm_out)->as<uint32_t>();
  } else if (carry) {
    carry_out->dtype = carry->dtype;
    carry_out->length = n;
    carry_out->null_count = 0;
    carry_out->own_values = ctx->alloc(8 * (size_t)n + 16);
    carry_out->values = carry_out->own_values->p;
    po = carry_out->own_values->as<uint64_t>();
  }
  const char *fc_e = hook("SQLRS_ORDER_FINISH_COUNT"); // (A/B hook, read per call: 0 = LSD passes + neighbour walk only)
  const int fin_count = !(fc_e && fc_e[0] == '0');
  {
    ProfScope ps(ctx, "order_finish");
    // The splitters balance the groups only statistically (16 samples per group: sizes spread like a Gamma(16), the largest of
    // 65 536 is ~2.5x the mean), and the LDS a workgroup asks for decides how many are resident: the groups of up to
    // 2048 rows (9 in 10) go through a launch of their own with 37 KB each, those of up to 4096 through a second, the few
    // beyond (if any) through a third.
#define SQ_WFIN(NP, RR, ABOVE, UPTO)                                                                                 \
  do {                                                                                                               \
    auto kfn = owk_finish_kernel<KIND, NP, RR, NP == 1>;                                                             \
    const size_t lds = (size_t)RR * FIN_WG * 8 * (1 + NP) + 4 * (FIN_WAVES * 256 + 256);                             \
    if (lds > 64 * 1024) allow_big_lds(ctx, kfn);                                                                    \
    kfn<<<dim3(G), dim3(FIN_WG), lds, ctx->stream>>>(out2->as<uint64_t>(), nullptr, gstart->as<uint32_t>(), gend->as<uint32_t>(), subp, G, \
                                                     desc, imin, key_out->own_values->p, po, perm,         \
                                                     (uint32_t)(ABOVE), (uint32_t)(UPTO), fin_count);                \
  } while (0)
#define SQ_WFIN_R(NP)                                                                                                \
  do {                                                                                                               \
    SQ_WFIN(NP, 8, 0, 8 * FIN_WG);                                                                                   \
    if (max_group > 8 * FIN_WG) SQ_WFIN(NP, 16, 8 * FIN_WG, 16 * FIN_WG);                                            \
    if (max_group > 16 * FIN_WG) SQ_WFIN(NP, 24, 16 * FIN_WG, FIN_CAP);                                              \
  } while (0)
    if (has_pay) SQ_WFIN_R(1);
    else SQ_WFIN_R(0);
#undef SQ_WFIN_R
#undef SQ_WFIN
    if (pure_chunks) { // groups of ONE value (heavy hitters): copied out as they are
      const uint2 *items = (const uint2 *)pure_items->p;
      if (has_pay)
        owk_pure_copy_kernel<KIND, 1, true><<<dim3(pure_chunks), dim3(256), 0, ctx->stream>>>(
            out2->as<uint64_t>(), items, gstart->as<uint32_t>(), gend->as<uint32_t>(), desc, imin, key_out->own_values->p, po, perm);
      else
        owk_pure_copy_kernel<KIND, 0, false><<<dim3(pure_chunks), dim3(256), 0, ctx->stream>>>(
            out2->as<uint64_t>(), items, gstart->as<uint32_t>(), gend->as<uint32_t>(), desc, imin, key_out->own_values->p, po, perm);
    }
    SQ_HIP(hipGetLastError());
  }
  return true;
}

// `optimistic`: the key range comes from a SAMPLE (every 16th chunk of 2048 rows: 0.19 -> 0.03 ms for 1e8 rows), widened
// as far as the same number of key bits allows; the first split pass tests every key against it and *retry_exact is set
// (nothing produced, return false) when one lies outside — the caller runs the exact form once.
template <int KIND, int NPAY>
static bool order_fast_impl(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, DCol *key_out, DCol *carry_out,
                            BufP *perm_out, bool want_perm, bool optimistic, bool *retry_exact, bool *in_order,
                            bool hbm_only = false, bool *retry_hbm_only = nullptr) {
  // 0. key range
  BufP mm = ctx->alloc(16 * OW_MM_SLOTS + 16); // {min = ~0, max = 0} x OW_MM_SLOTS | out-of-range flag (u32), largest group (u32) | inversion seen (u32)
  constexpr int FLAG_W = 2 * OW_MM_SLOTS;       // index of the flag word (u64)
  unsigned int *inv = (unsigned int *)(mm->as<uint64_t>() + FLAG_W + 1); // (its upper half: the heavy-value probe's count)
  const char *hp_e = hook("SQLRS_ORDER_HEAVY_PROBE"); // (A/B hook, read per call: 0 = no probe)
  const bool heavy_probe = !hbm_only && !(hp_e && hp_e[0] == '0');
  {
    ProfScope ps(ctx, "order_minmax");
    order_minmax_init_kernel<<<dim3(1), dim3(128), 0, ctx->stream>>>(mm->as<unsigned long long>());
    if (in_order) order_inversion_kernel<KIND, true><<<dim3(256), dim3(256), 0, ctx->stream>>>(key.values, n, desc, inv);
    if (heavy_probe) order_heavy_probe_kernel<KIND><<<dim3(1), dim3(1024), 0, ctx->stream>>>(key.values, n, desc, inv + 1);
    const int every = optimistic ? 16 : 1;
    unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, (int64_t)256 * 8 * every), 8 * (int64_t)ctx->num_cus));
    order_minmax_kernel<KIND><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(key.values, n, desc, mm->as<unsigned long long>(), every);
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 16 * OW_MM_SLOTS + 16);
  const uint32_t heavy_cnt = (uint32_t)(h[FLAG_W + 1] >> 32);
  if (in_order && (uint32_t)h[FLAG_W + 1] == 0) { // no inversion among the sampled pairs: look at every pair
    ProfScope ps(ctx, "order_minmax");
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256 * 8), 8 * (int64_t)ctx->num_cus));
    order_inversion_kernel<KIND, false><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(key.values, n, desc, inv);
    SQ_HIP(hipGetLastError());
    if (ctx->fetch_value(inv) == 0) {
      *in_order = true; // the rows are in the requested order already: nothing to do
      return false;
    }
    h = (const uint64_t *)ctx->fetch(mm->p, 16 * OW_MM_SLOTS); // (the pinned staging buffer was reused)
  }
  uint64_t imin = ~0ull, imax = 0;
  for (int q = 0; q < OW_MM_SLOTS; q++) {
    imin = std::min(imin, h[2 * q]);
    imax = std::max(imax, h[2 * q + 1]);
  }
  uint64_t range = imax - imin;
  if (imin > imax) return false;
  if (range > 0xffffffffull) { // more than 32 varying key bits: splitters instead of bits (order_wide), or the general path
    if constexpr (KIND == OKIND_I32) return false;
    else {
      // (sampled extremes are not the extremes, and this route does not need them: offsets from 0 over the whole 64-bit
      //  range do — the first and the last group then span far more values than their rows use, which their workgroups
      //  notice as one long run of equal top bits and sort on all bits.  SQLRS_ORDER_WIDE_EXACT=1: the exact pass first)
      if (optimistic) {
        const char *ex = hook("SQLRS_ORDER_WIDE_EXACT");
        if (ex && ex[0] == '1') {
          *retry_exact = true;
          return false;
        }
        imin = 0;
        range = ~0ull;
      }
      return order_wide<KIND>(ctx, key, desc, carry, n, imin, range, key_out, carry_out, perm_out, want_perm);
    }
  }
  int kbits = 1;
  while (kbits < 32 && (1ull << kbits) <= range) kbits++;
  unsigned int *oob = nullptr;
  if (optimistic) {
    // The sample's extremes lie INSIDE the true range.  (a) When the sampled keys fit the 2^kbits-aligned window they start
    // in, that window is the guess (keys `x mod 2^k`, ids counted from 0: the true range is the window, and splitting the
    // few values the sample leaves free evenly between both ends misses one of them every other time); (b) otherwise one
    // more key bit, the sampled range in the middle of the window (the split plan changes by one bit, not the cost);
    // (c) no bit left: the exact pass.
    const uint64_t win = kbits < 64 ? (1ull << kbits) : 0, base = imin & ~(win - 1);
    if (imax - base < win) {
      imin = base;
    } else if (kbits < 32) {
      kbits++;
      const uint64_t slack = ((1ull << kbits) - 1) - range;
      imin -= std::min<uint64_t>(slack / 2, imin);
    } else {
      *retry_exact = true; // (nothing was launched beyond the sample)
      return false;
    }
    oob = (unsigned int *)(mm->as<uint64_t>() + FLAG_W);
  }
  // top <= 16 bits go through HBM — as many as leave groups of ~1-2 K rows for the in-LDS finish (2e6 rows: 10 bits;
  // with 16 the finish ran 65 536 workgroups of 30 rows each: 0.64 ms of its 1.27 ms)
  int want = 1;
  while (want < 16 && (n >> want) > 2048) want++;
  // (hbm_only: every key bit goes through the HBM passes, <= 4 of them, and the finish is a streaming unpack — the second
  //  try after a group turned out larger than the in-LDS finish takes: few distinct keys spread over many bits)
  const int top = hbm_only ? kbits : std::min(kbits, want), rbits = kbits - top;
  {
    // a value with a visible share of the rows: the splitter route gives it a group of its own that is copied, not sorted
    // (order_wide works on any range; if it declines, the plan below runs as before)
    // (only when low bits are left for the in-LDS finish: with every bit sorted in HBM no group is too large)
    if (heavy_probe && rbits > 0 && heavy_cnt >= OW_HEAVY_MIN &&
        order_wide<KIND>(ctx, key, desc, carry, n, optimistic ? 0 : imin, optimistic ? ~0ull : range, key_out, carry_out, perm_out, want_perm))
      return true;
  }
  // 1. stable multi-split passes on bits [32 + rbits, 32 + kbits) of the word, LSD order
  const int64_t nblocks = ceil_div(n, OW_TILE);
  // (the usual plan — two passes, one carried column, an in-LDS finish — moves {word, value} records through BOTH passes and
  //  needs none of the four column buffers; SQLRS_ORDER_REC1=0, read per call: records out of the last pass only)
  const char *tl_e0 = hook("SQLRS_ORDER_TILED"), *rec_e0 = hook("SQLRS_ORDER_REC"), *rec1_e = hook("SQLRS_ORDER_REC1");
  const bool rec1 = NPAY == 1 && rbits > 0 && top > 8 && top <= 16 && !(tl_e0 && std::atoi(tl_e0) == 0) && !(rec_e0 && std::atoi(rec_e0) == 0) &&
                    !(rec1_e && rec1_e[0] == '0');
  BufP wa = rec1 ? nullptr : ctx->alloc(8 * (size_t)n), wb = rec1 ? nullptr : ctx->alloc(8 * (size_t)n);
  BufP pa = NPAY && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr, pb = NPAY && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr;
  BufP recbuf1 = rec1 ? ctx->alloc(16 * (size_t)n) : nullptr;
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks)), total = ctx->alloc(8);
  const void *src = key.values;
  const uint64_t *psrc = NPAY ? carry->v<uint64_t>() : nullptr;
  uint64_t *wdst = rec1 ? nullptr : wa->as<uint64_t>(), *walt = rec1 ? nullptr : wb->as<uint64_t>();
  uint64_t *pdst = NPAY && !rec1 ? pa->as<uint64_t>() : nullptr, *palt = NPAY && !rec1 ? pb->as<uint64_t>() : nullptr;
  bool raw = true, rec_in = false;
  dim3 g((unsigned)nblocks), b(OW_WG);
  // The last of two HBM passes (rbits > 0: an in-LDS finish follows) runs over segment-aligned tiles, which makes
  // the group boundaries a by-product of its count matrix, and (one carried column) writes 16-byte records.
  // SQLRS_ORDER_TILED=0 / SQLRS_ORDER_REC=0 (read per call) keep the plain blocks / the column form for A/B runs.
  const char *tl_e = hook("SQLRS_ORDER_TILED"), *rec_e = hook("SQLRS_ORDER_REC");
  const bool use_tiled = rbits > 0 && !(tl_e && std::atoi(tl_e) == 0);
  const bool use_rec = use_tiled && NPAY == 1 && !(rec_e && std::atoi(rec_e) == 0);
  const int64_t ntmax = nblocks + 256; // tiles of the segment-aligned pass: at most one ragged tile per segment more
  BufP recbuf = use_rec ? ctx->alloc(16 * (size_t)n) : nullptr;
  BufP firsttile, segstart, tiles2, hist2, offs2;
  bool rec_form = false, tiled_done = false;
  auto one_pass = [&](int shift) {
    ProfScope ps(ctx, "order_split");
    const bool last = shift + 8 >= 32 + kbits;
    if (last && use_tiled && !raw) {
      firsttile = ctx->alloc(4 * 257);
      segstart = ctx->alloc(8 * 257);
      tiles2 = ctx->alloc(sizeof(OwTile) * (size_t)ntmax);
      hist2 = ctx->alloc(4 * (size_t)(256 * ntmax));
      offs2 = ctx->alloc(4 * (size_t)(256 * ntmax));
      ow_tile_plan_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(offs->as<uint32_t>(), nblocks, n, firsttile->as<uint32_t>(),
                                                                segstart->as<int64_t>());
      ow_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(
          firsttile->as<uint32_t>(), segstart->as<int64_t>(), (uint32_t)ntmax, (OwTile *)tiles2->p);
      dim3 g2((unsigned)ntmax);
      const OwTile *tp = (const OwTile *)tiles2->p;
      if (rec_in) ow_hist_kernel<KIND, false, true, true><<<g2, b, 0, ctx->stream>>>(src, n, desc, imin, shift, ntmax, hist2->as<uint32_t>(), tp);
      else ow_hist_kernel<KIND, false, true><<<g2, b, 0, ctx->stream>>>(src, n, desc, imin, shift, ntmax, hist2->as<uint32_t>(), tp);
      SQ_HIP(hipGetLastError());
      exclusive_scan_u32(ctx, hist2->as<uint32_t>(), 256 * ntmax, nullptr, offs2->as<uint32_t>(), total->as<uint64_t>());
      if (use_rec && rec_in) {
        ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1><<<g2, b, 0, ctx->stream>>>(
            src, nullptr, n, desc, imin, shift, ntmax, offs2->as<uint32_t>(), recbuf->as<uint64_t>(), nullptr, tp, oob);
        src = recbuf->p;
        psrc = nullptr;
        rec_form = true;
      } else if (use_rec) {
        ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1><<<g2, b, 0, ctx->stream>>>(
            src, psrc, n, desc, imin, shift, ntmax, offs2->as<uint32_t>(), recbuf->as<uint64_t>(), nullptr, tp, oob);
        src = recbuf->p;
        psrc = nullptr;
        rec_form = true;
      } else {
        ow_scatter_kernel<KIND, false, NPAY, true, false><<<g2, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, ntmax,
                                                                                    offs2->as<uint32_t>(), wdst, pdst, tp, oob);
        src = wdst;
        psrc = pdst;
      }
      SQ_HIP(hipGetLastError());
      tiled_done = true;
      return;
    }
    if (raw) ow_hist_kernel<KIND, true><<<g, b, 0, ctx->stream>>>(src, n, desc, imin, shift, nblocks, hist->as<uint32_t>(), nullptr, oob, kbits);
    else ow_hist_kernel<KIND, false><<<g, b, 0, ctx->stream>>>(src, n, desc, imin, shift, nblocks, hist->as<uint32_t>(), nullptr);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(), total->as<uint64_t>());
    if (raw && rec1) { // records out of the raw pass: the tiled pass behind it reads one 16-byte piece per row
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(),
                                                                                     recbuf1->as<uint64_t>(), nullptr, nullptr, oob);
      SQ_HIP(hipGetLastError());
      src = recbuf1->p;
      psrc = nullptr;
      raw = false;
      rec_in = true;
      return;
    }
    if (raw) ow_scatter_kernel<KIND, true, NPAY><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(), wdst, pdst, nullptr, oob);
    else ow_scatter_kernel<KIND, false, NPAY><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(), wdst, pdst, nullptr, oob);
    SQ_HIP(hipGetLastError());
    src = wdst;
    psrc = pdst;
    std::swap(wdst, walt);
    std::swap(pdst, palt);
    raw = false;
  };
  // The usual plan in its look-back form (round 5): ONE histogram of the column for both passes, the passes themselves
  // chained over their tiles — no count matrices, no scans (1e8 rows: 0.24 + 0.33 ms of histograms and 0.08 of scans
  // against 0.2 for the one histogram).  SQLRS_ORDER_LB=0 (read per call): the counting form; also taken for the rest of
  // the process once a look-back spin ran out (a predecessor tile that never showed up: see ow_scatter_kernel).
  const char *lb_e = hook("SQLRS_ORDER_LB");
  static thread_local bool lb_skip = false; // (set around the one re-run after a failed attempt)
  const char *lbf_e = hook("SQLRS_ORDER_LB_TEST_FAIL"); // (test hook, read per call: treat the attempt as failed)
  const bool two_pass = rbits > 0 && top > 8 && top <= 16 && use_tiled;
  const bool lb = two_pass && (NPAY == 1 ? (rec1 && use_rec) : true) && n < (1ll << 30) && !g_order_lb_off.load() && !lb_skip && !(lb_e && lb_e[0] == '0');
  BufP ghb, lbdesc, boundb;
  bool slim = false; // the look-back form moved 12-byte records (ow_scatter_kernel<.., SLIM>): the finish reads those
  unsigned int *lbw = nullptr; // {look-back spin ran out, largest group, key outside the optimistic range}
  bool lb_done = false;
  if (lb) {
    ProfScope ps(ctx, "order_split");
    ghb = ctx->alloc(4 * 520);
    lbdesc = ctx->alloc(4 * 256 * (size_t)(nblocks + ntmax));
    boundb = ctx->alloc(4 * 256 * 256);
    SQ_HIP(hipMemsetAsync(ghb->p, 0, 4 * 520, ctx->stream));
    SQ_HIP(hipMemsetAsync(lbdesc->p, 0, 4 * 256 * (size_t)(nblocks + ntmax), ctx->stream));
    uint32_t *gh = ghb->as<uint32_t>();
    lbw = gh + 512;
    unsigned int *oob_lb = oob ? lbw + 2 : nullptr;
    const int s1 = 32 + rbits, s2 = s1 + 8;
    const unsigned gblocks = (unsigned)std::min<int64_t>(nblocks, 4 * (int64_t)ctx->num_cus);
    ow_ghist_kernel<KIND><<<dim3(gblocks), b, 0, ctx->stream>>>(src, n, desc, imin, s1, s2, nblocks, gh, oob_lb, kbits);
    uint64_t *out1 = NPAY == 1 ? recbuf1->as<uint64_t>() : wdst, *out2 = NPAY == 1 ? recbuf->as<uint64_t>() : walt; // (records / words)
    const char *two_e = hook("SQLRS_ORDER_TWO"); // (read per call: 0 = word and value side by side in LDS, two workgroups per CU)
    const bool two = NPAY == 1 && !(two_e && two_e[0] == '0');
    const char *slim_e = hook("SQLRS_ORDER_SLIM"); // (A/B hook, read per call: 0 = 16-byte records {word, value} between the passes)
    slim = two && !want_perm && !(slim_e && slim_e[0] == '0'); // 12-byte records {key offset, value}: nobody reads the row id
    if (slim)
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true, NPAY == 1, NPAY == 1><<<g, b, 0, ctx->stream>>>(
          src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    else if (two)
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true, NPAY == 1><<<g, b, 0, ctx->stream>>>(
          src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    else
    ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true><<<g, b, 0, ctx->stream>>>(
        src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    if (lbf_e && lbf_e[0] == '2') { // test hook: as if a spin had run out in the first pass — its output is garbage, the flag is up
      SQ_HIP(hipMemsetAsync(out1, 0xff, (NPAY == 1 ? 16 : 8) * (size_t)n, ctx->stream));
      SQ_HIP(hipMemsetAsync(lbw, 1, 4, ctx->stream));
    }
    firsttile = ctx->alloc(4 * 257);
    segstart = ctx->alloc(8 * 257);
    tiles2 = ctx->alloc(sizeof(OwTile) * (size_t)ntmax);
    ow_tile_plan_gh_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(gh, n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    ow_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(
        firsttile->as<uint32_t>(), segstart->as<int64_t>(), (uint32_t)ntmax, (OwTile *)tiles2->p);
    if (slim)
      ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true, NPAY == 1, NPAY == 1><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
          out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
          lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    else if (two)
      ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true, NPAY == 1><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
          out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
          lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    else
    ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
        out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
        lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    SQ_HIP(hipGetLastError());
    src = out2;
    psrc = nullptr;
    rec_form = NPAY == 1;
    lb_done = true;
  } else
    for (int shift = 32 + rbits; shift < 32 + kbits || raw; shift += 8) one_pass(shift); // (>= 1 pass: the words must exist)
  const uint64_t *words = (const uint64_t *)src;
  const uint64_t *pays = psrc;
  // outputs
  key_out->dtype = key.dtype;
  key_out->length = n;
  key_out->null_count = 0;
  key_out->own_values = ctx->alloc((KIND == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
  key_out->values = key_out->own_values->p;
  if (want_perm) *perm_out = ctx->alloc(4 * (size_t)n);
  uint32_t *perm = want_perm ? (*perm_out)->as<uint32_t>() : nullptr;
  if (rbits == 0) {
    if (oob && ctx->fetch_value(oob)) {
      *retry_exact = true;
      return false;
    }
    ProfScope ps(ctx, "order_finish");
    ow_unpack_kernel<KIND, NPAY><<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(words, n, desc, imin,
                                                                                               key_out->own_values->p, perm);
    SQ_HIP(hipGetLastError());
    if (NPAY) { // the carried column is already in final order: adopt the buffer it sits in
      carry_out->dtype = carry->dtype;
      carry_out->length = n;
      carry_out->null_count = 0;
      carry_out->own_values = (pays == pa->as<uint64_t>()) ? pa : pb;
      carry_out->values = carry_out->own_values->p;
    }
    return true;
  }
  // 2. groups = distinct values of the top bits
  const uint32_t G = 1u << top;
  BufP gstart = ctx->alloc(4 * (size_t)G), gend = ctx->alloc(4 * ((size_t)G + 1)); // gend[G] = largest group
  if (!lb_done) { // (the look-back form's table kernel writes every entry, its largest group goes to lbw[1])
    SQ_HIP(hipMemsetAsync(gstart->p, 0xff, 4 * (size_t)G, ctx->stream));
    SQ_HIP(hipMemsetAsync(gend->p, 0, 4 * ((size_t)G + 1), ctx->stream));
  }
  {
    ProfScope ps(ctx, "order_groups");
    if (lb_done) {
      ow_group_table_lb_kernel<<<dim3(G >> 8), dim3(256), 0, ctx->stream>>>(boundb->as<uint32_t>(), ghb->as<uint32_t>(), gstart->as<uint32_t>(),
                                                                       gend->as<uint32_t>(), lbw);
    } else if (tiled_done) { // (two passes: 8 bits, then top - 8)
      ow_group_table_kernel<<<dim3(G >> 8), dim3(256), 0, ctx->stream>>>(offs2->as<uint32_t>(), ntmax, firsttile->as<uint32_t>(), n,
                                                                    gstart->as<uint32_t>(), gend->as<uint32_t>());
    } else {
      ow_group_bounds_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(n, 256 * 8), 16 * (int64_t)ctx->num_cus)), dim3(256), 0, ctx->stream>>>(
          words, n, 32 + rbits, gstart->as<uint32_t>(), gend->as<uint32_t>());
      ow_group_max_kernel<<<dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, ctx->stream>>>(gstart->as<uint32_t>(), gend->as<uint32_t>(), G,
                                                                                        gend->as<uint32_t>() + G);
    }
    SQ_HIP(hipGetLastError());
  }
  uint32_t max_group;
  if (lb_done) { // one round trip for the three
    const uint32_t *hv = (const uint32_t *)ctx->fetch(lbw, 12);
    if (hv[0] || (lbf_e && lbf_e[0] == '1')) { // a look-back spin ran out: nothing of this attempt is valid; the counting form from here on
      if (hv[0] && !(lbf_e && lbf_e[0] == '2')) { // (not for the test hook's forced failure)
        g_order_lb_off.store(true);
        ctx->order_lb_fallbacks++;
      }
      struct Skip {
        Skip() { lb_skip = true; }
        ~Skip() { lb_skip = false; }
      } skip;
      return order_fast_impl<KIND, NPAY>(ctx, key, desc, carry, n, key_out, carry_out, perm_out, want_perm, optimistic, retry_exact, nullptr,
                                         hbm_only, retry_hbm_only);
    }
    if (oob && hv[2]) {
      *retry_exact = true;
      return false;
    }
    max_group = hv[1];
  } else if (oob) { // one round trip for both: the largest group and the verdict on the optimistic key range
    SQ_HIP(hipMemcpyAsync(mm->as<uint32_t>() + 2 * FLAG_W + 1, gend->as<uint32_t>() + G, 4, hipMemcpyDeviceToDevice, ctx->stream));
    const uint32_t *hv = (const uint32_t *)ctx->fetch(mm->as<uint64_t>() + FLAG_W, 8);
    if (hv[0]) {
      *retry_exact = true;
      return false;
    }
    max_group = hv[1];
  } else
    max_group = ctx->fetch_value(gend->as<uint32_t>() + G);
  if (max_group > FIN_CAP) { // heavily repeated top bits: all bits through HBM passes instead (no limit on a group there)
    if (retry_hbm_only) *retry_hbm_only = true;
    return false;
  }
  if (NPAY) {
    carry_out->dtype = carry->dtype;
    carry_out->length = n;
    carry_out->null_count = 0;
    carry_out->own_values = ctx->alloc(8 * (size_t)n + 16);
    carry_out->values = carry_out->own_values->p;
  }
  {
    ProfScope ps(ctx, "order_finish");
    uint64_t *po = NPAY ? carry_out->own_values->as<uint64_t>() : nullptr;
    const char *fc_e = hook("SQLRS_ORDER_FINISH_COUNT"); // (A/B hook, read per call: 0 = the finish sorts on its low bits with LSD passes only)
    const int fin_count = !(fc_e && fc_e[0] == '0');
#define SQ_FIN(RR)                                                                                                   \
  do {                                                                                                               \
    auto kfn = ow_finish_kernel<KIND, NPAY, RR>;                                                                     \
    if (NPAY == 1 && rec_form) kfn = ow_finish_kernel<KIND, NPAY, RR, NPAY == 1>;                                    \
    if (NPAY == 1 && rec_form && slim) kfn = ow_finish_kernel<KIND, NPAY, RR, NPAY == 1, NPAY == 1>;                 \
    const size_t lds = (size_t)RR * FIN_WG * 8 * (1 + NPAY) + 4 * (FIN_WAVES * 256 + 256);                           \
    if (lds > 64 * 1024) allow_big_lds(ctx, kfn);                                                                    \
    kfn<<<dim3(G), dim3(FIN_WG), lds, ctx->stream>>>(words, pays, gstart->as<uint32_t>(), gend->as<uint32_t>(), rbits, desc, imin, \
                                                     key_out->own_values->p, po, perm, fin_count);                   \
  } while (0)
    if (max_group <= 8 * FIN_WG) SQ_FIN(8);
    else if (max_group <= 16 * FIN_WG) SQ_FIN(16);
    else SQ_FIN(24);
#undef SQ_FIN
    SQ_HIP(hipGetLastError());
  }
  return true;
}

// ORDER BY one key column (int64 / float64 / int32, no NULLs) of >= 2^20 rows, optionally carrying one
// 8-byte column without NULLs; `perm` (row ids in output order) is produced when asked for.  Returns false
// when the shape or the data do not fit (nothing has been produced then).
// `in_order` (optional, out): set when the rows are in the requested order already — the call then returns false having
// produced nothing, and the caller emits its input as it is
bool order_fast(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, DCol *key_out, DCol *carry_out,
                BufP *perm, bool want_perm, bool *in_order) {
  if (in_order) *in_order = false;
  if (n < (1 << 20) || n > 0xffffffffll || key.stride == 0 || (key.validity && key.null_count != 0)) return false;
  if (carry && (width_of(carry->dtype) != 8 || carry->stride == 0 || (carry->validity && carry->null_count != 0))) return false;
  // optimistic key range for large columns (SQLRS_ORDER_SAMPLE, read per call: 0 = always the exact pass, 1 = always sampled)
  const char *smp_e = hook("SQLRS_ORDER_SAMPLE");
  const bool optimistic = smp_e ? std::atoi(smp_e) != 0 : n >= (1ll << 24); // (1 = whatever the size: tests)
#define SQ_OF(K)                                                                                                     \
  do {                                                                                                               \
    bool retry = false, hbm = false;                                                                                 \
    bool ok = carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, optimistic, &retry, in_order, false, &hbm) \
                    : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, optimistic, &retry, in_order, false, &hbm); \
    if (ok || (!retry && !hbm)) return ok;                                                                           \
    if (retry) {                                                                                                     \
      retry = false;                                                                                                 \
      ok = carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, false, &hbm) \
                 : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, false, &hbm); \
      if (ok || !hbm) return ok;                                                                                     \
    }                                                                                                                \
    return carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, true) \
                 : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, true); \
  } while (0)
  switch (key.dtype) {
  case SQLRS_INT64: SQ_OF(OKIND_I64);
  case SQLRS_FLOAT64: SQ_OF(OKIND_F64);
  case SQLRS_INT32: SQ_OF(OKIND_I32);
  default: return false;
  }
#undef SQ_OF
}

// ==== several integer keys, nullable integer keys ====================================================================
// ORDER BY a, b [, c, d] over plain int64 / int32 columns (order.rs:27-66: lexsort over the sort columns, NULLs first
// whatever the direction): when the keys' ranges together need <= 64 bits, the rows are ordered by ONE composite key
//     field_c = [valid bit, only for a column with NULLs :] asc ? value - min_c : max_c - value      (a NULL: all zero)
//     composite = field_0 : field_1 : ...   (most significant first)
// through the single-key routes above (<= 32 bits: the narrow one; more: the splitter route), and the key columns of the
// result — values and validity — are decoded from the sorted composite instead of being gathered.  The general path runs
// one stable radix sort of (key, row id) pairs per key, last key first, and gathers every column.
struct CompKeys {
  const void *vals[4];
  const uint64_t *valid[4]; // nullptr = no NULLs in this key
  uint64_t imin[4], imax[4];
  int kind[4], bits[4], desc[4]; // bits: of the value part (the valid bit sits above it)
  int nk;
};
__device__ __forceinline__ uint64_t comp_image(const CompKeys &ck, int c, int64_t i) {
  return ck.kind[c] == OKIND_I64 ? order_image<OKIND_I64>(ck.vals[c], i, 0) : order_image<OKIND_I32>(ck.vals[c], i, 0);
}
__global__ __launch_bounds__(256) void comp_build_kernel(CompKeys ck, int64_t n, int64_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t comp = 0;
  for (int c = 0; c < ck.nk; c++) {
    const int fb = ck.bits[c] + (ck.valid[c] ? 1 : 0);
    if (fb == 0) continue; // (a constant column; a shift by 64 would also be undefined)
    uint64_t field = 0;
    if (!ck.valid[c] || ((ck.valid[c][i >> 6] >> (i & 63)) & 1ull)) {
      const uint64_t img = comp_image(ck, c, i);
      field = (ck.desc[c] ? ck.imax[c] - img : img - ck.imin[c]) | (ck.valid[c] ? 1ull << ck.bits[c] : 0ull);
    }
    comp = (fb < 64 ? comp << fb : 0) | field;
  }
  out[i] = (int64_t)(comp ^ (1ull << 63)); // (as int64 whose order-preserving image is the composite itself)
}
// one key column out of the sorted composite; out_valid (nullable keys): one word per wave
__global__ __launch_bounds__(256) void comp_decode_kernel(const int64_t *__restrict__ comp, int64_t n, int shift, int bits, int nullable,
                                                          int desc, uint64_t imin, uint64_t imax, int kind, void *__restrict__ out,
                                                          uint64_t *__restrict__ out_valid) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int fb = bits + nullable;
  bool valid = false;
  int64_t v = 0;
  if (i < n) {
    const uint64_t c = (uint64_t)comp[i] ^ (1ull << 63);
    const uint64_t field = fb == 0 ? 0 : ((c >> shift) & (fb < 64 ? (1ull << fb) - 1 : ~0ull));
    valid = !nullable || ((field >> bits) & 1ull);
    const uint64_t f = bits == 0 ? 0 : (field & (bits < 64 ? (1ull << bits) - 1 : ~0ull));
    if (valid) v = ordered_to_i64(desc ? imax - f : imin + f);
    if (kind == OKIND_I32) ((int32_t *)out)[i] = (int32_t)v;
    else ((int64_t *)out)[i] = v;
  }
  if (out_valid) {
    const uint64_t m = __ballot(valid);
    if (lane_id() == 0 && (i >> 6) < (n + 63) / 64) out_valid[i >> 6] = m;
  }
}

bool order_composite(Ctx *ctx, const std::vector<const DCol *> &keys, const std::vector<int> &desc, const DCol *carry, int64_t n,
                     std::vector<DCol> *keys_out, DCol *carry_out, BufP *perm, bool want_perm, bool *in_order) {
  if (in_order) *in_order = false;
  const int nk = (int)keys.size();
  if (nk < 1 || nk > 4 || n < (1 << 20) || n > 0xffffffffll) return false;
  if (const char *e = hook("SQLRS_ORDER_COMPOSITE")) // (A/B hook, read per call: 0 = the general path)
    if (e[0] == '0') return false;
  CompKeys ck{};
  ck.nk = nk;
  bool any_nullable = false;
  for (int c = 0; c < nk; c++) {
    const DCol &k = *keys[(size_t)c];
    if ((k.dtype != SQLRS_INT64 && k.dtype != SQLRS_INT32) || k.stride == 0) return false;
    ck.vals[c] = k.values;
    ck.valid[c] = (k.validity && k.null_count != 0) ? k.validity : nullptr;
    any_nullable |= ck.valid[c] != nullptr;
    ck.kind[c] = k.dtype == SQLRS_INT64 ? OKIND_I64 : OKIND_I32;
    ck.desc[c] = desc[(size_t)c];
  }
  if (nk == 1 && !any_nullable) return false; // (the single-key route has had its say)
  // ranges of the keys (over all rows: what sits under a NULL can only widen them): one pass per column, one round trip for all
  constexpr size_t MM_WORDS = 2 * OW_MM_SLOTS + 2;
  BufP mm = ctx->alloc(8 * MM_WORDS * (size_t)nk);
  {
    ProfScope ps(ctx, "order_minmax");
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, (int64_t)256 * 8), 8 * (int64_t)ctx->num_cus));
    for (int c = 0; c < nk; c++) {
      unsigned long long *m = mm->as<unsigned long long>() + MM_WORDS * (size_t)c;
      order_minmax_init_kernel<<<dim3(1), dim3(128), 0, ctx->stream>>>(m);
      if (ck.kind[c] == OKIND_I64) order_minmax_kernel<OKIND_I64><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(ck.vals[c], n, 0, m, 1);
      else order_minmax_kernel<OKIND_I32><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(ck.vals[c], n, 0, m, 1);
    }
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 8 * MM_WORDS * (size_t)nk);
  int total = 0;
  for (int c = 0; c < nk; c++) {
    uint64_t lo = ~0ull, hi = 0;
    for (int q = 0; q < OW_MM_SLOTS; q++) {
      lo = std::min(lo, h[MM_WORDS * (size_t)c + 2 * q]);
      hi = std::max(hi, h[MM_WORDS * (size_t)c + 2 * q + 1]);
    }
    if (lo > hi) return false;
    ck.imin[c] = lo;
    ck.imax[c] = hi;
    ck.bits[c] = hi == lo ? 0 : 64 - __builtin_clzll(hi - lo);
    total += ck.bits[c] + (ck.valid[c] ? 1 : 0);
  }
  if (total > 64) return false; // the composite does not fit one word: general path
  std::vector<int64_t> nulls((size_t)nk, 0);
  for (int c = 0; c < nk; c++)
    if (ck.valid[c]) nulls[(size_t)c] = count_nulls(ctx, *keys[(size_t)c]);
  DCol comp;
  comp.dtype = SQLRS_INT64;
  comp.length = n;
  comp.null_count = 0;
  comp.own_values = ctx->alloc(8 * (size_t)n + 16);
  comp.values = comp.own_values->p;
  {
    ProfScope ps(ctx, "order_keys");
    comp_build_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(ck, n, comp.own_values->as<int64_t>());
    SQ_HIP(hipGetLastError());
  }
  DCol sorted;
  if (!order_fast(ctx, comp, 0, carry, n, &sorted, carry_out, perm, want_perm, in_order)) return false;
  comp = DCol(); // (its buffer goes back to the pool before the outputs are allocated)
  {
    ProfScope ps(ctx, "order_keys");
    keys_out->clear();
    int shift = total;
    for (int c = 0; c < nk; c++) {
      const int nullable = ck.valid[c] ? 1 : 0;
      shift -= ck.bits[c] + nullable;
      DCol o;
      o.dtype = keys[(size_t)c]->dtype;
      o.length = n;
      o.null_count = nulls[(size_t)c];
      o.own_values = ctx->alloc((ck.kind[c] == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
      o.values = o.own_values->p;
      if (nullable) {
        o.own_validity = ctx->alloc(8 * (size_t)ceil_div(n, 64) + 16);
        o.validity = o.own_validity->as<uint64_t>();
      }
      comp_decode_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(
          (const int64_t *)sorted.values, n, shift, ck.bits[c], nullable, ck.desc[c], ck.imin[c], ck.imax[c], ck.kind[c], o.own_values->p,
          nullable ? o.own_validity->as<uint64_t>() : nullptr);
      keys_out->push_back(o);
    }
    SQ_HIP(hipGetLastError());
  }
  return true;
}

} // namespace sq
