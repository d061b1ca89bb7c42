// partition.hip — hash partitioning of a batch for the multi-GPU exchange.
// p(key) = mulhi(mix64(key ^ C), G): a mixer independent of the hash-table mixer so that
// partition and bucket choice are uncorrelated.  One stable 8-bit radix pass on the partition
// id gives the permutation (row order kept inside a partition), then every column is gathered.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

__device__ __forceinline__ uint32_t part_of(uint64_t key, uint32_t parts) {
  uint64_t h = mix64(key ^ 0x5851f42d4c957f2dULL);
  return (uint32_t)(((h >> 32) * (uint64_t)parts) >> 32);
}

__global__ __launch_bounds__(BLOCK) void part_ids_kernel(const uint64_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ validity,
                                                         int64_t n, uint32_t parts,
                                                         uint64_t *__restrict__ pid,
                                                         unsigned long long *__restrict__ counts) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    bool valid = !validity || ((validity[i >> 6] >> (i & 63)) & 1);
    uint32_t p = valid ? part_of(keys[i], parts) : 0u;
    pid[i] = p;
    atomicAdd(&h[p], 1u);
  }
  __syncthreads();
  if (threadIdx.x < parts && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

} // namespace sq

using namespace sq;

extern "C" int sqlrs_hash_partition(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, const sqlrs_expr_t *key,
                                    int num_parts, int out_mem, sqlrs_batch_t **out, int64_t *offsets) {
  return guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    if (num_parts < 1 || num_parts > 256) fail(SQLRS_ERR_INTERNAL, "num_parts must be in [1, 256]");
    Expr e = expr_from_abi(key);
    InBatch ib(ctx, in);
    auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
    int64_t n = ib.rows();
    if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "partition: more than 2^32 rows");
    std::vector<DCol> kc{eval_expr(ctx, e, colfn, n, true)};
    NKeys nk = normalize_keys(ctx, kc, n);
    int64_t n1 = std::max<int64_t>(n, 1);
    BufP pid = ctx->alloc(8 * (size_t)n1), perm = ctx->alloc(4 * (size_t)n1);
    BufP counts = ctx->alloc_zero(8 * 256);
    if (n) {
      ProfScope ps(ctx, "hash_partition");
      unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, BLOCK), 4096);
      part_ids_kernel<<<dim3(blocks), dim3(BLOCK), 0, ctx->stream>>>(
          nk.keys->as<uint64_t>(), nk.validity, n, (uint32_t)num_parts, pid->as<uint64_t>(),
          counts->as<unsigned long long>());
      SQ_HIP(hipGetLastError());
      iota_u32(ctx, perm->as<uint32_t>(), n);
      if (num_parts > 1) radix_sort_pairs(ctx, pid->as<uint64_t>(), perm->as<uint32_t>(), n, 0, 8);
    }
    std::vector<uint64_t> hc(256);
    SQ_HIP(hipMemcpyAsync(hc.data(), counts->p, 8 * 256, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    offsets[0] = 0;
    for (int p = 0; p < num_parts; p++) offsets[p + 1] = offsets[p] + (int64_t)hc[(size_t)p];
    DBatch o;
    o.rows = n;
    for (int c = 0; c < ib.num_columns(); c++)
      o.cols.push_back(gather_column(ctx, ib.col(c), perm->p, false, nullptr, n));
    *out = emit_batch(ctx, std::move(o), out_mem);
  });
}
