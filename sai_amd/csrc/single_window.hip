// sai_single_window: the whole of UStatistic.compute / QStatistic.compute for ONE window in one
// call (entry 4): fused site pass over the window's tiled blocks, window
// statistics over [0, n_sites), results copied into the caller's host buffers.

#include "common.hpp"

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int grow(sai_ctx* ctx, size_t dev_bytes, size_t host_bytes) {
  if (dev_bytes > ctx->sw_dev_cap) {
    if (ctx->sw_dev) SAI_HIP(hipFree(ctx->sw_dev));
    ctx->sw_dev = nullptr;
    ctx->sw_dev_cap = 0;
    const size_t want = dev_bytes + dev_bytes / 2;
    SAI_HIP(hipMalloc(&ctx->sw_dev, want));
    ctx->sw_dev_cap = want;
  }
  if (host_bytes > ctx->sw_host_cap) {
    if (ctx->sw_host) SAI_HIP(hipHostFree(ctx->sw_host));
    ctx->sw_host = nullptr;
    ctx->sw_host_cap = 0;
    const size_t want = host_bytes + host_bytes / 2;
    SAI_HIP(hipHostMalloc(&ctx->sw_host, want, hipHostMallocDefault));
    ctx->sw_host_cap = want;
  }
  return SAI_OK;
}

}  // namespace

extern "C" int sai_single_window(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops,
                                 const sai_params* set_host, sai_window_record* record_host, int32_t* cdd_u_host,
                                 int32_t* cdd_q_host, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (!set_host || !record_host) return fail(SAI_ERR_ARG, "NULL argument");
  if (n_sites > 0 && (!cdd_u_host || !cdd_q_host)) return fail(SAI_ERR_ARG, "NULL candidate buffer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_sites == 0) {  // q_statistic.py:96-98: nothing selected
    record_host->n_sites = record_host->u_count = record_host->n_cond = record_host->n_cdd_q = 0;
    record_host->q = std::numeric_limits<double>::quiet_NaN();
    return SAI_OK;
  }
  const size_t n = static_cast<size_t>(n_sites);
  // device scratch: tgt_freq | flag planes | lo,hi | record | offsets | totals | cdd_u | cdd_q
  size_t o = 0;
  const size_t o_freq = o;   o = align_up(o + n * sizeof(double), 256);
  const size_t o_flags = o;  o = align_up(o + static_cast<size_t>(sai_plane_words(n_sites, 1)) * sizeof(uint64_t), 256);
  const size_t o_lohi = o;   o = align_up(o + 2 * sizeof(int32_t), 256);
  const size_t o_head = o;   // record (24) | offsets (16) | totals (16 + scratch): one copy brings them back
  // record | offsets (2) | totals (2) + the prefix sum's scratch pair for the one record
  const size_t head_bytes = sizeof(sai_window_record) + 2 * sizeof(int64_t) + static_cast<size_t>(sai_window_total_words(1, 1)) * sizeof(int64_t);
  o = align_up(o + head_bytes, 256);
  const size_t o_u = o;      o = align_up(o + n * sizeof(int32_t), 256);
  const size_t o_q = o;      o = align_up(o + n * sizeof(int32_t), 256);
  if (int rc = grow(ctx, o, head_bytes + 2 * n * sizeof(int32_t))) return rc;
  char* d = static_cast<char*>(ctx->sw_dev);
  char* h = static_cast<char*>(ctx->sw_host);
  double* tgt_freq = reinterpret_cast<double*>(d + o_freq);
  uint64_t* planes = reinterpret_cast<uint64_t*>(d + o_flags);
  int32_t* lohi = reinterpret_cast<int32_t*>(d + o_lohi);
  sai_window_record* rec = reinterpret_cast<sai_window_record*>(d + o_head);
  int64_t* off = reinterpret_cast<int64_t*>(d + o_head + sizeof(sai_window_record));
  int64_t* totals = off + 2;
  int32_t* cdd_u = reinterpret_cast<int32_t*>(d + o_u);
  int32_t* cdd_q = reinterpret_cast<int32_t*>(d + o_q);

  const int32_t bounds[2] = {0, static_cast<int32_t>(n_sites)};
  SAI_HIP(hipMemcpyAsync(lohi, bounds, sizeof(bounds), hipMemcpyHostToDevice, st));
  if (int rc = sai_site_pass(ctx, n_sites, n_pops, pops, nullptr, 1, set_host, SAI_FREQ_CANDIDATES, tgt_freq, planes,
                             SAI_PLANES_PER_SET, st))
    return rc;
  if (int rc = sai_window_stats(ctx, n_sites, tgt_freq, planes, SAI_PLANES_PER_SET, 1, set_host, 1, lohi, lohi + 1, nullptr, rec, off, cdd_u,
                                n_sites, cdd_q, n_sites, totals, st))
    return rc;
  // results through the pinned mirror: head first (its counts size the two list copies)
  SAI_HIP(hipMemcpyAsync(h, d + o_head, head_bytes, hipMemcpyDeviceToHost, st));
  SAI_HIP(hipStreamSynchronize(st));
  std::memcpy(record_host, h, sizeof(sai_window_record));
  const int64_t n_u = record_host->u_count, n_q = record_host->n_cdd_q;
  if (n_u < 0 || n_u > n_sites || n_q < 0 || n_q > n_sites) return fail(SAI_ERR_HIP, "window record out of range");
  char* hu = h + head_bytes;
  char* hq = hu + n * sizeof(int32_t);
  if (n_u) SAI_HIP(hipMemcpyAsync(hu, cdd_u, static_cast<size_t>(n_u) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (n_q) SAI_HIP(hipMemcpyAsync(hq, cdd_q, static_cast<size_t>(n_q) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (n_u || n_q) SAI_HIP(hipStreamSynchronize(st));
  if (n_u) std::memcpy(cdd_u_host, hu, static_cast<size_t>(n_u) * sizeof(int32_t));
  if (n_q) std::memcpy(cdd_q_host, hq, static_cast<size_t>(n_q) * sizeof(int32_t));
  return SAI_OK;
}
