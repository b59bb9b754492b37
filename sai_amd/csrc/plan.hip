// Prepared launch sequences ("plans"): the arguments of a resident block's site pass and windows
// stage are fixed from step to step, so they are marshalled ONCE and a step is then a single call
// from the host language per sequence.  A plan replays the library's own entry points with stored
// copies of their arguments -- same validation, same kernels -- on the stream it is run on.
//
// Why: for a small job (C2: 0.4 GB, 73 us of site pass) the nine ctypes calls of a step cost the
// host 60-75 us -- more than the GPU needs -- and every call rebuilt its parameter-set array.

#include <functional>
#include <memory>
#include <vector>

#include "common.hpp"

struct sai_plan {
  sai_ctx* ctx;
  std::vector<std::function<int(void*)>> ops;  // each enqueues on the stream it is handed
  hipEvent_t pass_start = nullptr, pass_stop = nullptr;  // carried by the plan's site pass (sai_plan_set_pass_events)
};

namespace {

template <typename F>
int guarded_add(sai_plan* plan, F&& make) {
  if (!plan) return fail(SAI_ERR_ARG, "plan is NULL");
  try {
    plan->ops.emplace_back(make());
    return SAI_OK;
  } catch (const std::bad_alloc&) {
    return fail(SAI_ERR_HIP, "out of host memory");
  }
}

std::vector<sai_params> copy_sets(int32_t n_sets, const sai_params* sets) {
  return (n_sets > 0 && sets) ? std::vector<sai_params>(sets, sets + n_sets) : std::vector<sai_params>();
}

// After a site-pass call of a plan: the events the plan carries have been taken by the launch -- or, when the
// call returned without launching (no sites) or failed, they are still pending: they are then recorded on the
// stream as ordinary markers, so that a caller waiting for `stop` waits for the work before it on the stream
// instead of finding an event that was never recorded (which reads as done at once).
int finish_pass(sai_ctx* ctx, int rc, void* st) {
  hipEvent_t start = ctx->next_start, stop = ctx->next_stop;
  ctx->next_start = ctx->next_stop = nullptr;
  if (rc == SAI_OK) {
    if (start) SAI_HIP(hipEventRecord(start, static_cast<hipStream_t>(st)));
    if (stop) SAI_HIP(hipEventRecord(stop, static_cast<hipStream_t>(st)));
  }
  return rc;
}

std::vector<sai_pop> copy_pops(int32_t n_pops, const sai_pop* pops) {
  return (n_pops > 0 && pops) ? std::vector<sai_pop>(pops, pops + n_pops) : std::vector<sai_pop>();
}

}  // namespace

extern "C" {

int sai_plan_create(sai_ctx* ctx, sai_plan** plan_out) {
  if (int rc = enter(ctx)) return rc;
  if (!plan_out) return fail(SAI_ERR_ARG, "plan_out is NULL");
  sai_plan* p = new (std::nothrow) sai_plan{ctx, {}, nullptr, nullptr};
  if (!p) return fail(SAI_ERR_HIP, "out of host memory");
  *plan_out = p;
  return SAI_OK;
}

int sai_plan_destroy(sai_plan* plan) {
  delete plan;
  return SAI_OK;
}

int sai_plan_run(sai_plan* plan, void* stream) {
  if (!plan) return fail(SAI_ERR_ARG, "plan is NULL");
  for (auto& op : plan->ops)
    if (int rc = op(stream)) return rc;
  return SAI_OK;
}

int sai_plan_add_site_counts(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts) {
  return guarded_add(plan, [&] {
    auto pv = copy_pops(n_pops, pops);
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      ctx->next_start = plan->pass_start;
      ctx->next_stop = plan->pass_stop;
      const int rc = sai_site_counts(ctx, n_sites, n_pops, pv.data(), counts, st);
      return finish_pass(ctx, rc, st);
    };
  });
}

int sai_plan_add_site_pass(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                           int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                           uint64_t* planes, int64_t plane_stride, int32_t packed2) {
  return guarded_add(plan, [&] {
    auto pv = copy_pops(n_pops, pops);
    auto sv = copy_sets(n_sets, sets_host);
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      const sai_params* s = sv.empty() ? nullptr : sv.data();
      ctx->next_start = plan->pass_start;
      ctx->next_stop = plan->pass_stop;
      const int rc = packed2 ? sai_site_pass_packed2(ctx, n_sites, n_pops, pv.data(), counts, n_sets, s, freq_mode, tgt_freq,
                                                     planes, plane_stride, st)
                             : sai_site_pass(ctx, n_sites, n_pops, pv.data(), counts, n_sets, s, freq_mode, tgt_freq, planes,
                                             plane_stride, st);
      return finish_pass(ctx, rc, st);
    };
  });
}

int sai_plan_add_site_pass_dd(sai_plan* plan, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                              int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                              uint64_t* planes, int64_t plane_stride, const sai_dd_rows* dd) {
  if (!dd) return fail(SAI_ERR_ARG, "dd is NULL");
  return guarded_add(plan, [&] {
    auto pv = copy_pops(n_pops, pops);
    auto sv = copy_sets(n_sets, sets_host);
    const sai_dd_rows rows = *dd;
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      ctx->next_start = plan->pass_start;
      ctx->next_stop = plan->pass_stop;
      const int rc = sai_site_pass_dd(ctx, n_sites, n_pops, pv.data(), counts, n_sets, sv.empty() ? nullptr : sv.data(),
                                      freq_mode, tgt_freq, planes, plane_stride, &rows, st);
      return finish_pass(ctx, rc, st);
    };
  });
}

int sai_plan_add_site_flags(sai_plan* plan, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                            const uint32_t* counts, int32_t n_sets, const sai_params* sets_host, double* tgt_freq,
                            uint64_t* planes, int64_t plane_stride) {
  return guarded_add(plan, [&] {
    std::vector<int32_t> pl(ploidy_host && n_pops > 0 ? ploidy_host : nullptr,
                            ploidy_host && n_pops > 0 ? ploidy_host + n_pops : nullptr);
    auto sv = copy_sets(n_sets, sets_host);
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      return sai_site_flags(ctx, n_sites, n_pops, pl.empty() ? nullptr : pl.data(), counts, n_sets,
                            sv.empty() ? nullptr : sv.data(), tgt_freq, planes, plane_stride, nullptr, st);
    };
  });
}

int sai_plan_add_window_bounds(sai_plan* plan, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                               const int64_t* win_start, const int64_t* win_end, const int32_t* seg_lo,
                               const int32_t* seg_hi, int32_t* lo, int32_t* hi) {
  return guarded_add(plan, [&] {
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      return seg_lo ? sai_window_bounds_seg(ctx, pos, n_sites, n_windows, win_start, win_end, seg_lo, seg_hi, lo, hi, st)
                    : sai_window_bounds(ctx, pos, n_sites, n_windows, win_start, win_end, lo, hi, st);
    };
  });
}

int sai_plan_add_window_stats(sai_plan* plan, int64_t n_sites, const double* tgt_freq, const uint64_t* planes,
                              int64_t plane_stride, int32_t n_sets, const sai_params* sets_host, int32_t n_windows,
                              const int32_t* lo, const int32_t* hi, const int32_t* pos, sai_window_record* records,
                              int64_t* cdd_off, int32_t* cdd_u, int64_t cap_u, int32_t* cdd_q, int64_t cap_q,
                              int64_t* cdd_total) {
  return guarded_add(plan, [&] {
    auto sv = copy_sets(n_sets, sets_host);
    sai_ctx* ctx = plan->ctx;
    return [=](void* st) {
      return sai_window_stats(ctx, n_sites, tgt_freq, planes, plane_stride, n_sets, sv.empty() ? nullptr : sv.data(),
                              n_windows, lo, hi, pos, records, cdd_off, cdd_u, cap_u, cdd_q, cap_q, cdd_total, st);
    };
  });
}

int sai_plan_set_pass_events(sai_plan* plan, void* start_event, void* stop_event) {
  if (!plan) return fail(SAI_ERR_ARG, "plan is NULL");
  plan->pass_start = static_cast<hipEvent_t>(start_event);
  plan->pass_stop = static_cast<hipEvent_t>(stop_event);
  return SAI_OK;
}

int sai_plan_add_copy_to_host(sai_plan* plan, void* dst_host, const void* src, int64_t n_bytes) {
  if (n_bytes < 0 || (n_bytes > 0 && (!dst_host || !src))) return fail(SAI_ERR_ARG, "bad copy");
  return guarded_add(plan, [&] {
    return [=](void* st) {
      if (n_bytes == 0) return static_cast<int>(SAI_OK);
      SAI_HIP(hipMemcpyAsync(dst_host, src, static_cast<size_t>(n_bytes), hipMemcpyDeviceToHost, static_cast<hipStream_t>(st)));
      return static_cast<int>(SAI_OK);
    };
  });
}

}  // extern "C"
