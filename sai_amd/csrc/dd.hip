// DD.

#include "numpy_sum.hpp"
#include "stream_loops.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// DD (sai/stats/dd_statistic.py:60-77): mean city-block distance of the source individuals to
// the reference individuals minus that to the target individuals.  site_absdiff is the
// site_counts streaming loop with a different byte operation: for NS source individuals at a
// time, per site, sum over the population's individuals of |src - g| on the raw dosage values
// (missing calls enter as their negative numbers, exactly as scipy's cdist sees them).  Bytes are
// biased to unsigned (x ^ 0x80), widened to packed 16-bit fields, |a - b| = max(a - b, b - a) with
// v_pk_sub_i16 / v_pk_max_i16.  window_dd sums those per-site integers over each window (exact)
// and forms the means in numpy's order.
// ------------------------------------------------------------------------------------------

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t absdiff_u16x2(uint32_t x, uint32_t y) {
  const s16x2 d = __builtin_bit_cast(s16x2, x) - __builtin_bit_cast(s16x2, y);
  const s16x2 m = __builtin_elementwise_max(d, -d);
  return __builtin_bit_cast(uint32_t, m);
}

struct AbsArgs {
  int64_t n_sites;
  int64_t n_tiles;
  const int8_t* pop_tiles;
  int32_t n_ind;
  const int8_t* src_tiles;
  int32_t n_src_ind;
  int32_t a0;     // first source individual of this launch
  uint32_t* out;  // [n_src_ind][n_sites]
};

template <int NS>
__global__ __launch_bounds__(64) void site_absdiff_kernel(AbsArgs a) {
  const int lane = threadIdx.x;
  const int r = lane >> 2;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    // this lane's 16 sites of each source individual, biased and widened to 16-bit fields
    uint32_t sv_lo[NS][4], sv_hi[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const u32x4 v = *(reinterpret_cast<const u32x4*>(a.src_tiles + (tile * a.n_src_ind + a.a0 + k) * kTile) + (lane & 3));
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = w[j] ^ 0x80808080u;
        sv_lo[k][j] = u & 0x00FF00FFu;
        sv_hi[k][j] = (u >> 8) & 0x00FF00FFu;
      }
    }
    const u32x4* base = reinterpret_cast<const u32x4*>(a.pop_tiles + tile * static_cast<int64_t>(a.n_ind) * kTile) + lane;
    uint32_t sum32[NS][16];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) sum32[k][j] = 0;
    const int n_full = a.n_ind >> 4;         // iterations in which all 16 rows exist
    const int n_iter = (a.n_ind + 15) >> 4;  // plus at most one partial iteration
    int it = 0;
    while (it < n_iter) {
      const int full_end = min(n_full, it + kChunkIters);  // 255 * (248 + 4) < 2^16: the fields cannot overflow
      uint32_t acc_lo[NS][4], acc_hi[NS][4];
#pragma unroll
      for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_lo[k][j] = acc_hi[k][j] = 0;
      auto consume = [&](const u32x4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t u = w[j] ^ 0x80808080u;
          const uint32_t lo = u & 0x00FF00FFu, hi = (u >> 8) & 0x00FF00FFu;
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            acc_lo[k][j] += absdiff_u16x2(lo, sv_lo[k][j]);
            acc_hi[k][j] += absdiff_u16x2(hi, sv_hi[k][j]);
          }
        }
      };
      for (; it + kUnroll <= full_end; it += kUnroll) {
        u32x4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) consume(v[u]);
      }
      // tail of the population (as in accumulate_rows): one batch of clamped loads, rows that do
      // not exist are skipped after the loads have been issued
      const int last = (full_end == n_full) ? n_iter : full_end;
      if (it < last) {
        u32x4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int row = min(it + u, last - 1) * 16 + r;
          v[u] = __builtin_nontemporal_load(base + (min(row, a.n_ind - 1) - r) * 4);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
          if ((it + u < last) && ((it + u) * 16 + r < a.n_ind)) consume(v[u]);
        it = last;
      }
#pragma unroll
      for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sum32[k][4 * j + 0] += acc_lo[k][j] & 0xFFFFu;
          sum32[k][4 * j + 1] += acc_hi[k][j] & 0xFFFFu;
          sum32[k][4 * j + 2] += acc_lo[k][j] >> 16;
          sum32[k][4 * j + 3] += acc_hi[k][j] >> 16;
        }
    }
    const int64_t site = tile * kTile + (lane & 3) * 16 + r;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      reduce_scatter_step<16, 32>(sum32[k], lane);
      reduce_scatter_step<8, 16>(sum32[k], lane);
      reduce_scatter_step<4, 8>(sum32[k], lane);
      reduce_scatter_step<2, 4>(sum32[k], lane);
      if (site < a.n_sites) __builtin_nontemporal_store(sum32[k][0], a.out + static_cast<int64_t>(a.a0 + k) * a.n_sites + site);
    }
  }
}

struct DdArgs {
  int64_t n_sites;
  int32_t n_src_ind;
  int32_t n_ref_ind;
  int32_t n_tgt_ind;
  int32_t n_windows;
  const uint32_t* ad_ref;  // [n_src_ind][n_sites]
  const uint32_t* ad_tgt;
  const int32_t* lo;
  const int32_t* hi;
  double* scratch;  // [n_windows][n_src_ind]
  double* dd;       // [n_windows]
};

struct ScratchElem {
  const double* p;
  __device__ __forceinline__ double operator()(int i) const { return p[i]; }
};

__global__ __launch_bounds__(256) void window_dd_kernel(DdArgs a) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= a.n_windows) return;
  // the caller's ranges, clamped into the block
  const int lo = static_cast<int>(min(max(static_cast<int64_t>(a.lo[w]), int64_t{0}), a.n_sites));
  const int hi = static_cast<int>(min(max(static_cast<int64_t>(a.hi[w]), int64_t{0}), a.n_sites));
  double* d = a.scratch + static_cast<int64_t>(w) * a.n_src_ind;
  for (int s = 0; s < a.n_src_ind; ++s) {
    const uint32_t* pr = a.ad_ref + static_cast<int64_t>(s) * a.n_sites;
    const uint32_t* pt = a.ad_tgt + static_cast<int64_t>(s) * a.n_sites;
    long long tr = 0, tt = 0;
    for (int i = lo + lane; i < hi; i += 64) {
      tr += pr[i];
      tt += pt[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      tr += __shfl_xor(tr, o, 64);
      tt += __shfl_xor(tt, o, 64);
    }
    if (lane == 0)  // np.mean(cdist(...), axis=1): exact integer sums, one division each
      d[s] = static_cast<double>(tr) / static_cast<double>(a.n_ref_ind) - static_cast<double>(tt) / static_cast<double>(a.n_tgt_ind);
  }
  if (lane == 0) {
    __threadfence_block();
    ScratchElem e{d};
    a.dd[w] = numpy_sum(e, 0, a.n_src_ind) / static_cast<double>(a.n_src_ind);  // np.mean over the individuals
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int sai_site_absdiff(sai_ctx* ctx, int64_t n_sites, const sai_pop* pop, const sai_pop* src, uint32_t* out, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (!pop || !src) return fail(SAI_ERR_ARG, "NULL population");
  if (pop->n_ind < 0 || src->n_ind < 0 || pop->n_ind > (1 << 24)) return fail(SAI_ERR_ARG, "bad n_ind");
  if (n_sites == 0 || src->n_ind == 0) return SAI_OK;
  if (!out || !src->tiles || (pop->n_ind > 0 && !pop->tiles)) return fail(SAI_ERR_ARG, "NULL buffer");
  if ((reinterpret_cast<uintptr_t>(pop->tiles) | reinterpret_cast<uintptr_t>(src->tiles)) & 15u)
    return fail(SAI_ERR_ARG, "tiles must be 16-byte aligned");
  AbsArgs a;
  a.n_sites = n_sites;
  a.n_tiles = (n_sites + kTile - 1) / kTile;
  a.pop_tiles = pop->tiles;
  a.n_ind = pop->n_ind;
  a.src_tiles = src->tiles;
  a.n_src_ind = src->n_ind;
  a.out = out;
  const dim3 grid(stream_grid(ctx, a.n_tiles));
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int a0 = 0; a0 < src->n_ind; a0 += 2) {  // two source individuals per pass over the population
    a.a0 = a0;
    if (src->n_ind - a0 >= 2) hipLaunchKernelGGL((site_absdiff_kernel<2>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((site_absdiff_kernel<1>), grid, dim3(64), 0, st, a);
    if (int rc = check_launch("site_absdiff")) return rc;
  }
  return SAI_OK;
}

int sai_window_dd(sai_ctx* ctx, int64_t n_sites, int32_t n_src_ind, const uint32_t* ad_ref, int32_t n_ref_ind,
                  const uint32_t* ad_tgt, int32_t n_tgt_ind, int32_t n_windows, const int32_t* lo, const int32_t* hi,
                  double* scratch, double* dd, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll || n_windows < 0) return fail(SAI_ERR_ARG, "size out of range");
  if (n_src_ind < 1 || n_ref_ind < 1 || n_tgt_ind < 1) return fail(SAI_ERR_ARG, "every population needs an individual");
  if (n_windows == 0) return SAI_OK;
  if ((n_sites > 0 && (!ad_ref || !ad_tgt)) || !lo || !hi || !scratch || !dd) return fail(SAI_ERR_ARG, "NULL buffer");
  DdArgs a;
  a.n_sites = n_sites;
  a.n_src_ind = n_src_ind;
  a.n_ref_ind = n_ref_ind;
  a.n_tgt_ind = n_tgt_ind;
  a.n_windows = n_windows;
  a.ad_ref = ad_ref;
  a.ad_tgt = ad_tgt;
  a.lo = lo;
  a.hi = hi;
  a.scratch = scratch;
  a.dd = dd;
  hipLaunchKernelGGL(window_dd_kernel, dim3(static_cast<unsigned>((n_windows + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("window_dd");
}

}  // extern "C"
