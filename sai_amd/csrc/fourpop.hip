// fd / df / Danc / Dplus.

#include "common.hpp"
#include "numpy_sum.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// ABBA-BABA family (fd, df, Danc, Dplus; sai/stats/{fd,df,danc,dplus}_statistic.py and
// stat_utils.py:171-272).  site_freqs turns the counts of site_counts into f64 frequencies;
// window_pattern_sums evaluates, per (window, source, pattern), the per-site products
// ((x0*x1)*x2)*x3 with x = f or 1 - f and adds them in numpy's np.sum order (pairwise within
// 8192-element pieces, pieces accumulated in order), so the sums -- and the ratios formed from
// them by window_fourpop -- are the reference's doubles bit for bit.  NaN frequencies (a
// population with no called individual at a site) poison the window's sums, as np.sum does.
// ------------------------------------------------------------------------------------------

struct FreqArgs {
  int64_t n_sites;
  int32_t n_pops;
  int32_t ploidy[kMaxPops + 1];
  const uint2* counts;
  double* freqs;
};

__global__ __launch_bounds__(256) void site_freqs_kernel(FreqArgs a) {
  const int64_t site = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (site >= a.n_sites) return;
  for (int p = 0; p < a.n_pops; ++p) {
    const uint2 c = a.counts[static_cast<int64_t>(p) * a.n_sites + site];
    const int64_t den = static_cast<int64_t>(c.y) * a.ploidy[p];
    a.freqs[static_cast<int64_t>(p) * a.n_sites + site] =
        den > 0 ? static_cast<double>(c.x) / static_cast<double>(den) : std::numeric_limits<double>::quiet_NaN();
  }
}

constexpr int kPatternSlots = 7;  // abba, baba, bbaa, baaa, abaa, abba_d, baba_d

struct PatternElem {
  const double* fr;
  const double* ft;
  const double* fs;
  const double* fo;  // nullptr: outgroup frequency 0 everywhere (stat_utils.py:213-214)
  int bits;          // bit k set: population k contributes f ('b'), else 1 - f ('a'); k = ref,tgt,src,out
  bool donor;        // fd's denominators: tgt and src both replaced by max(tgt, src) (fd_statistic.py:80-83)
  __device__ __forceinline__ double operator()(int i) const {
    const double r = fr[i];
    double t = ft[i], s = fs[i];
    const double o = fo ? fo[i] : 0.0;
    if (donor) {
      const double d = (t != t || s != s) ? std::numeric_limits<double>::quiet_NaN() : (t > s ? t : s);
      t = d;
      s = d;
    }
    double p = (bits & 1) ? r : 1.0 - r;
    p = p * ((bits & 2) ? t : 1.0 - t);
    p = p * ((bits & 4) ? s : 1.0 - s);
    p = p * ((bits & 8) ? o : 1.0 - o);
    return p;
  }
};

struct PatternArgs {
  int64_t n_sites;
  int32_t n_src;
  int32_t has_out;
  int32_t n_windows;
  const double* freqs;  // [2 + n_src + has_out][n_sites]: ref, tgt, sources..., outgroup
  const int32_t* lo;
  const int32_t* hi;
  double* sums;   // [n_windows][n_src][7]
  double* stats;  // [n_windows][n_src][4]: fd, df, Danc, Dplus
};

__global__ __launch_bounds__(64) void window_pattern_sums_kernel(PatternArgs a) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = static_cast<int64_t>(a.n_windows) * a.n_src * kPatternSlots;
  if (tid >= total) return;
  const int slot = static_cast<int>(tid % kPatternSlots);
  const int src = static_cast<int>((tid / kPatternSlots) % a.n_src);
  const int w = static_cast<int>(tid / (kPatternSlots * a.n_src));
  PatternElem e;
  e.fr = a.freqs;
  e.ft = a.freqs + a.n_sites;
  e.fs = a.freqs + static_cast<int64_t>(2 + src) * a.n_sites;
  e.fo = a.has_out ? a.freqs + static_cast<int64_t>(2 + a.n_src) * a.n_sites : nullptr;
  //                 abba  baba  bbaa  baaa  abaa  abba_d baba_d   (bit0 ref, bit1 tgt, bit2 src, bit3 out)
  const int bits[kPatternSlots] = {0x6, 0x5, 0x3, 0x1, 0x2, 0x6, 0x5};
  e.bits = bits[slot];
  e.donor = slot >= 5;
  const int lo = a.lo[w], hi = a.hi[w];
  a.sums[tid] = numpy_sum(e, lo, hi - lo);
}

__device__ __forceinline__ double ratio_or_nan(double num, double den) {
  return den != 0.0 ? num / den : std::numeric_limits<double>::quiet_NaN();
}

__global__ __launch_bounds__(256) void window_fourpop_kernel(PatternArgs a) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (tid >= static_cast<int64_t>(a.n_windows) * a.n_src) return;
  const double* s = a.sums + tid * kPatternSlots;
  const double abba = s[0], baba = s[1], bbaa = s[2], baaa = s[3], abaa = s[4], abba_d = s[5], baba_d = s[6];
  double* out = a.stats + tid * 4;
  out[0] = ratio_or_nan(abba - baba, abba_d - baba_d);                     // fd_statistic.py:85-88
  out[1] = ratio_or_nan(abba - baba, abba + baba + 2 * bbaa);             // df_statistic.py:79-82
  out[2] = ratio_or_nan(baaa - abaa, baaa + abaa);                         // danc_statistic.py:78-81
  out[3] = ratio_or_nan(abba - baba + baaa - abaa, abba + baba + baaa + abaa);  // dplus_statistic.py:81-84
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int sai_site_freqs(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host, const uint32_t* counts,
                   double* freqs, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 1 || n_pops > kMaxPops + 1) return fail(SAI_ERR_ARG, "n_pops must be 1..%d", kMaxPops + 1);
  if (!ploidy_host) return fail(SAI_ERR_ARG, "ploidy_host is NULL");
  if (n_sites == 0) return SAI_OK;
  if (!counts || !freqs) return fail(SAI_ERR_ARG, "NULL buffer");
  FreqArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_sites = n_sites;
  a.n_pops = n_pops;
  for (int p = 0; p < n_pops; ++p) {
    if (ploidy_host[p] <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
    a.ploidy[p] = ploidy_host[p];
  }
  a.counts = reinterpret_cast<const uint2*>(counts);
  a.freqs = freqs;
  hipLaunchKernelGGL(site_freqs_kernel, dim3(static_cast<unsigned>((n_sites + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("site_freqs");
}

int sai_window_fourpop(sai_ctx* ctx, int64_t n_sites, int32_t n_src, int32_t has_outgroup, const double* freqs,
                       int32_t n_windows, const int32_t* lo, const int32_t* hi, double* sums, double* stats,
                       void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll || n_windows < 0) return fail(SAI_ERR_ARG, "size out of range");
  if (n_src < 1 || n_src > SAI_MAX_SRC) return fail(SAI_ERR_ARG, "n_src must be 1..%d", SAI_MAX_SRC);
  if (n_windows == 0) return SAI_OK;
  if ((n_sites > 0 && !freqs) || !lo || !hi || !sums || !stats) return fail(SAI_ERR_ARG, "NULL buffer");
  PatternArgs a;
  a.n_sites = n_sites;
  a.n_src = n_src;
  a.has_out = has_outgroup ? 1 : 0;
  a.n_windows = n_windows;
  a.freqs = freqs;
  a.lo = lo;
  a.hi = hi;
  a.sums = sums;
  a.stats = stats;
  const int64_t n_sum = static_cast<int64_t>(n_windows) * n_src * kPatternSlots;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(window_pattern_sums_kernel, dim3(static_cast<unsigned>((n_sum + 63) / 64)), dim3(64), 0, st, a);
  if (int rc = check_launch("window_pattern_sums")) return rc;
  const int64_t n_stat = static_cast<int64_t>(n_windows) * n_src;
  hipLaunchKernelGGL(window_fourpop_kernel, dim3(static_cast<unsigned>((n_stat + 255) / 256)), dim3(256), 0, st, a);
  return check_launch("window_fourpop");
}

}  // extern "C"
