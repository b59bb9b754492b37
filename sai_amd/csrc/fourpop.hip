// fd / df / Danc / Dplus.

#include "common.hpp"
#include "numpy_sum.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// ABBA-BABA family (fd, df, Danc, Dplus; sai/stats/{fd,df,danc,dplus}_statistic.py and
// stat_utils.py:171-272).  site_freqs turns the counts of site_counts into f64 frequencies;
// window_pattern_sums evaluates, per (window, source), the per-site products ((x0*x1)*x2)*x3 with
// x = f or 1 - f of all seven patterns and adds them in numpy's np.sum order (pairwise within
// 8192-element pieces, pieces accumulated in order) with one wavefront, so the sums -- and the ratios formed from
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

// The per-site products of one source population, all pattern slots at once: every product is
// ((x0*x1)*x2)*x3 with x_k = f_k ('b') or 1 - f_k ('a'), k = ref, tgt, src, out -- calc_pattern_sum's
// `product *= ...` chain (stat_utils.py:259-270; 1.0 * x0 is x0).  Shared prefixes are the same
// operations, so sharing them changes no bit.
struct FourPopElems {
  static constexpr int kSlots = kPatternSlots;
  const double* fr;
  const double* ft;
  const double* fs;
  const double* fo;  // nullptr: outgroup frequency 0 everywhere (stat_utils.py:213-214)
  __device__ __forceinline__ void operator()(int i, double (&p)[kSlots]) const {
    const double r = fr[i], t = ft[i], s = fs[i];
    const double o = fo ? fo[i] : 0.0;
    // fd's denominators: tgt and src both replaced by max(tgt, src) (fd_statistic.py:80-83)
    const double d = (t != t || s != s) ? std::numeric_limits<double>::quiet_NaN() : (t > s ? t : s);
    const double nr = 1.0 - r, nt = 1.0 - t, ns = 1.0 - s, no = 1.0 - o, nd = 1.0 - d;
    const double nr_t = nr * t, r_nt = r * nt;
    p[0] = (nr_t * s) * no;          // abba
    p[1] = (r_nt * s) * no;          // baba
    p[2] = ((r * t) * ns) * no;      // bbaa
    p[3] = (r_nt * ns) * no;         // baaa
    p[4] = (nr_t * ns) * no;         // abaa
    p[5] = ((nr * d) * d) * no;      // abba with the donor frequency
    p[6] = ((r * nd) * d) * no;      // baba with the donor frequency
  }
};

// One arbitrary pattern (calc_pattern_sum's public form): bit k set = population k contributes f.
struct OnePatternElem {
  static constexpr int kSlots = 1;
  const double* f[4];
  int bits;
  __device__ __forceinline__ void operator()(int i, double (&p)[1]) const {
    double v = (bits & 1) ? f[0][i] : 1.0 - f[0][i];
#pragma unroll
    for (int k = 1; k < 4; ++k) v = v * ((bits & (1 << k)) ? f[k][i] : 1.0 - f[k][i]);
    p[0] = v;
  }
};

// ------------------------------------------------------------------------------------------
// np.sum's order, in parallel.  numpy adds an f64 array in 8192-element pieces (the ufunc buffer),
// pieces accumulated in order; a piece is summed by recursive halving (n2 = n/2 rounded down to a
// multiple of 8) down to leaves of <= 128 elements; a leaf keeps EIGHT running sums r[j] over the
// elements j, j+8, j+16, ... and closes with ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a tail of
// < 8 elements.  The order of every addition is fixed, but the eight running sums of a leaf and the
// leaves of a piece are independent: one wavefront sums a window with lane = (leaf % 8, j) -- eight
// leaves at a time, 16 dependent additions per lane for a full leaf instead of thousands for a
// serial walk -- closes each leaf with an xor-butterfly over j (IEEE addition commutes, so the
// butterfly forms exactly numpy's pairs), and one lane per slot then replays the recursion over
// the leaf sums.  Same additions, same order, same bits (golden: fourpop_cases.json incl. windows
// beyond 8192 sites).
// ------------------------------------------------------------------------------------------

constexpr int kMaxLeaves = 72;  // an 8192-element piece has at most 65 leaves (sizes 64..128)
constexpr int kMaxDepth = 8;    // ... at most seven levels down (numpy_sum.hpp)

struct LeafList {
  int off[kMaxLeaves];
  short n[kMaxLeaves];
  short depth[kMaxLeaves];
  int stack_off[kMaxDepth + 1];
  short stack_n[kMaxDepth + 1];
  short stack_depth[kMaxDepth + 1];
  int count;
};

// Leaves of numpy's halving tree over elements [off, off + n), n <= 8192, in left-to-right order
// with their depths (one lane; an explicit stack instead of the recursion).
__device__ __forceinline__ void list_leaves(LeafList* L, int off, int n) {
  int sp = 0, count = 0;
  L->stack_off[0] = off;
  L->stack_n[0] = static_cast<short>(n);
  L->stack_depth[0] = 0;
  while (sp >= 0) {
    const int a = L->stack_off[sp];
    const int len = L->stack_n[sp];
    const int d = L->stack_depth[sp];
    --sp;
    if (len <= 128) {
      if (count < kMaxLeaves) {
        L->off[count] = a;
        L->n[count] = static_cast<short>(len);
        L->depth[count] = static_cast<short>(d);
      }
      ++count;
    } else {
      int n2 = len / 2;
      n2 -= n2 % 8;
      ++sp;  // right half waits on the stack, the left half is visited first
      L->stack_off[sp] = a + n2;
      L->stack_n[sp] = static_cast<short>(len - n2);
      L->stack_depth[sp] = static_cast<short>(d + 1);
      ++sp;
      L->stack_off[sp] = a;
      L->stack_n[sp] = static_cast<short>(n2);
      L->stack_depth[sp] = static_cast<short>(d + 1);
    }
  }
  L->count = count;
}

// np.sum over elements [off, off + n) for every slot of `e`; the whole wave calls it, lanes
// 0..kSlots-1 return their slot's sum.  `list` and `leaf_sums` are this wave's LDS scratch.
template <typename E>
__device__ __forceinline__ double wave_numpy_sum(const E& e, int off, int n, int lane, LeafList* list,
                                                 double (*leaf_sums)[kMaxLeaves]) {
  constexpr int S = E::kSlots;
  constexpr int kBatch = 8;  // leaf iterations whose loads are in flight together
  double total = 0.0;        // meaningful in lanes < S
  for (int o = 0; o < n; o += 8192) {
    const int m = min(8192, n - o);
    if (lane == 0) list_leaves(list, off + o, m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n_leaves = list->count;  // <= 65
    const int j = lane & 7, g = lane >> 3;
    for (int lb = 0; lb < n_leaves; lb += 8) {
      const int leaf = lb + g;
      const bool live = leaf < n_leaves;
      const int lo = live ? list->off[leaf] : 0;
      const int ln = live ? list->n[leaf] : 0;
      const int body = ln - (ln % 8);  // the part the eight running sums cover (0 when ln < 8)
      double r[S];
#pragma unroll
      for (int s = 0; s < S; ++s) r[s] = 0.0;
      // r[j] = a[j]; r[j] += a[i + j] for i = 8, 16, ...: kBatch iterations' loads go out together
      // (indices clamped into the leaf, results of iterations past its end dropped), so a lane
      // waits for memory 4 times per full leaf instead of 16
      for (int i0 = 0; i0 < body; i0 += 8 * kBatch) {
        double p[kBatch][S];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) e(lo + min(i0 + 8 * u, body - 8) + j, p[u]);
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int i = i0 + 8 * u;
          if (i < body) {
#pragma unroll
            for (int s = 0; s < S; ++s) r[s] = (i == 0) ? p[u][s] : r[s] + p[u][s];
          }
        }
      }
      // close the eight running sums: ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); lanes of dead or short
      // leaves shuffle zeros
#pragma unroll
      for (int s = 0; s < S; ++s) {
        r[s] = r[s] + __shfl_xor(r[s], 1, 64);
        r[s] = r[s] + __shfl_xor(r[s], 2, 64);
        r[s] = r[s] + __shfl_xor(r[s], 4, 64);
      }
      if (live && j == 0) {
        double p[S];
        if (ln < 8) {  // numpy's plain loop: res = 0.; res += a[i]
#pragma unroll
          for (int s = 0; s < S; ++s) r[s] = 0.0;
        }
        for (int i = body; i < ln; ++i) {  // the whole of a short leaf, or the tail after the eight-way body
          e(lo + i, p);
#pragma unroll
          for (int s = 0; s < S; ++s) r[s] += p[s];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) leaf_sums[s][leaf] = r[s];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < S) {
      // Rebuild the tree from the leaves' depths: a finished subtree at depth d either waits as a
      // left child (pend[d]) or meets the left sibling waiting there and becomes left + right, one
      // level up -- the recursion's `run(left) + run(right)`, pair for pair.
      double pend[kMaxDepth];
      unsigned waiting = 0;
      double piece = 0.0;
      for (int k = 0; k < n_leaves; ++k) {
        double v = leaf_sums[lane][k];
        int d = list->depth[k];
#pragma unroll
        for (int lv = kMaxDepth - 1; lv >= 1; --lv) {
          if (d == lv && (waiting >> lv & 1u)) {
            v = pend[lv] + v;
            waiting ^= 1u << lv;
            d = lv - 1;
          }
        }
#pragma unroll
        for (int lv = kMaxDepth - 1; lv >= 1; --lv)
          if (d == lv) pend[lv] = v;
        if (d > 0) waiting |= 1u << d;
        else piece = v;
      }
      total += piece;
    }
    __builtin_amdgcn_wave_barrier();  // the next piece overwrites the scratch
  }
  return total;
}

constexpr int kSumWaves = 4;  // waves per workgroup, each with its own scratch

__global__ __launch_bounds__(64 * kSumWaves) void window_pattern_sums_kernel(PatternArgs a) {
  __shared__ LeafList lists[kSumWaves];
  __shared__ double leaf_sums[kSumWaves][kPatternSlots][kMaxLeaves];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // (window, source); each XCD works on a run of neighbouring windows, which share their sites
  const int64_t item = static_cast<int64_t>(xcd_contiguous(blockIdx.x, gridDim.x)) * kSumWaves + wv;
  if (item >= static_cast<int64_t>(a.n_windows) * a.n_src) return;       // whole wave
  const int src = static_cast<int>(item % a.n_src);
  const int w = static_cast<int>(item / a.n_src);
  FourPopElems e;
  e.fr = a.freqs;
  e.ft = a.freqs + a.n_sites;
  e.fs = a.freqs + static_cast<int64_t>(2 + src) * a.n_sites;
  e.fo = a.has_out ? a.freqs + static_cast<int64_t>(2 + a.n_src) * a.n_sites : nullptr;
  // the caller's ranges, clamped into the block
  const int lo = static_cast<int>(min(max(static_cast<int64_t>(a.lo[w]), int64_t{0}), a.n_sites));
  const int hi = static_cast<int>(min(max(static_cast<int64_t>(a.hi[w]), int64_t{0}), a.n_sites));
  const double total = wave_numpy_sum(e, lo, hi - lo, lane, &lists[wv], leaf_sums[wv]);
  if (lane < kPatternSlots) a.sums[item * kPatternSlots + lane] = total;
}

__global__ __launch_bounds__(64) void pattern_sum_kernel(OnePatternElem e, int64_t n, double* out) {
  __shared__ LeafList list;
  __shared__ double leaf_sums[1][kMaxLeaves];
  const int lane = threadIdx.x;
  // np.sum's buffer pieces are counted from the start of the array; int offsets cover 2^31 elements
  const double total = wave_numpy_sum(e, 0, static_cast<int>(n), lane, &list, leaf_sums);
  if (lane == 0) *out = total;
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
  if (n_src < 1 || n_src > SAI_FUSED_SRC) return fail(SAI_ERR_ARG, "n_src must be 1..%d", SAI_FUSED_SRC);
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
  const int64_t n_items = static_cast<int64_t>(n_windows) * n_src;  // one wavefront each
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(window_pattern_sums_kernel, dim3(static_cast<unsigned>((n_items + kSumWaves - 1) / kSumWaves)),
                     dim3(64 * kSumWaves), 0, st, a);
  if (int rc = check_launch("window_pattern_sums")) return rc;
  const int64_t n_stat = static_cast<int64_t>(n_windows) * n_src;
  hipLaunchKernelGGL(window_fourpop_kernel, dim3(static_cast<unsigned>((n_stat + 255) / 256)), dim3(256), 0, st, a);
  return check_launch("window_fourpop");
}

int sai_pattern_sum(sai_ctx* ctx, int64_t n_sites, const double* ref_freq, const double* tgt_freq,
                    const double* src_freq, const double* out_freq, int32_t pattern_bits, double* sum_out,
                    void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (pattern_bits < 0 || pattern_bits > 15) return fail(SAI_ERR_ARG, "pattern_bits must be 0..15");
  if (!sum_out || (n_sites > 0 && (!ref_freq || !tgt_freq || !src_freq || !out_freq))) return fail(SAI_ERR_ARG, "NULL buffer");
  OnePatternElem e;
  e.f[0] = ref_freq;
  e.f[1] = tgt_freq;
  e.f[2] = src_freq;
  e.f[3] = out_freq;
  e.bits = pattern_bits;
  hipLaunchKernelGGL(pattern_sum_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), e, n_sites, sum_out);
  return check_launch("pattern_sum");
}

}  // extern "C"
