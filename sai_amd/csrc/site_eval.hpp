// The per-site decision of compute_matching_loci, shared by site_flags, the fused site pass and
// the packed2 site pass.
#pragma once

#include "common.hpp"

// ------------------------------------------------------------------------------------------
// Per-site decision: calc_freq's f64 division and compute_matching_loci for every parameter set,
// exactly as numpy evaluates it.  One device function, used by the stand-alone site_flags kernel
// (counts read back from HBM) and by the fused tail of site_counts (counts still in LDS).
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bool cmp_op(int op, double f, double y) {
  switch (op) {
    case SAI_OP_EQ: return f == y;
    case SAI_OP_LT: return f < y;
    case SAI_OP_GT: return f > y;
    case SAI_OP_LE: return f <= y;
    default: return f >= y;
  }
}

// Flag planes: what the per-site decision leaves behind for the windows stage (layout: saihip.h).  A
// tile's decisions are taken by the 64 lanes of one wavefront (lane = site), two ballots per set
// collect them, and ONE store instruction per tile writes the tile's row (lane k holds word k):
//   word 0          "any": the sites whose tgt_freq is stored (OR of the conditions, or all ones)
//   word 1 + s      condition of set s (compute_matching_loci's, stat_utils.py:166)
//   word 1 + n + s  site inverted for set s (stat_utils.py:156) -- only when a set lacks ancestral alleles
// and the stored frequencies of the tile go to its first popcount(any) slots in site order.  What is
// written in the middle of the genotype stream is paid per 64-byte LINE: round 2
// kept a byte per site and set (18 B per site for C5), the first plane layout three words per set and
// every candidate's frequency in its own slot (7.5 + 8 partly written lines per tile for C5's 20 sets);
// this one writes 2.6 + 2.5 there and 0.25 + 0.06 for one set.
constexpr int kPlanesPerSet = SAI_PLANES_PER_SET;  // row words RESERVED per set (1 + 2 n <= 3 n are used)

// get(p) -> uint2 {alt_sum, n_called} of population p at this lane's site (site = tile * 64 + lane;
// `live` = the site exists).  Must be called by the whole wavefront.
template <typename GetCounts>
__device__ __forceinline__ void eval_site(int n_pops, const int32_t* ploidy, GetCounts get, int n_sets,
                                          const sai_params* sets, int64_t tile, int lane, bool live, int64_t n_sites,
                                          double* tgt_freq, uint64_t* planes, int64_t plane_stride, double* adj_freq,
                                          bool sparse_freq, bool with_inv) {
  const int64_t site = tile * kTile + lane;
  double f[kMaxPops];
  bool valid = live;
#pragma unroll
  for (int p = 0; p < kMaxPops; ++p) {
    if (p < n_pops) {
      const uint2 c = get(p);
      const int64_t den = static_cast<int64_t>(c.y) * ploidy[p];
      const double v = den > 0 ? static_cast<double>(c.x) / static_cast<double>(den)
                               : std::numeric_limits<double>::quiet_NaN();
      f[p] = v;
      valid = valid && (v >= 0.0) && (v <= 1.0);  // false for NaN; the quotient is never inf
    } else {
      f[p] = 0.0;
    }
  }
  const int n_src = n_pops - 2;
  uint64_t row_word = 0;  // lane k ends up with word k of the tile's row
  uint64_t any = sparse_freq ? 0ull : ~0ull;
  for (int s = 0; s < n_sets; ++s) {
    const sai_params& ps = sets[s];
    bool hit_y = true, hit_m = true;
#pragma unroll
    for (int k = 0; k < SAI_MAX_SRC; ++k) {
      if (k < n_src) {
        hit_y = hit_y && cmp_op(ps.op[k], f[2 + k], ps.y[k]);
        hit_m = hit_m && cmp_op(ps.op[k], f[2 + k], ps.one_minus_y[k]);
      }
    }
    const bool anc = ps.anc_allele_available != 0;
    const bool inverted = !anc && hit_m && valid;
    const bool hit = anc ? hit_y : (hit_y || hit_m);
    const double rf = inverted ? 1.0 - f[0] : f[0];
    const bool cond = valid && hit && (rf < ps.w);
    const uint64_t bc = __ballot(cond);
    any |= bc;
    row_word = lane == 1 + s ? bc : row_word;
    if (with_inv) {  // uniform
      const uint64_t bi = __ballot(inverted);
      row_word = lane == 1 + n_sets + s ? bi : row_word;
    }
    if (adj_freq && live) {
      adj_freq[(static_cast<int64_t>(s) * 2 + 0) * n_sites + site] = rf;
      adj_freq[(static_cast<int64_t>(s) * 2 + 1) * n_sites + site] = inverted ? 1.0 - f[1] : f[1];
    }
  }
  row_word = lane == 0 ? any : row_word;
  // non-temporal stores: measured on MI355X, plain stores in the middle of the genotype stream cost
  // twice as much of the pass as streaming ones
  if (lane < 1 + (with_inv ? 2 : 1) * n_sets) __builtin_nontemporal_store(row_word, planes + tile * plane_stride + lane);
  // The windows stage reads tgt_freq only where a set's condition bit is up (about 1 site in 1000; one in
  // three for C5's loosest sets), and dense f64 stores in the middle of the genotype stream cost ~10 % of
  // the pass: SAI_FREQ_CANDIDATES stores those sites' values only, packed at the start of the tile's slots.
  if (live && ((any >> lane) & 1ull))
    __builtin_nontemporal_store(f[1], tgt_freq + tile * kTile + __popcll(any & ((1ull << lane) - 1ull)));
}

// LDS operations of one wavefront execute in order; this only stops the compiler from moving
// accesses across a point where lanes start reading what other lanes of the wave wrote.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// parameter sets the fused tail of site_counts carries in its kernel arguments (20 x 152 B + the
// rest stay below the 4 KiB kernarg segment)
constexpr int kFusedSets = SAI_FUSED_SETS;

struct FusedArgs {
  int32_t n_sets;  // 0 = plain site_counts
  int16_t sparse_freq;
  int16_t with_inv;  // some set lacks ancestral alleles: the rows carry inverted words
  int32_t ploidy[kMaxPops];
  double* tgt_freq;
  uint64_t* planes;
  int64_t plane_stride;  // words per tile row
  sai_params sets[kFusedSets];
};
