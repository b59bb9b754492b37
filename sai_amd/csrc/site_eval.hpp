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

// ------------------------------------------------------------------------------------------
// Predicate table.  A call's parameter sets repeat the same comparisons: C5's 18 sets hold 6 distinct
// (source, operator, y) and one w, yet the set-by-set form below evaluates 72 source comparisons per site,
// each behind a scalar load and a five-way branch on the operator (a quarter of a C5 tile's time went there,
// during which the wave loads nothing).  The host therefore lists every DISTINCT comparison of a call once --
// (value slot, operator, threshold) with slot p = frequency of population p and slot kBigPops = 1 - ref_freq,
// the reference frequency of an inverted site (stat_utils.py:159) -- sorted by slot, at most 32 of them; a
// site evaluates each into one bit of a 32-bit word, and a set is then the test "all bits of a mask are up":
//   cond        = the set's y comparisons and ref_freq < w                       (stat_utils.py:141-144, 166)
//   mirror      = its 1 - y comparisons                                           (stat_utils.py:148-152)
//   mirror_cond = mirror and 1 - ref_freq < w   (an inverted site is tested on 1 - ref_freq, :156-166)
// The arithmetic per comparison is the reference's: one f64 compare of the same two doubles.
// ------------------------------------------------------------------------------------------
constexpr int kMaxPreds = 32;  // with the masks below the table stays under 1 KiB of LDS next to the 4 KiB of parked counts
constexpr int kTableFromSets = 4;  // parameter sets from which a call's decision is taken from the table
constexpr int kMirrorRefSlot = kBigPops;  // any slot beyond a kernel's populations reads as 1 - ref_freq

struct PredEntry {
  double value;
  int32_t op_bits;  // 1: f < value, 2: f == value, 4: f > value (<= is 3, >= is 6)
  int32_t slot;
};

struct SetMasks {
  uint32_t cond, mirror, mirror_cond;
  int32_t anc;  // anc_allele_available: mirror / mirror_cond are not looked at
};

struct PredTable {
  int32_t n_preds;
  uint8_t slot_end[kBigPops + 2];  // the comparisons of slot s are [slot_end[s - 1], slot_end[s])
  uint8_t pad[2];
  PredEntry preds[kMaxPreds];
  SetMasks sets[SAI_MAX_SETS];
};

// A parameter set as the set-by-set form of the decision reads it: the fields of sai_params that
// compute_matching_loci looks at, for the at most SAI_FUSED_SRC sources of a streaming pass (twenty whole
// sai_params would not fit the kernel arguments).
struct DevSet {
  double w;
  int32_t anc;
  int32_t op[SAI_FUSED_SRC];
  int32_t pad;
  double y[SAI_FUSED_SRC];
  double one_minus_y[SAI_FUSED_SRC];
};

// What the per-site decision is handed: the table (use_table) or, for a call with more than 32 distinct
// comparisons, the parameter sets themselves.
struct EvalSets {
  int32_t use_table;
  int32_t pad;
  union {
    DevSet sets[SAI_MAX_SETS];
    PredTable table;
  };
};

inline int op_bits_of(int op) {
  switch (op) {
    case SAI_OP_EQ: return 2;
    case SAI_OP_LT: return 1;
    case SAI_OP_GT: return 4;
    case SAI_OP_LE: return 3;
    default: return 6;
  }
}

// host: the call's parameter sets as kernel arguments (n_src = source populations of the call).  Returns
// whether the table holds them; when it does not, the sets themselves are handed over -- unless `table_only`
// (the kernel for more than SAI_FUSED_SRC sources has no set-by-set form: its caller takes fewer sets per launch).
inline bool fill_eval_sets(EvalSets& es, int32_t n_sets, const sai_params* sets, int32_t n_src, bool table_only = false) {
  std::memset(&es, 0, sizeof(es));
  struct Raw { int slot, op_bits; double value; };
  Raw raw[SAI_MAX_SETS * (2 * SAI_MAX_SRC + 2)];
  int n_raw = 0;
  bool fits = !std::getenv("SAI_NO_PRED_TABLE");  // test knob: the set-by-set form for every call
  auto index_of = [&](int slot, int op_bits, double value) {
    for (int i = 0; i < n_raw; ++i)
      if (raw[i].slot == slot && raw[i].op_bits == op_bits && std::memcmp(&raw[i].value, &value, sizeof(double)) == 0) return i;
    raw[n_raw] = Raw{slot, op_bits, value};
    return n_raw++;
  };
  for (int s = 0; s < n_sets; ++s) {
    const sai_params& ps = sets[s];
    for (int k = 0; k < n_src && k < SAI_MAX_SRC; ++k) {
      index_of(2 + k, op_bits_of(ps.op[k]), ps.y[k]);
      if (!ps.anc_allele_available) index_of(2 + k, op_bits_of(ps.op[k]), ps.one_minus_y[k]);
    }
    index_of(0, 1, ps.w);
    if (!ps.anc_allele_available) index_of(kMirrorRefSlot, 1, ps.w);
  }
  // one to three sets: the set-by-set form is as fast or faster (its few scalar loads against the table's LDS round
  // trips per tile: site_flags 0.104 against 0.122 ms for one set, 0.172 against 0.147 for four -- profiles/r05_eval_cost.txt;
  // a packed2 tile is over in a quarter of an int8 tile's time and feels it)
  fits = ((fits && n_sets >= kTableFromSets) || table_only) && n_raw <= kMaxPreds;
  if (!fits) {
    if (table_only) return false;
    for (int s = 0; s < n_sets; ++s) {
      DevSet& d = es.sets[s];
      d.w = sets[s].w;
      d.anc = sets[s].anc_allele_available != 0;
      for (int k = 0; k < SAI_FUSED_SRC; ++k) {
        d.op[k] = sets[s].op[k];
        d.y[k] = sets[s].y[k];
        d.one_minus_y[k] = sets[s].one_minus_y[k];
      }
    }
    return false;
  }
  es.use_table = 1;
  PredTable& t = es.table;
  // bit numbers in slot order
  int bit_of[SAI_MAX_SETS * (2 * SAI_MAX_SRC + 2)];
  int n = 0;
  for (int slot = 0; slot <= kMirrorRefSlot; ++slot) {
    for (int i = 0; i < n_raw; ++i)
      if (raw[i].slot == slot) {
        bit_of[i] = n;
        t.preds[n].value = raw[i].value;
        t.preds[n].op_bits = raw[i].op_bits;
        t.preds[n].slot = slot;
        ++n;
      }
    t.slot_end[slot] = static_cast<uint8_t>(n);
  }
  t.n_preds = n;
  auto bit = [&](int slot, int op_bits, double value) { return uint32_t{1} << bit_of[index_of(slot, op_bits, value)]; };
  for (int s = 0; s < n_sets; ++s) {
    const sai_params& ps = sets[s];
    SetMasks& m = t.sets[s];
    m.anc = ps.anc_allele_available != 0;
    m.cond = bit(0, 1, ps.w);
    for (int k = 0; k < n_src && k < SAI_MAX_SRC; ++k) m.cond |= bit(2 + k, op_bits_of(ps.op[k]), ps.y[k]);
    if (!m.anc) {
      for (int k = 0; k < n_src && k < SAI_MAX_SRC; ++k) m.mirror |= bit(2 + k, op_bits_of(ps.op[k]), ps.one_minus_y[k]);
      m.mirror_cond = m.mirror | bit(kMirrorRefSlot, 1, ps.w);
    }
  }
  return true;
}

// `value` (wave-uniform) into lane `lane` (wave-uniform) of `row`, the other lanes as they were: v_writelane_b32
// (this compiler offers no builtin for it; the lane select travels in M0 -- an instruction reads one SGPR only)
__device__ __forceinline__ uint32_t write_lane(uint32_t value, int lane, uint32_t row) {
  const uint32_t v = __builtin_amdgcn_readfirstlane(value);  // uniform already; this pins it to an SGPR
  const int l = __builtin_amdgcn_readfirstlane(lane);
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(row) : "s"(v), "s"(l) : "m0");
  return row;
}

// a value every lane holds alike, as a scalar
__device__ __forceinline__ double uniform_f64(double x) {
  const uint64_t u = __builtin_bit_cast(uint64_t, x);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
  return __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
}

// The workgroup's copy of the table: called by every thread of the workgroup before the first eval_site,
// followed by a workgroup barrier (a wave barrier for single-wave workgroups).
__device__ __forceinline__ void stage_pred_table(const EvalSets& es, PredTable* lds_table) {
  if (!es.use_table) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&es.table);
  uint32_t* dst = reinterpret_cast<uint32_t*>(lds_table);
  for (int i = threadIdx.x; i < static_cast<int>(sizeof(PredTable) / 4); i += blockDim.x) dst[i] = src[i];
}

// get(p) -> uint2 {alt_sum, n_called} of population p at this lane's site (site = tile * 64 + lane;
// `live` = the site exists).  Must be called by the whole wavefront.  MAXP = populations the kernel is built
// for.  `set0` / `n_row_sets`: the call evaluates sets set0 .. set0 + n_sets - 1 of a row that holds n_row_sets
// sets (a decision over more than SAI_FUSED_SRC sources may take a row's sets in several launches; the other
// words of the row are left alone).
template <int MAXP, typename GetCounts>
__device__ __forceinline__ void eval_site(int n_pops, const int32_t* ploidy, GetCounts get, int n_sets,
                                          const EvalSets& es, const PredTable* lds_table, int64_t tile, int lane, bool live, int64_t n_sites,
                                          double* tgt_freq, uint64_t* planes, int64_t plane_stride, double* adj_freq,
                                          bool sparse_freq, bool with_inv, int set0 = 0, int n_row_sets = -1) {
  const int64_t site = tile * kTile + lane;
  const int row_sets = n_row_sets < 0 ? n_sets : n_row_sets;
  double f[MAXP];
  bool valid = live;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
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
  // lane k ends up with word k of the tile's row: a ballot is wave-uniform, v_writelane drops it into its lane
  uint32_t row_lo = 0, row_hi = 0;
  uint64_t any = sparse_freq ? 0ull : ~0ull;
  // what a set leaves behind, whichever way its decision was taken
  auto emit = [&](int s, bool cond, bool inverted) {
    const uint64_t bc = __ballot(cond);
    any |= bc;
    row_lo = write_lane(static_cast<uint32_t>(bc), 1 + set0 + s, row_lo);
    row_hi = write_lane(static_cast<uint32_t>(bc >> 32), 1 + set0 + s, row_hi);
    if (with_inv) {  // uniform
      const uint64_t bi = __ballot(inverted);
      row_lo = write_lane(static_cast<uint32_t>(bi), 1 + row_sets + set0 + s, row_lo);
      row_hi = write_lane(static_cast<uint32_t>(bi >> 32), 1 + row_sets + set0 + s, row_hi);
    }
    if (adj_freq && live) {
      adj_freq[(static_cast<int64_t>(set0 + s) * 2 + 0) * n_sites + site] = inverted ? 1.0 - f[0] : f[0];
      adj_freq[(static_cast<int64_t>(set0 + s) * 2 + 1) * n_sites + site] = inverted ? 1.0 - f[1] : f[1];
    }
  };
  if (es.use_table) {  // uniform
    // The table is read from the workgroup's LDS copy (stage_pred_table), not with a scalar load per comparison and
    // set from the kernel arguments: those loads are waited for one by one and answer as fast as the scalar cache
    // happens to (it is shared with whatever runs next to the pass); LDS answers in the same time whatever else
    // the chip is doing.  Every lane reads the same address; the values are made wave-uniform again with
    // v_readfirstlane so that compares take them as scalars and branches stay scalar.  Alone an 18-set pass is
    // 2 % faster than with the set-by-set form below (profiles/r05_eval_cost.txt; the stand-alone site_flags
    // kernel 0.24 against 0.56 ms); what decides the pipelined step is the pass's grid (profiles/r05_c5_grid.txt).
    const PredTable& t = *lds_table;
    uint32_t miss = ~0u;  // bit i DOWN = comparison i holds at this site
    const int n_preds = __builtin_amdgcn_readfirstlane(t.n_preds);
    PredEntry e = t.preds[0];
    int cur_slot = -1;
    double v = 0.0;
    for (int i = 0; i < n_preds; ++i) {
      const PredEntry next = t.preds[i + 1 < n_preds ? i + 1 : i];  // fetched while entry i is evaluated
      const int slot = __builtin_amdgcn_readfirstlane(e.slot);
      const int ob = __builtin_amdgcn_readfirstlane(e.op_bits);
      const double y = uniform_f64(e.value);
      if (slot != cur_slot) {  // uniform: the comparisons come sorted by value slot
        cur_slot = slot;
        v = 1.0 - f[0];
#pragma unroll
        for (int p = 0; p < MAXP; ++p) v = cur_slot == p ? f[p] : v;
      }
      bool holds;  // one compare per comparison, chosen by a scalar branch; its lane mask selects the bit
      switch (ob) {
        case 2: holds = v == y; break;
        case 1: holds = v < y; break;
        case 4: holds = v > y; break;
        case 3: holds = v <= y; break;
        default: holds = v >= y; break;
      }
      miss &= holds ? ~(1u << i) : ~0u;
      e = next;
    }
    SetMasks m = t.sets[0];
    for (int s = 0; s < n_sets; ++s) {
      const SetMasks next = t.sets[s + 1 < n_sets ? s + 1 : s];
      const uint32_t cond_mask = __builtin_amdgcn_readfirstlane(m.cond);
      if (__builtin_amdgcn_readfirstlane(m.anc)) {
        emit(s, valid && (miss & cond_mask) == 0u, false);
      } else {
        const bool mirrored = (miss & __builtin_amdgcn_readfirstlane(m.mirror)) == 0u;
        const bool mirror_cond = (miss & __builtin_amdgcn_readfirstlane(m.mirror_cond)) == 0u;
        emit(s, valid && (mirror_cond | (!mirrored & ((miss & cond_mask) == 0u))), mirrored && valid);
      }
      m = next;
    }
  } else if constexpr (MAXP == kMaxPops) {  // the set-by-set form: a streaming pass's at most SAI_FUSED_SRC sources
    for (int s = 0; s < n_sets; ++s) {
      const DevSet& ps = es.sets[s];
      bool hit_y = true, hit_m = true;
#pragma unroll
      for (int k = 0; k < SAI_FUSED_SRC; ++k) {
        if (k < n_src) {
          hit_y = hit_y && cmp_op(ps.op[k], f[2 + k], ps.y[k]);
          hit_m = hit_m && cmp_op(ps.op[k], f[2 + k], ps.one_minus_y[k]);
        }
      }
      const bool anc = ps.anc != 0;
      const bool inverted = !anc && hit_m && valid;
      const bool hit = anc ? hit_y : (hit_y || hit_m);
      const double rf = inverted ? 1.0 - f[0] : f[0];
      emit(s, valid && hit && (rf < ps.w), inverted);
    }
  }
  row_lo = write_lane(static_cast<uint32_t>(any), 0, row_lo);
  row_hi = write_lane(static_cast<uint32_t>(any >> 32), 0, row_hi);
  // non-temporal stores: measured on MI355X, plain stores in the middle of the genotype stream cost
  // twice as much of the pass as streaming ones
  const bool my_word = lane == 0 || (lane >= 1 + set0 && lane < 1 + set0 + n_sets) ||
                       (with_inv && lane >= 1 + row_sets + set0 && lane < 1 + row_sets + set0 + n_sets);
  if (my_word) __builtin_nontemporal_store((static_cast<uint64_t>(row_hi) << 32) | row_lo, planes + tile * plane_stride + lane);
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

// parameter sets the fused tail of site_counts carries in its kernel arguments (20 x 152 B, or the
// predicate table in their place, + the rest stay below the 4 KiB kernarg segment)
constexpr int kFusedSets = SAI_FUSED_SETS;

struct FusedArgs {
  int32_t n_sets;  // 0 = plain site_counts
  int16_t sparse_freq;
  int16_t with_inv;  // some set lacks ancestral alleles: the rows carry inverted words
  int32_t pad;
  int32_t ploidy[kMaxPops];
  double* tgt_freq;
  uint64_t* planes;
  int64_t plane_stride;  // words per tile row
  EvalSets es;
};
