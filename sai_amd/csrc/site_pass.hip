// site_counts / site_flags / the fused site pass: the hot kernel of the U / Q path.

#include "site_eval.hpp"
#include "stream_loops.hpp"

namespace {

struct FlagArgs {
  int64_t n_sites;
  int32_t n_pops;
  int16_t n_sets;      // sets of THIS launch
  int16_t with_inv;
  int16_t set0;        // ... which are sets set0 .. set0 + n_sets - 1 of a row of n_row_sets sets
  int16_t n_row_sets;
  int32_t pad;
  int32_t ploidy[kBigPops];
  const uint2* counts;
  double* tgt_freq;
  uint64_t* planes;
  int64_t plane_stride;
  double* adj_freq;
  EvalSets es;
};
static_assert(sizeof(FlagArgs) <= 4096, "kernel arguments exceed the kernarg segment");

// one wavefront per tile (four per workgroup): lane = site inside the tile.  MAXP = kMaxPops: the usual build;
// kBigPops: up to SAI_MAX_SRC sources (table form only; its launcher takes as many sets per launch as the
// table holds).
template <int MAXP>
__global__ __launch_bounds__(256) void site_flags_kernel(FlagArgs a) {
  __shared__ PredTable table;
  stage_pred_table(a.es, &table);
  __syncthreads();
  const int64_t site = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if ((site & ~int64_t{63}) >= a.n_sites) return;  // whole wavefronts beyond the last tile: there is no row for them
  const bool live = site < a.n_sites;              // the last tile's spare lanes vote 0
  eval_site<MAXP>(
      a.n_pops, a.ploidy,
      [&](int p) { return live ? a.counts[static_cast<int64_t>(p) * a.n_sites + site] : make_uint2(0u, 0u); }, a.n_sets,
      a.es, &table, site >> 6, static_cast<int>(threadIdx.x & 63), live, a.n_sites, a.tgt_freq, a.planes, a.plane_stride,
      a.adj_freq, false, a.with_inv != 0, a.set0, a.n_row_sets);
}

// ------------------------------------------------------------------------------------------
// site_counts: the HBM-bound kernel.
//
// One wavefront owns one 64-site tile and streams every population's rows of that tile.  A wave
// instruction loads 16 rows x 64 B = 1 KiB contiguous: lane l holds individual (16*q + l/4),
// sites (l%4)*16 .. +15 as four 32-bit words.  Bytes are accumulated SWAR-style into 16-bit
// (dosage) and 8-bit (missing) fields, widened to 32 bit every <= 248 rows per lane, and the 16
// row-groups are combined with a 4-step butterfly reduce-scatter so that lane l ends with the
// totals of site (l%4)*16 + l/4.  No barriers, no LDS staging of the genotypes (the fused form
// parks each lane's per-population counts in LDS as an indexed register file); 4 KiB of loads in
// flight per wave and 4 waves per SIMD hide HBM latency.
// ------------------------------------------------------------------------------------------

static_assert(sizeof(CountsArgs) + sizeof(FusedArgs) <= 4096, "kernel arguments exceed the kernarg segment");

// MULTI: some population has more than 16 * kChunkIters individuals, so the packed fields are
// widened several times per population (keeps 32 more registers live across the load loop).
// FUSED: evaluate the parameter sets at the end of each tile (site_flags folded in).
// The second launch bound keeps the usual form at 5 waves per SIMD (<= 96 VGPRs): four resident waves of
// more than that leave the windows stage's waves no room next to the pass, and the pipelined step loses
// what the overlap gives (measured at 99 VGPRs: C5 3.63 -> 3.79 ms per step).
// (Round 4 also had a 64-register form for passes with many parameter sets, at 16 waves per CU.  With the
// predicate table of site_eval.hpp the per-site decision no longer decides the grid: the usual form at 12
// waves per CU gives C5's pipelined step 3.09 ms where that form gave 3.15 with the set-by-set decision and
// 3.41 with the table -- profiles/r05_c5_grid.txt -- and the form is gone.)
template <bool MULTI, bool FUSED>
__global__ __launch_bounds__(64, MULTI ? 4 : 5) void site_counts_kernel(CountsArgs a, FusedArgs fa) {
  // FUSED: the butterfly leaves lane l with site (l%4)*16 + l/4 of the tile; each lane parks those
  // {alt_sum, n_called} per population in LDS AT ITS SITE'S INDEX, and once all populations of the
  // tile are done lane l takes site l back and evaluates the parameter sets for it -- lanes in site
  // order, so the ballots per set are the tile's flag planes and the candidates' tgt_freq leave packed
  __shared__ uint2 stash[FUSED ? kMaxPops : 1][FUSED ? 64 : 1];
  __shared__ uint32_t table_words[FUSED ? sizeof(PredTable) / 4 : 1];
  PredTable* table = reinterpret_cast<PredTable*>(table_words);
  if (FUSED) {
    stage_pred_table(fa.es, table);
    wave_lds_fence();
  }
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int lane = threadIdx.x;
    const int r = lane >> 2;
    for (int p = 0; p < a.n_pops; ++p) {
      const int n_ind = a.pop[p].n_ind;
      const u32x4* base =
          reinterpret_cast<const u32x4*>(a.pop[p].tiles + tile * static_cast<int64_t>(n_ind) * kTile) + lane;
      const int n_full = n_ind >> 4;         // iterations in which all 16 rows exist
      const int n_iter = (n_ind + 15) >> 4;  // plus at most one partial iteration
      int it = 0;
      uint2 cnt;
      if (MULTI) {
        uint32_t sum32[16], miss32[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) sum32[j] = miss32[j] = 0;
        while (it < n_iter) {
          uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
          accumulate_rows(base, it, min(n_full, it + kChunkIters), n_full, n_iter, n_ind, r, lo, hi, ms);
          widen_fields(lo, hi, ms, sum32, miss32);
        }
        reduce_scatter_step<16, 32>(sum32, lane);
        reduce_scatter_step<8, 16>(sum32, lane);
        reduce_scatter_step<4, 8>(sum32, lane);
        reduce_scatter_step<2, 4>(sum32, lane);
        reduce_scatter_step<16, 32>(miss32, lane);
        reduce_scatter_step<8, 16>(miss32, lane);
        reduce_scatter_step<4, 8>(miss32, lane);
        reduce_scatter_step<2, 4>(miss32, lane);
        cnt = make_uint2(sum32[0], static_cast<uint32_t>(n_ind) - miss32[0]);
      } else {  // n_iter <= kChunkIters + 1: one pass, widened once -- and ONE butterfly for both sums
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
        accumulate_rows(base, it, n_full, n_full, n_iter, n_ind, r, lo, hi, ms);
        // a site's alt sum stays below 2^20 and its missing count below 2^12 for the at most 3 984 individuals of this
        // form, so the two travel through the butterfly in one word: half of the cross-lane work per population, which
        // is what a tile of NARROW populations is made of (a lone C2: 0.689 against 0.677 of peak; wide populations: the same)
        static_assert(16 * (kChunkIters + 1) * 255 < (1 << 20) && 16 * (kChunkIters + 1) < (1 << 12), "packed butterfly fields overflow");
        uint32_t both[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          both[4 * j + 0] = lo[j] & 0xFFFFu;
          both[4 * j + 1] = hi[j] & 0xFFFFu;
          both[4 * j + 2] = lo[j] >> 16;
          both[4 * j + 3] = hi[j] >> 16;
        }
        if (__ballot((ms[0] | ms[1] | ms[2] | ms[3]) != 0u) != 0ull) {  // wave-uniform: some lane met a missing call
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            both[4 * j + 0] |= (ms[j] & 0xFFu) << 20;
            both[4 * j + 1] |= ((ms[j] >> 8) & 0xFFu) << 20;
            both[4 * j + 2] |= ((ms[j] >> 16) & 0xFFu) << 20;
            both[4 * j + 3] |= (ms[j] >> 24) << 20;
          }
        }
        reduce_scatter_step<16, 32>(both, lane);
        reduce_scatter_step<8, 16>(both, lane);
        reduce_scatter_step<4, 8>(both, lane);
        reduce_scatter_step<2, 4>(both, lane);
        cnt = make_uint2(both[0] & 0xFFFFFu, static_cast<uint32_t>(n_ind) - (both[0] >> 20));
      }
      const int64_t site = tile * kTile + (lane & 3) * 16 + r;
      if (a.counts && site < a.n_sites) store_counts_nt(a.counts + static_cast<int64_t>(p) * a.n_sites + site, cnt);
      if (FUSED) stash[p][(lane & 3) * 16 + r] = cnt;
    }
    if (FUSED) {
      wave_lds_fence();
      eval_site<kMaxPops>(
          a.n_pops, fa.ploidy, [&](int p) { return stash[p][lane]; }, fa.n_sets, fa.es, table, tile, lane,
          tile * kTile + lane < a.n_sites, a.n_sites, fa.tgt_freq, fa.planes, fa.plane_stride, nullptr,
          fa.sparse_freq != 0, fa.with_inv != 0);
      wave_lds_fence();  // the next tile's counts must not overtake these reads
    }
  }
}

}  // namespace

int check_plane_stride(int64_t plane_stride, int32_t n_sets) {
  if (plane_stride < static_cast<int64_t>(kPlanesPerSet) * n_sets || plane_stride > (int64_t{1} << 20))
    return fail(SAI_ERR_ARG, "plane_stride %lld does not hold %d sets", static_cast<long long>(plane_stride), n_sets);
  return SAI_OK;
}

extern "C" int64_t sai_plane_words(int64_t n_sites, int32_t n_sets) {
  if (n_sites < 0 || n_sets < 0 || n_sites >= 0x7FFFFFFFll) return -1;
  return (n_sites + kTile - 1) / kTile * kPlanesPerSet * n_sets;
}

int check_sets(int32_t n_sets, const sai_params* sets, int32_t n_src, int32_t max_sets) {
  if (n_sets < 1 || n_sets > max_sets) return fail(SAI_ERR_ARG, "n_sets must be 1..%d", max_sets);
  if (!sets) return fail(SAI_ERR_ARG, "sets_host is NULL");
  for (int s = 0; s < n_sets; ++s) {
    if (n_src >= 0 && sets[s].n_src != n_src)
      return fail(SAI_ERR_ARG, "set %d: n_src %d != source populations %d", s, sets[s].n_src, n_src);
    for (int k = 0; k < sets[s].n_src && k < SAI_MAX_SRC; ++k)
      if (sets[s].op[k] < SAI_OP_EQ || sets[s].op[k] > SAI_OP_GE)
        return fail(SAI_ERR_ARG, "set %d: bad operator %d", s, sets[s].op[k]);
  }
  return SAI_OK;
}

// shared by sai_site_counts (n_sets == 0) and sai_site_pass
static int launch_site_counts(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                              int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                              uint64_t* planes, int64_t plane_stride, void* stream) {
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 1 || n_pops > kMaxPops) return fail(SAI_ERR_ARG, "n_pops must be 1..%d", kMaxPops);
  if (!pops) return fail(SAI_ERR_ARG, "pops is NULL");
  if (n_sites == 0) return SAI_OK;
  CountsArgs a;
  FusedArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  a.n_sites = n_sites;
  a.n_tiles = (n_sites + kTile - 1) / kTile;
  a.n_pops = n_pops;
  bool multi = false;
  for (int p = 0; p < n_pops; ++p) {
    if (pops[p].n_ind < 0) return fail(SAI_ERR_ARG, "population %d: negative n_ind", p);
    if (pops[p].n_ind > 0 && !pops[p].tiles) return fail(SAI_ERR_ARG, "population %d: NULL tiles", p);
    if (pops[p].n_ind > (1 << 24)) return fail(SAI_ERR_UNSUPPORTED, "population %d: n_ind > 2^24", p);
    if (reinterpret_cast<uintptr_t>(pops[p].tiles) & 15u)
      return fail(SAI_ERR_ARG, "population %d: tiles must be 16-byte aligned", p);
    a.pop[p].tiles = pops[p].tiles;
    a.pop[p].n_ind = pops[p].n_ind;
    a.pop[p].pad = 0;
    multi = multi || pops[p].n_ind > 16 * kChunkIters;
    fa.ploidy[p] = pops[p].ploidy;
  }
  a.counts = reinterpret_cast<uint2*>(counts);
  fa.n_sets = n_sets;
  fa.sparse_freq = freq_mode == SAI_FREQ_CANDIDATES;
  fa.with_inv = n_sets > 0 && sets_with_inverted(n_sets, sets_host);
  fa.tgt_freq = tgt_freq;
  fa.planes = planes;
  fa.plane_stride = plane_stride;
  if (n_sets > 0) fill_eval_sets(fa.es, n_sets, sets_host, n_pops - 2);
  int64_t individuals = 0;
  for (int p = 0; p < n_pops; ++p) individuals += pops[p].n_ind;
  const dim3 grid(stream_grid(ctx, a.n_tiles, site_pass_waves_per_cu(ctx, a.n_tiles, n_sets, n_pops, individuals)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_sets > 0) {
    if (multi) launch_pass(ctx, site_counts_kernel<true, true>, grid, dim3(64), st, a, fa);
    else launch_pass(ctx, site_counts_kernel<false, true>, grid, dim3(64), st, a, fa);
  } else {
    if (multi) launch_pass(ctx, site_counts_kernel<true, false>, grid, dim3(64), st, a, fa);
    else launch_pass(ctx, site_counts_kernel<false, false>, grid, dim3(64), st, a, fa);
  }
  return check_launch("site_counts");
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int sai_site_counts(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                    void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (!counts && n_sites > 0) return fail(SAI_ERR_ARG, "counts is NULL");
  return launch_site_counts(ctx, n_sites, n_pops, pops, counts, 0, nullptr, SAI_FREQ_DENSE, nullptr, nullptr, 0, stream);
}

int sai_site_pass(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                  int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq, uint64_t* planes,
                  int64_t plane_stride, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (freq_mode != SAI_FREQ_DENSE && freq_mode != SAI_FREQ_CANDIDATES) return fail(SAI_ERR_ARG, "bad freq_mode %d", freq_mode);
  if (n_pops < 2) return fail(SAI_ERR_ARG, "n_pops must be >= 2 (ref, tgt, sources)");
  if (n_sets > kFusedSets)
    return fail(SAI_ERR_UNSUPPORTED, "sai_site_pass carries at most %d parameter sets; use sai_site_counts + sai_site_flags",
                kFusedSets);
  if (int rc = check_sets(n_sets, sets_host, n_pops - 2, kFusedSets)) return rc;
  if (pops)
    for (int p = 0; p < n_pops && p < kMaxPops; ++p)
      if (pops[p].ploidy <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
  if (n_sites > 0 && (!tgt_freq || !planes)) return fail(SAI_ERR_ARG, "NULL buffer");
  if (int rc = check_plane_stride(plane_stride, n_sets)) return rc;
  return launch_site_counts(ctx, n_sites, n_pops, pops, counts, n_sets, sets_host, freq_mode, tgt_freq, planes,
                            plane_stride, stream);
}

int sai_site_flags(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                   const uint32_t* counts, int32_t n_sets, const sai_params* sets_host, double* tgt_freq,
                   uint64_t* planes, int64_t plane_stride, double* adj_freq, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 2 || n_pops > kBigPops) return fail(SAI_ERR_ARG, "n_pops must be 2..%d (ref, tgt, sources)", kBigPops);
  if (!ploidy_host) return fail(SAI_ERR_ARG, "ploidy_host is NULL");
  if (int rc = check_sets(n_sets, sets_host, n_pops - 2)) return rc;
  if (n_sites == 0) return SAI_OK;
  if (!counts || !tgt_freq || !planes) return fail(SAI_ERR_ARG, "NULL buffer");
  if (int rc = check_plane_stride(plane_stride, n_sets)) return rc;
  FlagArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_sites = n_sites;
  a.n_pops = n_pops;
  a.with_inv = sets_with_inverted(n_sets, sets_host);
  a.n_row_sets = static_cast<int16_t>(n_sets);
  for (int p = 0; p < n_pops; ++p) {
    if (ploidy_host[p] <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
    a.ploidy[p] = ploidy_host[p];
  }
  a.counts = reinterpret_cast<const uint2*>(counts);
  a.tgt_freq = tgt_freq;
  a.planes = planes;
  a.plane_stride = plane_stride;
  a.adj_freq = adj_freq;
  const unsigned grid = static_cast<unsigned>((n_sites + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_pops <= kMaxPops) {
    a.n_sets = static_cast<int16_t>(n_sets);
    fill_eval_sets(a.es, n_sets, sets_host, n_pops - 2);
    hipLaunchKernelGGL(site_flags_kernel<kMaxPops>, dim3(grid), dim3(256), 0, st, a);
    return check_launch("site_flags");
  }
  // more than SAI_FUSED_SRC sources (stat_utils.py:114-119 loops over any number): the table form only, as many
  // sets of the row per launch as the table's 32 comparisons hold (one set of 14 sources needs at most 30)
  for (int s0 = 0; s0 < n_sets;) {
    int k = n_sets - s0;
    while (k > 1 && !fill_eval_sets(a.es, k, sets_host + s0, n_pops - 2, true)) --k;
    if (k == 1 && !fill_eval_sets(a.es, 1, sets_host + s0, n_pops - 2, true))
      return fail(SAI_ERR_UNSUPPORTED, "set %d: more distinct comparisons than one launch evaluates", s0);
    a.set0 = static_cast<int16_t>(s0);
    a.n_sets = static_cast<int16_t>(k);
    hipLaunchKernelGGL(site_flags_kernel<kBigPops>, dim3(grid), dim3(256), 0, st, a);
    if (int rc = check_launch("site_flags")) return rc;
    s0 += k;
  }
  return SAI_OK;
}

}  // extern "C"
