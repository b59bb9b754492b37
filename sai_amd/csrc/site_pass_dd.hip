// The site pass with DD's per-site terms riding along: the genotypes of ref and tgt are read ONCE for the
// counts, the per-site decision AND the city-block terms of up to SAI_DD_FUSED_ROWS source individuals
// (sai_site_pass_dd).  The stand-alone form (dd.hip: sai_site_absdiff) streams a population once more per
// two source individuals -- an all-seven-statistics run read the genotypes two to three times with it.

#include "site_eval.hpp"
#include "stream_loops.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// Per site and source individual a, DD needs  sum_b |s_a - g_b|  over the individuals b of ref (resp. tgt)
// on the raw int8 dosages (dd_statistic.py:64-66: scipy's cdist "cityblock"; a missing call enters as its
// negative number).  v_sad_u8 adds up the absolute differences of the four bytes of two words -- but a loaded
// word holds four SITES of one individual.  So the four words that the four loads in flight hold for the same
// four sites (four individuals) are transposed in registers (eight v_perm_b32) into one word per site with
// four individuals in it, every byte biased to unsigned (x ^ 0x80: u = g + 128, s' = s + 128), and ONE
// instruction per site-word and source individual accumulates  DD_a += sad(T, s'_a x 4)  in 32 bits (any
// population size).  The source's byte replicated four times is made once per tile.  The COUNTS of the same loads
// are the site pass's own (stream_loops.hpp: a group without a missing call and without a dosage of 64 or more is
// added up bytewise) -- until the middle of round 5 they were three more SADs per site-word (sum u, sum |g|,
// 128 [g >= 0]).  12 + 4 NS operations per word of four genotypes for the DD terms (NS = 2: 20 per load and
// lane + 9-11 for the counts; the SAD-only form: 44; the packed-16-bit form of dd.hip: 66 on top of the counts).
// Rows that do not exist (the last batch of four row groups is padded) are loaded as zero words, i.e. as called
// dosages of 0: they add nothing to the counts, and the P |s_a| they add to DD_a are taken off afterwards.  All
// integers are exact (u32 arithmetic; every result is below 255 * 2^24 < 2^32).
// ------------------------------------------------------------------------------------------

constexpr int kDdRows = SAI_DD_FUSED_ROWS;

struct DdRow {
  const int8_t* tiles;  // the source population's block
  int64_t tile_bytes;   // n_ind * 64
  int32_t row_off;      // row * 64
  int32_t pad;
};

struct DdArgs {
  int32_t n_rows;
  int32_t pad;
  uint32_t* out;  // [2][n_rows][n_sites]
  DdRow row[kDdRows];
};
static_assert(sizeof(CountsArgs) + sizeof(FusedArgs) + sizeof(DdArgs) <= 4096, "kernel arguments exceed the kernarg segment");

// bytes b of four words -> four words of one byte position each: t[b] = {w0.b, w1.b, w2.b, w3.b}
__device__ __forceinline__ void transpose_bytes(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t (&t)[4]) {
  const uint32_t a_lo = __builtin_amdgcn_perm(w1, w0, 0x05010400u);  // w0.b0 w1.b0 w0.b1 w1.b1
  const uint32_t a_hi = __builtin_amdgcn_perm(w1, w0, 0x07030602u);  // w0.b2 w1.b2 w0.b3 w1.b3
  const uint32_t b_lo = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
  const uint32_t b_hi = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
  t[0] = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);
  t[1] = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
  t[2] = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
  t[3] = __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
}

// DD's terms of four row groups (the four loads in flight) of this lane's 16 sites; sv[k][j] = biased word j of source row k
template <int NS>
__device__ __forceinline__ void dd_batch(const u32x4 (&v)[4], uint32_t (&dd)[NS][16], const uint32_t (&sv)[NS][4]) {
  const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                            {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t t[4];
    transpose_bytes(w[0][j] ^ 0x80808080u, w[1][j] ^ 0x80808080u, w[2][j] ^ 0x80808080u, w[3][j] ^ 0x80808080u, t);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        // byte b of the source's word, four times (a value per tile: the compiler keeps it in a register where it has one)
        const uint32_t rep = __builtin_amdgcn_perm(0u, sv[k][j], 0x01010101u * static_cast<uint32_t>(b));
        dd[k][4 * j + b] = __builtin_amdgcn_sad_u8(t[b], rep, dd[k][4 * j + b]);
      }
  }
}

// the lanes' 16 partial sums -> this lane's ONE site ((lane % 4) * 16 + lane / 4 of the tile)
__device__ __forceinline__ uint32_t butterfly(uint32_t (&sum32)[16], int lane) {
  reduce_scatter_step<16, 32>(sum32, lane);
  reduce_scatter_step<8, 16>(sum32, lane);
  reduce_scatter_step<4, 8>(sum32, lane);
  reduce_scatter_step<2, 4>(sum32, lane);
  return sum32[0];
}

// A chunk's packed fields -> {alt_sum, missing} of this lane's site: one butterfly for both (a chunk is at most
// kChunkIters + kUnroll iterations: the alt sum stays below 2^20, the missing count below 2^12 -- site_pass.hip)
__device__ __forceinline__ uint2 close_chunk(const uint32_t (&lo)[4], const uint32_t (&hi)[4], const uint32_t (&ms)[4], int lane) {
  static_assert(16 * (kChunkIters + kUnroll) * 255 < (1 << 20) && 16 * (kChunkIters + kUnroll) < (1 << 12), "packed butterfly fields overflow");
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
  const uint32_t v = butterfly(both, lane);
  return make_uint2(v & 0xFFFFFu, v >> 20);
}

// A population whose counts only are needed (sources, outgroup): the site pass's loop, any size -- a population of
// more than 3 968 individuals closes its chunks one by one (a butterfly per 248 wave loads) instead of widening
// 32 partial sums per lane.
__device__ __forceinline__ uint2 count_population(const u32x4* base, int n_ind, int lane) {
  const int n_full = n_ind >> 4, n_iter = (n_ind + 15) >> 4;
  uint32_t sum = 0, miss = 0;
  int it = 0;
  while (it < n_iter) {
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
    accumulate_rows(base, it, min(n_full, it + kChunkIters), n_full, n_iter, n_ind, lane >> 2, lo, hi, ms);
    const uint2 c = close_chunk(lo, hi, ms, lane);
    sum += c.x;
    miss += c.y;
  }
  return make_uint2(sum, static_cast<uint32_t>(n_ind) - miss);
}

// ref or tgt of a tile: streams its rows four groups at a time and leaves this lane's site with {alt_sum, n_called}
// and the NS terms of DD (`ad`).  s_abs[k] = |s_k| at this lane's site.
template <int NS>
__device__ __forceinline__ uint2 dd_population(const u32x4* base, int n_ind, int lane, const uint32_t (&sv)[NS][4],
                                               const uint32_t (&s_abs)[NS], uint32_t (&ad)[NS]) {
  const int r = lane >> 2;
  const int n_full = n_ind >> 4;         // iterations in which all 16 rows exist
  const int n_iter = (n_ind + 15) >> 4;  // plus at most one partial iteration
  uint32_t dd[NS][16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int k = 0; k < NS; ++k) dd[k][s] = 0;
  uint32_t sum = 0, miss = 0;
  int it = 0, slots = 0;
  while (it < n_iter) {  // chunks the 16-/8-bit fields of the counts can absorb
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
    const int chunk_end = min(n_iter, it + kChunkIters);
    const int full_end = min(n_full, chunk_end);
    for (; it + kUnroll <= full_end; it += kUnroll) {
      u32x4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
      acc_group(v, lo, hi, ms);
      dd_batch<NS>(v, dd, sv);
    }
    slots = it;
    if (chunk_end == n_iter && it < n_iter) {  // the last one to four groups as ONE batch of clamped loads; rows that do not exist become zero words
      u32x4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int row = min(it + u, n_iter - 1) * 16 + r;
        v[u] = __builtin_nontemporal_load(base + (min(row, n_ind - 1) - r) * 4);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
        if (!((it + u < n_iter) && ((it + u) * 16 + r < n_ind))) v[u] = u32x4{0u, 0u, 0u, 0u};
      acc_group(v, lo, hi, ms);
      dd_batch<NS>(v, dd, sv);
      slots = it + kUnroll;
      it = n_iter;
    }
    const uint2 c = close_chunk(lo, hi, ms, lane);  // (a butterfly per 248 wave loads for populations of more than 3 968 individuals)
    sum += c.x;
    miss += c.y;
  }
  const uint32_t pad_rows = 16u * static_cast<uint32_t>(slots) - static_cast<uint32_t>(n_ind);
#pragma unroll
  for (int k = 0; k < NS; ++k) ad[k] = butterfly(dd[k], lane) - pad_rows * s_abs[k];
  return make_uint2(sum, static_cast<uint32_t>(n_ind) - miss);
}

// NS source individuals ride along.  Registers: the counts' 32 partial sums and 12 fields + 32 per source individual
// (16 sums, 16 replicated bytes) + the loads in flight -- three waves per SIMD for one or two source individuals,
// two for three or four.
template <int NS, bool FUSED>
__global__ __launch_bounds__(64, NS <= 2 ? 3 : 2) void site_counts_dd_kernel(CountsArgs a, FusedArgs fa, DdArgs d) {
  __shared__ uint2 stash[FUSED ? kMaxPops : 1][FUSED ? 64 : 1];
  __shared__ uint32_t table_words[FUSED ? sizeof(PredTable) / 4 : 1];
  PredTable* table = reinterpret_cast<PredTable*>(table_words);
  if (FUSED) {
    stage_pred_table(fa.es, table);
    wave_lds_fence();
  }
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int lane = threadIdx.x;
    const int my_site = (lane & 3) * 16 + (lane >> 2);  // the site this lane holds after the butterfly
    // this lane's 16 sites of every source individual, biased; and |s| of its own site
    uint32_t sv[NS][4], s_abs[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int8_t* row = d.row[k].tiles + tile * d.row[k].tile_bytes + d.row[k].row_off;
      const u32x4 v = *(reinterpret_cast<const u32x4*>(row) + (lane & 3));
      const int own = row[my_site];
      s_abs[k] = static_cast<uint32_t>(own < 0 ? -own : own);
      sv[k][0] = v.x ^ 0x80808080u;
      sv[k][1] = v.y ^ 0x80808080u;
      sv[k][2] = v.z ^ 0x80808080u;
      sv[k][3] = v.w ^ 0x80808080u;
    }
    const int64_t site = tile * kTile + my_site;
    for (int p = 0; p < a.n_pops; ++p) {
      const int n_ind = a.pop[p].n_ind;
      const u32x4* base =
          reinterpret_cast<const u32x4*>(a.pop[p].tiles + tile * static_cast<int64_t>(n_ind) * kTile) + lane;
      uint2 cnt;
      if (p < 2) {  // ref, tgt: counts and DD terms from the same loads
        uint32_t ad[NS];
        cnt = dd_population<NS>(base, n_ind, lane, sv, s_abs, ad);
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (site < a.n_sites)
            __builtin_nontemporal_store(ad[k], d.out + (static_cast<int64_t>(p) * d.n_rows + k) * a.n_sites + site);
      } else {  // sources, outgroup: the counts
        cnt = count_population(base, n_ind, lane);
      }
      if (a.counts && site < a.n_sites) store_counts_nt(a.counts + static_cast<int64_t>(p) * a.n_sites + site, cnt);
      if (FUSED) stash[p][my_site] = cnt;
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

template <int NS>
void launch_dd(sai_ctx* ctx, bool fused, dim3 grid, hipStream_t st, const CountsArgs& a, const FusedArgs& fa, const DdArgs& d) {
  if (fused) launch_pass(ctx, site_counts_dd_kernel<NS, true>, grid, dim3(64), st, a, fa, d);
  else launch_pass(ctx, site_counts_dd_kernel<NS, false>, grid, dim3(64), st, a, fa, d);
}

}  // namespace

extern "C" int sai_site_pass_dd(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                                int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                                uint64_t* planes, int64_t plane_stride, const sai_dd_rows* dd, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (freq_mode != SAI_FREQ_DENSE && freq_mode != SAI_FREQ_CANDIDATES) return fail(SAI_ERR_ARG, "bad freq_mode %d", freq_mode);
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 2 || n_pops > kMaxPops) return fail(SAI_ERR_ARG, "n_pops must be 2..%d (ref, tgt, sources)", kMaxPops);
  if (!pops) return fail(SAI_ERR_ARG, "pops is NULL");
  if (!dd) return fail(SAI_ERR_ARG, "dd is NULL");
  if (n_sets < 0 || n_sets > kFusedSets) return fail(SAI_ERR_UNSUPPORTED, "at most %d parameter sets", kFusedSets);
  if (dd->first_pop < 2 || dd->n_pops < 1 || dd->first_pop + dd->n_pops > n_pops)
    return fail(SAI_ERR_ARG, "dd: populations %d..%d are not source populations of this call", dd->first_pop,
                dd->first_pop + dd->n_pops - 1);
  if (n_sets > 0) {
    // with parameter sets every population behind tgt is a source of the decision (sai_site_pass)
    if (int rc = check_sets(n_sets, sets_host, n_pops - 2, kFusedSets)) return rc;
    if (n_sites > 0 && (!tgt_freq || !planes)) return fail(SAI_ERR_ARG, "NULL buffer");
    if (int rc = check_plane_stride(plane_stride, n_sets)) return rc;
  }
  CountsArgs a;
  FusedArgs fa;
  DdArgs d;
  std::memset(&fa, 0, sizeof(fa));
  std::memset(&d, 0, sizeof(d));
  a.n_sites = n_sites;
  a.n_tiles = (n_sites + kTile - 1) / kTile;
  a.n_pops = n_pops;
  int64_t individuals = 0;
  for (int p = 0; p < n_pops; ++p) {
    if (pops[p].n_ind < 0) return fail(SAI_ERR_ARG, "population %d: negative n_ind", p);
    if (pops[p].n_ind > 0 && !pops[p].tiles) return fail(SAI_ERR_ARG, "population %d: NULL tiles", p);
    if (reinterpret_cast<uintptr_t>(pops[p].tiles) & 15u)
      return fail(SAI_ERR_ARG, "population %d: tiles must be 16-byte aligned", p);
    if (n_sets > 0 && pops[p].ploidy <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
    if (pops[p].n_ind > (1 << 24)) return fail(SAI_ERR_UNSUPPORTED, "population %d: n_ind > 2^24", p);
    a.pop[p].tiles = pops[p].tiles;
    a.pop[p].n_ind = pops[p].n_ind;
    a.pop[p].pad = 0;
    fa.ploidy[p] = pops[p].ploidy;
    individuals += pops[p].n_ind;
  }
  int64_t n_rows = 0;
  for (int q = dd->first_pop; q < dd->first_pop + dd->n_pops; ++q) n_rows += pops[q].n_ind;
  if (n_rows < 1) return fail(SAI_ERR_ARG, "dd: the source populations hold no individual");
  if (n_rows > kDdRows)
    return fail(SAI_ERR_UNSUPPORTED, "dd: %lld source individuals, at most %d ride along; use sai_site_pass + sai_site_absdiff",
                static_cast<long long>(n_rows), kDdRows);
  if (n_sites == 0) return SAI_OK;
  if (!dd->absdiff) return fail(SAI_ERR_ARG, "dd: absdiff is NULL");
  d.n_rows = static_cast<int32_t>(n_rows);
  d.out = dd->absdiff;
  int k = 0;
  for (int q = dd->first_pop; q < dd->first_pop + dd->n_pops; ++q)
    for (int row = 0; row < pops[q].n_ind; ++row, ++k) {
      d.row[k].tiles = pops[q].tiles;
      d.row[k].tile_bytes = static_cast<int64_t>(pops[q].n_ind) * kTile;
      d.row[k].row_off = row * kTile;
    }
  a.counts = reinterpret_cast<uint2*>(counts);
  fa.n_sets = n_sets;
  fa.sparse_freq = freq_mode == SAI_FREQ_CANDIDATES;
  fa.with_inv = n_sets > 0 && sets_with_inverted(n_sets, sets_host);
  fa.tgt_freq = tgt_freq;
  fa.planes = planes;
  fa.plane_stride = plane_stride;
  if (n_sets > 0) fill_eval_sets(fa.es, n_sets, sets_host, n_pops - 2);
  const dim3 grid(stream_grid(ctx, a.n_tiles, dd_pass_waves_per_cu(ctx, a.n_tiles, individuals, static_cast<int>(n_rows))));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (n_rows) {
    case 1: launch_dd<1>(ctx, n_sets > 0, grid, st, a, fa, d); break;
    case 2: launch_dd<2>(ctx, n_sets > 0, grid, st, a, fa, d); break;
    case 3: launch_dd<3>(ctx, n_sets > 0, grid, st, a, fa, d); break;
    default: launch_dd<4>(ctx, n_sets > 0, grid, st, a, fa, d); break;
  }
  return check_launch("site_counts_dd");
}
