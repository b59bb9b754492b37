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
// four individuals in it, and with every byte biased to unsigned (x ^ 0x80: u = g + 128, s' = s + 128) ONE
// instruction per site-word serves each sum, accumulating in 32 bits (no field ever needs widening, so any
// population size is served):
//     RAW'  += sad(T, 0)             sum_b u_b
//     ABS   += sad(T, 0x80808080)    sum_b |g_b|        -> alt_sum = (ABS + sum g) / 2   (stat_utils.py:48)
//     C128  += sad(T & 0x80808080, 0) 128 [g_b >= 0]     -> n_called                       (stat_utils.py:46)
//     DD_a  += sad(T, s'_a x 4)      sum_b |s_a - g_b|
// -- 7 + 2 NS operations per word of four genotypes where the packed 16-bit form of dd.hip needs 4 + 8 NS on top
// of the counts' 11.  Rows that do not exist (the last batch of four row groups is padded) are loaded as zero
// words, i.e. as called dosages of 0, and what P such rows add to a site's sums is taken off afterwards:
// 128 P from RAW', P from the called count, P |s_a| from DD_a, nothing from ABS.  All integers are exact
// (u32 arithmetic wraps; every result is below 255 * 2^24 < 2^32).
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

// sums per site of this lane's 16 sites; NS = 0: the counts only (sources, outgroup)
template <int NS>
struct SadAcc {
  uint32_t raw[16], abs_[16], c128[16];
  uint32_t dd[NS > 0 ? NS : 1][16];
};

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

// four row groups (the four loads in flight) of this lane's 16 sites; sv[k][j] = biased word j of source row k
template <int NS>
__device__ __forceinline__ void sad_batch(const u32x4 (&v)[4], SadAcc<NS>& a, const uint32_t (&sv)[NS > 0 ? NS : 1][4]) {
  const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                            {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t t[4];
    transpose_bytes(w[0][j] ^ 0x80808080u, w[1][j] ^ 0x80808080u, w[2][j] ^ 0x80808080u, w[3][j] ^ 0x80808080u, t);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int s = 4 * j + b;
      a.raw[s] = __builtin_amdgcn_sad_u8(t[b], 0u, a.raw[s]);
      a.abs_[s] = __builtin_amdgcn_sad_u8(t[b], 0x80808080u, a.abs_[s]);
      a.c128[s] = __builtin_amdgcn_sad_u8(t[b] & 0x80808080u, 0u, a.c128[s]);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const uint32_t rep = __builtin_amdgcn_perm(0u, sv[k][j], 0x01010101u * static_cast<uint32_t>(b));  // byte b x 4
        a.dd[k][s] = __builtin_amdgcn_sad_u8(t[b], rep, a.dd[k][s]);
      }
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

// One population of a tile: streams its rows four groups at a time and leaves this lane's site with
// {alt_sum, n_called} and, for NS > 0, the NS terms of DD (`ad`).  `pad_rows` comes back as the number of zero
// rows the sums of a site include.
template <int NS>
__device__ __forceinline__ uint2 sad_population(const u32x4* base, int n_ind, int lane, const uint32_t (&sv)[NS > 0 ? NS : 1][4],
                                                const uint32_t (&s_own)[NS > 0 ? NS : 1], uint32_t (&ad)[NS > 0 ? NS : 1]) {
  const int r = lane >> 2;
  const int n_full = n_ind >> 4;         // iterations in which all 16 rows exist
  const int n_iter = (n_ind + 15) >> 4;  // plus at most one partial iteration
  SadAcc<NS> acc;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc.raw[s] = acc.abs_[s] = acc.c128[s] = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) acc.dd[k][s] = 0;
  }
  int it = 0;
  for (; it + kUnroll <= n_full; it += kUnroll) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
    sad_batch<NS>(v, acc, sv);
  }
  int slots = it;
  if (it < n_iter) {  // the last one to four groups as ONE batch of clamped loads; rows that do not exist become zero words
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int row = min(it + u, n_iter - 1) * 16 + r;
      v[u] = __builtin_nontemporal_load(base + (min(row, n_ind - 1) - r) * 4);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
      if (!((it + u < n_iter) && ((it + u) * 16 + r < n_ind))) v[u] = u32x4{0u, 0u, 0u, 0u};
    sad_batch<NS>(v, acc, sv);
    slots += kUnroll;
  }
  const uint32_t n = static_cast<uint32_t>(n_ind);
  const uint32_t pad_rows = 16u * static_cast<uint32_t>(slots) - n;
  const uint32_t raw = butterfly(acc.raw, lane);                  // sum u over the real rows + 128 pad_rows
  const uint32_t abs_sum = butterfly(acc.abs_, lane);             // sum |g|
  const uint32_t called = butterfly(acc.c128, lane) / 128u - pad_rows;
  const uint32_t g_sum = raw - 128u * (n + pad_rows);             // sum g (two's complement when negative)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const uint32_t s_abs = s_own[k] >= 128u ? s_own[k] - 128u : 128u - s_own[k];  // |s| of this lane's site
    ad[k] = butterfly(acc.dd[k], lane) - pad_rows * s_abs;
  }
  return make_uint2((abs_sum + g_sum) >> 1, called);
}

// NS source individuals ride along.  Registers: the 48 sums of the counts + 16 per source individual + the
// loads in flight -- three waves per SIMD for one or two source individuals, two for three or four.
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
    // this lane's 16 sites of every source individual, biased; and the byte of its own site
    uint32_t sv[NS][4], s_own[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int8_t* row = d.row[k].tiles + tile * d.row[k].tile_bytes + d.row[k].row_off;
      const u32x4 v = *(reinterpret_cast<const u32x4*>(row) + (lane & 3));
      s_own[k] = static_cast<uint32_t>(static_cast<uint8_t>(row[my_site])) ^ 0x80u;
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
        cnt = sad_population<NS>(base, n_ind, lane, sv, s_own, ad);
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (site < a.n_sites)
            __builtin_nontemporal_store(ad[k], d.out + (static_cast<int64_t>(p) * d.n_rows + k) * a.n_sites + site);
      } else {  // sources, outgroup: the counts
        const uint32_t none[1][4] = {{0u, 0u, 0u, 0u}}, none_own[1] = {0u};
        uint32_t unused[1];
        cnt = sad_population<0>(base, n_ind, lane, none, none_own, unused);
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
