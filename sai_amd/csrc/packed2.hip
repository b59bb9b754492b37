// packed2: the optional 2-bit genotype layout.

#include "site_eval.hpp"
#include "stream_loops.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// packed2: an optional 4x denser layout for dosages in {0, 1, 2} (+ missing), i.e. unphased
// diploid (or haploid) biallelic calls -- SURVEY.md section 8f #4.  2 bits per individual
// (0, 1, 2 = dosage, 3 = missing), blocked so that one lane owns one site:
//     tile t = 64 consecutive sites; the individuals form n_full = n_ind / 64 full groups of 64
//     (16 bytes per site: a 1 KiB block per tile, site-major) and one TAIL group of the remaining
//     n_ind % 64, held in w_tail = ceil(rem / 16) words per site (a 256 * w_tail byte block): a
//     2-individual source population costs 4 bytes per site, not 16;
//     words per tile W = n_full * 256 + w_tail * 64;
//     field(site, ind) = bits [2*(ind%16), +2) of uint32 word
//         (site/64) * W + (ind/64) * 256 + (site%64) * 4 + (ind%64)/16        in a full group,
//         (site/64) * W + n_full * 256 + (site%64) * w_tail + (ind%64)/16     in the tail group.
// A wave instruction reads one block: lane l gets the 64 individuals of site l of the tile, counts
// the three codes with v_bcnt_u32_b32 (popcount with accumulate) and keeps the totals in its own
// registers across the groups -- no cross-lane step at all, and the per-site tail (eval_site) runs
// on values the lane already holds.  Results are identical to the int8 path; the algorithmic
// bytes are 4x fewer, so this is reported as a separate roofline, never mixed with the int8
// numbers.  Padding individuals carry code 0 (n_called = n_ind - missing), padding sites code 3.
// ------------------------------------------------------------------------------------------

__host__ __device__ __forceinline__ int packed2_full_groups(int n_ind) { return n_ind / 64; }
__host__ __device__ __forceinline__ int packed2_tail_words(int n_ind) { return ((n_ind % 64) + 15) / 16; }
__host__ __device__ __forceinline__ int64_t packed2_tile_words(int n_ind) {
  return static_cast<int64_t>(packed2_full_groups(n_ind)) * 256 + packed2_tail_words(n_ind) * 64;
}

constexpr int kPackedUnroll = 8;  // wave loads in flight per batch (16 measured: the same 0.81-0.84 ms)
constexpr int kPackedMaxInd = 1 << 24;  // as for the int8 layout: per-site totals are 32-bit

// tiled int8 -> packed2.  One workgroup per (tile, group); n_bad counts words holding a byte above 2.
__global__ __launch_bounds__(256) void pack2_from_tiles_kernel(const int8_t* __restrict__ tiles, int64_t n_sites,
                                                                int32_t n_ind,
                                                                uint32_t* __restrict__ packed, int32_t* n_bad) {
  __shared__ int8_t blk[kTile][kTile + 4];  // [individual of the group][site]
  const int64_t tile = blockIdx.x;
  const int ind0 = blockIdx.y * kTile;
  const int tid = threadIdx.x;
  {
    const int i = tid >> 2, part = tid & 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ind0 + i < n_ind) v = *reinterpret_cast<const uint4*>(tiles + (tile * n_ind + ind0 + i) * kTile + part * 16);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) blk[i][part * 16 + j * 4 + k] = static_cast<int8_t>((w[j] >> (8 * k)) & 0xFF);
  }
  __syncthreads();
  const int s = tid >> 2, j = tid & 3;  // site in tile, 16-individual word of the group
  uint32_t w = 0xFFFFFFFFu;             // padding sites of the last tile: all missing
  bool bad = false;
  if (tile * kTile + s < n_sites) {
    w = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int g = blk[j * 16 + k][s];
      bad = bad || g > 2;
      w |= static_cast<uint32_t>(g < 0 ? 3 : (g & 3)) << (2 * k);
    }
  }
  const int n_full = packed2_full_groups(n_ind), w_tail = packed2_tail_words(n_ind);
  uint32_t* tile_words = packed + tile * packed2_tile_words(n_ind);
  if (static_cast<int>(blockIdx.y) < n_full) tile_words[blockIdx.y * 256 + s * 4 + j] = w;  // a 1 KiB block, in order
  else if (j < w_tail) tile_words[n_full * 256 + s * w_tail + j] = w;                       // the narrow tail block
  if (bad) atomicAdd(n_bad, 1);
}

struct PackedPop {
  const uint32_t* data;
  int32_t n_ind;
  int32_t n_full;   // full groups of 64 individuals
  int32_t w_tail;   // words per site of the tail group (0..4)
  int32_t pad;
};

struct PackedArgs {
  int64_t n_sites;
  int64_t n_tiles;
  int32_t n_pops;
  PackedPop pop[kMaxPops];
  uint2* counts;
};

__device__ __forceinline__ void count_codes(const u32x4& v, uint32_t& ones, uint32_t& twos, uint32_t& miss) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = w[j] & 0x55555555u, hi = (w[j] >> 1) & 0x55555555u;
    ones += __popc(lo & ~hi);
    twos += __popc(hi & ~lo);
    miss += __popc(lo & hi);
  }
}

template <bool FUSED>
__global__ __launch_bounds__(64) void site_counts_packed2_kernel(PackedArgs a, FusedArgs fa) {
  // indexed per-lane storage only: entry [p][lane] is written and read by the same lane
  __shared__ uint2 stash[kMaxPops][64];
  __shared__ uint32_t table_words[FUSED ? sizeof(PredTable) / 4 : 1];
  PredTable* table = reinterpret_cast<PredTable*>(table_words);
  if (FUSED) {
    stage_pred_table(fa.es, table);
    wave_lds_fence();
  }
  const int lane = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t site = tile * kTile + lane;
    for (int p = 0; p < a.n_pops; ++p) {
      const int n_groups = a.pop[p].n_full, w_tail = a.pop[p].w_tail;
      const uint32_t* tile_words = a.pop[p].data + tile * (static_cast<int64_t>(n_groups) * 256 + w_tail * 64);
      const u32x4* base = reinterpret_cast<const u32x4*>(tile_words) + lane;
      uint32_t ones = 0, twos = 0, miss = 0;
      int g = 0;
      for (; g + kPackedUnroll <= n_groups; g += kPackedUnroll) {
        u32x4 v[kPackedUnroll];
#pragma unroll
        for (int u = 0; u < kPackedUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (g + u) * kTile);
#pragma unroll
        for (int u = 0; u < kPackedUnroll; ++u) count_codes(v[u], ones, twos, miss);
      }
      // the tail group's words of this lane's site go out together with the remaining full groups
      u32x4 tail = {0u, 0u, 0u, 0u};
      if (w_tail) {
        const uint32_t* tw = tile_words + n_groups * 256 + lane * w_tail;
        tail.x = __builtin_nontemporal_load(tw);
        if (w_tail > 1) tail.y = __builtin_nontemporal_load(tw + 1);
        if (w_tail > 2) tail.z = __builtin_nontemporal_load(tw + 2);
        if (w_tail > 3) tail.w = __builtin_nontemporal_load(tw + 3);  // 49..63 individuals
      }
      if (g + 1 == n_groups) {  // one full group left: a single load
        count_codes(__builtin_nontemporal_load(base + g * kTile), ones, twos, miss);
      } else if (g < n_groups) {  // 2 .. kPackedUnroll-1 groups as one batch: clamped addresses, zeroed extras
        u32x4 v[kPackedUnroll - 1];
#pragma unroll
        for (int u = 0; u < kPackedUnroll - 1; ++u) v[u] = __builtin_nontemporal_load(base + min(g + u, n_groups - 1) * kTile);
#pragma unroll
        for (int u = 0; u < kPackedUnroll - 1; ++u) {
          if (g + u >= n_groups) v[u] = u32x4{0u, 0u, 0u, 0u};
          count_codes(v[u], ones, twos, miss);
        }
      }
      count_codes(tail, ones, twos, miss);  // padding individuals carry code 0
      const uint2 cnt = make_uint2(ones + 2u * twos, static_cast<uint32_t>(a.pop[p].n_ind) - miss);
      if (a.counts && site < a.n_sites) store_counts_nt(a.counts + static_cast<int64_t>(p) * a.n_sites + site, cnt);
      if (FUSED) stash[p][lane] = cnt;
    }
    if (FUSED)  // lane = site inside the tile already: the ballots of eval_site are the tile's flag planes
      eval_site<kMaxPops>(
          a.n_pops, fa.ploidy, [&](int p) { return stash[p][lane]; }, fa.n_sets, fa.es, table, tile, lane, site < a.n_sites,
          a.n_sites, fa.tgt_freq, fa.planes, fa.plane_stride, nullptr, fa.sparse_freq != 0, fa.with_inv != 0);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int64_t sai_packed2_bytes(int64_t n_sites, int32_t n_ind) {
  if (n_sites < 0 || n_ind < 0 || n_ind > kPackedMaxInd) return -1;
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  return n_tiles * packed2_tile_words(n_ind) * 4;  // 1 KiB per full group + 256 B per word of the tail group
}

int sai_pack2_from_tiles(sai_ctx* ctx, const int8_t* tiles, int64_t n_sites, int32_t n_ind, uint8_t* packed,
                         int32_t* n_unrepresentable, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_ind < 1 || n_ind > kPackedMaxInd) return fail(SAI_ERR_UNSUPPORTED, "packed2 supports 1..%d individuals", kPackedMaxInd);
  if (!n_unrepresentable) return fail(SAI_ERR_ARG, "n_unrepresentable is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  SAI_HIP(hipMemsetAsync(n_unrepresentable, 0, sizeof(int32_t), st));
  if (n_sites == 0) return SAI_OK;
  if (!tiles || !packed) return fail(SAI_ERR_ARG, "NULL buffer");
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  const dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>((n_ind + kTile - 1) / kTile));
  hipLaunchKernelGGL(pack2_from_tiles_kernel, grid, dim3(256), 0, st, tiles, n_sites, n_ind,
                     reinterpret_cast<uint32_t*>(packed), n_unrepresentable);
  return check_launch("pack2_from_tiles");
}

int sai_site_pass_packed2(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                          int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                          uint64_t* planes, int64_t plane_stride, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (freq_mode != SAI_FREQ_DENSE && freq_mode != SAI_FREQ_CANDIDATES) return fail(SAI_ERR_ARG, "bad freq_mode %d", freq_mode);
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 1 || n_pops > kMaxPops) return fail(SAI_ERR_ARG, "n_pops must be 1..%d", kMaxPops);
  if (!pops) return fail(SAI_ERR_ARG, "pops is NULL");
  if (n_sets < 0 || n_sets > kFusedSets) return fail(SAI_ERR_UNSUPPORTED, "at most %d parameter sets", kFusedSets);
  if (n_sets > 0) {
    if (n_pops < 2) return fail(SAI_ERR_ARG, "n_pops must be >= 2 (ref, tgt, sources)");
    if (int rc = check_sets(n_sets, sets_host, n_pops - 2, kFusedSets)) return rc;
    if (n_sites > 0 && (!tgt_freq || !planes)) return fail(SAI_ERR_ARG, "NULL buffer");
    if (int rc = check_plane_stride(plane_stride, n_sets)) return rc;
  } else if (!counts && n_sites > 0) {
    return fail(SAI_ERR_ARG, "nothing to compute: no parameter sets and counts is NULL");
  }
  if (n_sites == 0) return SAI_OK;
  PackedArgs a;
  FusedArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  a.n_sites = n_sites;
  a.n_tiles = (n_sites + kTile - 1) / kTile;
  a.n_pops = n_pops;
  for (int p = 0; p < n_pops; ++p) {
    if (pops[p].n_ind < 1 || pops[p].n_ind > kPackedMaxInd)
      return fail(SAI_ERR_UNSUPPORTED, "population %d: packed2 supports 1..%d individuals", p, kPackedMaxInd);
    if (!pops[p].tiles || (reinterpret_cast<uintptr_t>(pops[p].tiles) & 15u))
      return fail(SAI_ERR_ARG, "population %d: packed block must be a 16-byte aligned device pointer", p);
    if (n_sets > 0 && pops[p].ploidy <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
    a.pop[p].data = reinterpret_cast<const uint32_t*>(pops[p].tiles);
    a.pop[p].n_ind = pops[p].n_ind;
    a.pop[p].n_full = packed2_full_groups(pops[p].n_ind);
    a.pop[p].w_tail = packed2_tail_words(pops[p].n_ind);
    a.pop[p].pad = 0;
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
  if (n_sets > 0) launch_pass(ctx, site_counts_packed2_kernel<true>, grid, dim3(64), st, a, fa);
  else launch_pass(ctx, site_counts_packed2_kernel<false>, grid, dim3(64), st, a, fa);
  return check_launch("site_counts_packed2");
}

}  // extern "C"
