// libsaihip: sai's sliding-window U/Q statistics as hand-written HIP for MI355X (gfx950, CDNA4).
//
// Kernels (see DESIGN.md for the roofline of each):
//   tile_from_site_major  ingest: reference-order [site][ind] int8 -> tiled SoA
//   site_counts           HBM-bound byte reduction: per site/pop {alt_sum, n_called}
//   site_flags            f64 frequencies + compute_matching_loci conditions per parameter set
//   window_bounds         window (start,end) -> site index range (binary search)
//   window_stats          U count, numpy-'linear' quantile (radix select), candidate lists
//   synth_*               counter-based synthetic genotype / position generator
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared (see __graft_entry__.py).
// -ffp-contract=off is part of the contract: the f64 arithmetic must round exactly like numpy's
// (separate multiply and add in the quantile lerp, IEEE division for the frequencies).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>

#include "saihip.h"

// ------------------------------------------------------------------------------------------
// errors / context
// ------------------------------------------------------------------------------------------

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define SAI_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(SAI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                    \
  } while (0)

constexpr int kTile = SAI_TILE_SITES;
constexpr int kMaxPops = 2 + SAI_MAX_SRC;

}  // namespace

struct sai_ctx {
  int device;
  int n_cu;
  uint32_t* probe_partials;  // n_cu * kProbeWavesPerCu words, the only scratch the library owns
};

constexpr int kProbeWavesPerCu = 32;
constexpr int kStreamWavesPerCu = 16;  // grid of the streaming site passes, see launch_site_counts

namespace {

int enter(sai_ctx* ctx) {
  if (!ctx) return fail(SAI_ERR_ARG, "ctx is NULL");
  SAI_HIP(hipSetDevice(ctx->device));
  return SAI_OK;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SAI_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return SAI_OK;
}

// ------------------------------------------------------------------------------------------
// ingest: [site][ind] -> tiled SoA.  One 256-thread workgroup moves a 64-site x 64-individual
// block through LDS (the transpose of 64-byte rows).
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void tile_from_site_major_kernel(const int8_t* __restrict__ src,
                                                                    int64_t n_sites, int32_t n_ind,
                                                                    int64_t row_stride,
                                                                    int8_t* __restrict__ dst) {
  __shared__ int8_t blk[kTile][kTile + 4];
  const int64_t tile = blockIdx.x;
  const int ind0 = blockIdx.y * kTile;
  const int tid = threadIdx.x;
  {
    const int s = tid >> 2;        // site in tile
    const int part = tid & 3;      // 16 individuals
    const int64_t site = tile * kTile + s;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ind = ind0 + part * 16 + k;
      int8_t v = 0;
      if (site < n_sites && ind < n_ind) v = src[site * row_stride + ind];
      blk[s][part * 16 + k] = v;
    }
  }
  __syncthreads();
  {
    const int i = tid >> 2;        // individual in block
    const int part = tid & 3;      // 16 sites
    const int ind = ind0 + i;
    if (ind < n_ind) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v |= static_cast<uint32_t>(static_cast<uint8_t>(blk[part * 16 + j * 4 + k][i])) << (8 * k);
        w[j] = v;
      }
      uint4* out = reinterpret_cast<uint4*>(dst + (tile * n_ind + ind) * kTile + part * 16);
      *out = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

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

// get(p) -> uint2 {alt_sum, n_called} of population p at this site.
template <typename GetCounts>
__device__ __forceinline__ void eval_site(int n_pops, const int32_t* ploidy, GetCounts get, int n_sets,
                                          const sai_params* sets, int64_t site, int64_t n_sites, double* tgt_freq,
                                          uint8_t* flags, double* adj_freq, bool sparse_freq = false) {
  double f[kMaxPops];
  bool valid = true;
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
  bool any_cond = false;
  const int n_src = n_pops - 2;
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
    const double tf = inverted ? 1.0 - f[1] : f[1];
    const bool cond = valid && hit && (rf < ps.w);
    const bool ucand = cond && (tf > ps.x);
    any_cond = any_cond || cond;
    // non-temporal stores: measured on MI355X, plain stores in the middle of the genotype stream cost
    // twice as much of the pass as streaming ones
    __builtin_nontemporal_store(static_cast<uint8_t>((cond ? 1 : 0) | (ucand ? 2 : 0) | (inverted ? 4 : 0)),
                                flags + static_cast<int64_t>(s) * n_sites + site);
    if (adj_freq) {
      adj_freq[(static_cast<int64_t>(s) * 2 + 0) * n_sites + site] = rf;
      adj_freq[(static_cast<int64_t>(s) * 2 + 1) * n_sites + site] = tf;
    }
  }
  // The windows stage reads tgt_freq only where a set's bit 0 is up (about 1 site in 1000), and
  // dense f64 stores in the middle of the genotype stream cost ~10 % of the pass (HBM read/write
  // turnarounds): SAI_FREQ_CANDIDATES leaves every other entry untouched.
  if (!sparse_freq || any_cond) __builtin_nontemporal_store(f[1], tgt_freq + site);
}

struct FlagArgs {
  int64_t n_sites;
  int32_t n_pops;
  int32_t n_sets;
  int32_t ploidy[kMaxPops];
  const uint2* counts;
  double* tgt_freq;
  uint8_t* flags;
  double* adj_freq;
  sai_params sets[SAI_MAX_SETS];
};

__global__ __launch_bounds__(256) void site_flags_kernel(FlagArgs a) {
  const int64_t site = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (site >= a.n_sites) return;
  eval_site(
      a.n_pops, a.ploidy, [&](int p) { return a.counts[static_cast<int64_t>(p) * a.n_sites + site]; }, a.n_sets, a.sets,
      site, a.n_sites, a.tgt_freq, a.flags, a.adj_freq);
}

constexpr int kFusedSets = 4;  // parameter sets the fused tail of site_counts can carry in its arguments

struct FusedArgs {
  int32_t n_sets;  // 0 = plain site_counts
  int32_t sparse_freq;
  int32_t ploidy[kMaxPops];
  double* tgt_freq;
  uint8_t* flags;
  sai_params sets[kFusedSets];
};

// ------------------------------------------------------------------------------------------
// site_counts: the HBM-bound kernel.
//
// One wavefront owns one 64-site tile and streams every population's rows of that tile.  A wave
// instruction loads 16 rows x 64 B = 1 KiB contiguous: lane l holds individual (16*q + l/4),
// sites (l%4)*16 .. +15 as four 32-bit words.  Bytes are accumulated SWAR-style into 16-bit
// (dosage) and 8-bit (missing) fields, widened to 32 bit every <= 248 rows per lane, and the 16
// row-groups are combined with a 4-step butterfly reduce-scatter so that lane l ends with the
// totals of site (l%4)*16 + l/4.  No LDS, no barriers; occupancy and 4-8 KiB of loads in flight
// per wave hide HBM latency.
// ------------------------------------------------------------------------------------------

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_counts_nt(uint2* dst, uint2 v) {
  __builtin_nontemporal_store(u32x2{v.x, v.y}, reinterpret_cast<u32x2*>(dst));
}

struct PopArg {
  const int8_t* tiles;
  int32_t n_ind;
  int32_t pad;
};

struct CountsArgs {
  int64_t n_sites;
  int64_t n_tiles;
  int32_t n_pops;
  PopArg pop[kMaxPops];
  uint2* counts;
};

__device__ __forceinline__ void acc_word(uint32_t w, uint32_t& lo, uint32_t& hi, uint32_t& ms) {
  const uint32_t neg = w & 0x80808080u;   // sign bit of each byte
  const uint32_t m1 = neg >> 7;           // 0x01 per missing call
  ms += m1;
  const uint32_t mask = (neg - m1) | neg; // 0xFF per missing call
  const uint32_t val = w & ~mask;         // max(g, 0) per byte
  lo += val & 0x00FF00FFu;                // sites 0,2 of the word -> 16-bit fields
  hi += (val >> 8) & 0x00FF00FFu;         // sites 1,3
}

__device__ __forceinline__ void acc_vec(const u32x4& v, uint32_t (&lo)[4], uint32_t (&hi)[4],
                                        uint32_t (&ms)[4]) {
  acc_word(v.x, lo[0], hi[0], ms[0]);
  acc_word(v.y, lo[1], hi[1], ms[1]);
  acc_word(v.z, lo[2], hi[2], ms[2]);
  acc_word(v.w, lo[3], hi[3], ms[3]);
}

template <int N, int MASK>
__device__ __forceinline__ void reduce_scatter_step(uint32_t (&a)[16], int lane) {
  constexpr int H = N / 2;
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const uint32_t send = up ? a[k] : a[k + H];
    const uint32_t keep = up ? a[k + H] : a[k];
    a[k] = keep + __shfl_xor(send, MASK, 64);
  }
}

constexpr int kChunkIters = 248;  // iterations (rows per lane) the 16-/8-bit fields can absorb

// Accumulate iterations [it, full_end) of full 16-row groups plus, when it is the last one, the
// partial group, into the packed fields lo/hi (16-bit dosage sums) and ms (8-bit missing counts).
constexpr int kUnroll = 4;  // wave loads (1 KiB each) in flight per group
static_assert(kChunkIters % kUnroll == 0 && kChunkIters + kUnroll <= 255, "8-bit missing fields overflow");

// Accumulate iterations [it, full_end) of full 16-row groups plus, when it is the last one, the
// partial group, into the packed fields lo/hi (16-bit dosage sums) and ms (8-bit missing counts).
// Loads are non-temporal: every genotype byte is read exactly once.  (Measured alternatives --
// 8 loads per group, ping-pong prefetch of the next group, default cache policy, 4 waves per
// workgroup, 64-register builds with 8 waves per SIMD -- all landed within 2 % of this form: the
// kernel sits at the rate a plain streaming read reaches on the same box.)
__device__ __forceinline__ void accumulate_rows(const u32x4* base, int& it, int full_end, int n_full, int n_iter,
                                                int n_ind, int r, uint32_t (&lo)[4], uint32_t (&hi)[4],
                                                uint32_t (&ms)[4]) {
  for (; it + kUnroll <= full_end; it += kUnroll) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) acc_vec(v[u], lo, hi, ms);
  }
  // Tail: the < kUnroll remaining full groups and the partial group go out as ONE batch of
  // unconditional loads (addresses clamped into the tile, invalid lanes zeroed afterwards), so the
  // wave pays one memory latency for the tail instead of one per group; a tail of one group (small
  // source populations) is a single load.
  const int last = (full_end == n_full) ? n_iter : full_end;
  if (it + 1 == last) {
    const int row = it * 16 + r;
    u32x4 v = __builtin_nontemporal_load(base + (min(row, n_ind - 1) - r) * 4);
    if (row >= n_ind) v = u32x4{0u, 0u, 0u, 0u};
    acc_vec(v, lo, hi, ms);
    it = last;
  } else if (it < last) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int row = min(it + u, last - 1) * 16 + r;
      v[u] = __builtin_nontemporal_load(base + (min(row, n_ind - 1) - r) * 4);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const bool valid = (it + u < last) && ((it + u) * 16 + r < n_ind);
      if (!valid) v[u] = u32x4{0u, 0u, 0u, 0u};
      acc_vec(v[u], lo, hi, ms);
    }
    it = last;
  }
}

__device__ __forceinline__ void widen_fields(const uint32_t (&lo)[4], const uint32_t (&hi)[4], const uint32_t (&ms)[4],
                                             uint32_t (&sum32)[16], uint32_t (&miss32)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sum32[4 * j + 0] += lo[j] & 0xFFFFu;
    sum32[4 * j + 1] += hi[j] & 0xFFFFu;
    sum32[4 * j + 2] += lo[j] >> 16;
    sum32[4 * j + 3] += hi[j] >> 16;
    miss32[4 * j + 0] += ms[j] & 0xFFu;
    miss32[4 * j + 1] += (ms[j] >> 8) & 0xFFu;
    miss32[4 * j + 2] += (ms[j] >> 16) & 0xFFu;
    miss32[4 * j + 3] += ms[j] >> 24;
  }
}

// MULTI: some population has more than 16 * kChunkIters individuals, so the packed fields are
// widened several times per population (keeps 32 more registers live across the load loop).
// FUSED: evaluate the parameter sets at the end of each tile (site_flags folded in).
template <bool MULTI, bool FUSED>
__global__ __launch_bounds__(64) void site_counts_kernel(CountsArgs a, FusedArgs fa) {
  // FUSED: each lane parks its site's {alt_sum, n_called} per population here and evaluates the
  // parameter sets itself once all populations of the tile are done (only the lane that wrote a
  // slot reads it back, so no synchronisation is involved)
  __shared__ uint2 stash[FUSED ? kMaxPops : 1][FUSED ? 64 : 1];
  const int lane = threadIdx.x;
  const int r = lane >> 2;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    for (int p = 0; p < a.n_pops; ++p) {
      const int n_ind = a.pop[p].n_ind;
      const u32x4* base =
          reinterpret_cast<const u32x4*>(a.pop[p].tiles + tile * static_cast<int64_t>(n_ind) * kTile) + lane;
      const int n_full = n_ind >> 4;         // iterations in which all 16 rows exist
      const int n_iter = (n_ind + 15) >> 4;  // plus at most one partial iteration
      uint32_t sum32[16], miss32[16];
      int it = 0;
      if (MULTI) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sum32[j] = miss32[j] = 0;
        while (it < n_iter) {
          uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
          accumulate_rows(base, it, min(n_full, it + kChunkIters), n_full, n_iter, n_ind, r, lo, hi, ms);
          widen_fields(lo, hi, ms, sum32, miss32);
        }
      } else {  // n_iter <= kChunkIters + 1: one pass, widen once
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
        accumulate_rows(base, it, n_full, n_full, n_iter, n_ind, r, lo, hi, ms);
#pragma unroll
        for (int j = 0; j < 16; ++j) sum32[j] = miss32[j] = 0;
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
      const int64_t site = tile * kTile + (lane & 3) * 16 + r;
      const uint2 cnt = make_uint2(sum32[0], static_cast<uint32_t>(n_ind) - miss32[0]);
      if (a.counts && site < a.n_sites) store_counts_nt(a.counts + static_cast<int64_t>(p) * a.n_sites + site, cnt);
      if (FUSED) stash[p][lane] = cnt;
    }
    if (FUSED) {
      const int64_t site = tile * kTile + (lane & 3) * 16 + r;
      if (site < a.n_sites)
        eval_site(
            a.n_pops, fa.ploidy, [&](int p) { return stash[p][lane]; }, fa.n_sets, fa.sets, site,
            a.n_sites, fa.tgt_freq, fa.flags, nullptr, fa.sparse_freq != 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// packed2: an optional 4x denser layout for dosages in {0, 1, 2} (+ missing), i.e. unphased
// diploid (or haploid) biallelic calls -- SURVEY.md section 8f #4.  2 bits per individual
// (0, 1, 2 = dosage, 3 = missing), blocked so that one lane owns one site:
//     tile t = 64 consecutive sites, group g = 64 consecutive individuals (16 bytes per site);
//     block (t, g) = 64 sites x 16 B = 1 KiB at byte offset (t * n_groups + g) * 1024, site-major;
//     field(site, ind) = bits [2*(ind%16), +2) of uint32 word
//                        ((site/64 * n_groups + ind/64) * 64 + site%64) * 4 + (ind%64)/16.
// A wave instruction reads one block: lane l gets the 64 individuals of site l of the tile, counts
// the three codes with v_bcnt_u32_b32 (popcount with accumulate) and keeps the totals in its own
// registers across the groups -- no cross-lane step at all, and the per-site tail (eval_site) runs
// on values the lane already holds.  Results are identical to the int8 path; the algorithmic
// bytes are 4x fewer, so this is reported as a separate roofline, never mixed with the int8
// numbers.  Padding individuals carry code 0 (n_called = n_ind - missing), padding sites code 3.
// ------------------------------------------------------------------------------------------

__host__ __device__ __forceinline__ int packed2_groups(int n_ind) { return (n_ind + 63) / 64; }

constexpr int kPackedUnroll = 8;  // wave loads in flight per batch
constexpr int kPackedMaxInd = 1 << 24;  // as for the int8 layout: per-site totals are 32-bit

// tiled int8 -> packed2.  One workgroup per (tile, group); n_bad counts words holding a byte above 2.
__global__ __launch_bounds__(256) void pack2_from_tiles_kernel(const int8_t* __restrict__ tiles, int64_t n_sites,
                                                                int32_t n_ind, int32_t n_groups,
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
  packed[((tile * n_groups + blockIdx.y) * kTile + s) * 4 + j] = w;  // the workgroup writes its 1 KiB block in order
  if (bad) atomicAdd(n_bad, 1);
}

struct PackedPop {
  const u32x4* data;
  int32_t n_ind;
  int32_t n_groups;
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
  const int lane = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t site = tile * kTile + lane;
    for (int p = 0; p < a.n_pops; ++p) {
      const int n_groups = a.pop[p].n_groups;
      const u32x4* base = a.pop[p].data + tile * n_groups * kTile + lane;
      uint32_t ones = 0, twos = 0, miss = 0;
      int g = 0;
      for (; g + kPackedUnroll <= n_groups; g += kPackedUnroll) {
        u32x4 v[kPackedUnroll];
#pragma unroll
        for (int u = 0; u < kPackedUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (g + u) * kTile);
#pragma unroll
        for (int u = 0; u < kPackedUnroll; ++u) count_codes(v[u], ones, twos, miss);
      }
      if (g + 1 == n_groups) {  // one group left (small source populations): a single load
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
      const uint2 cnt = make_uint2(ones + 2u * twos, static_cast<uint32_t>(a.pop[p].n_ind) - miss);
      if (a.counts && site < a.n_sites) store_counts_nt(a.counts + static_cast<int64_t>(p) * a.n_sites + site, cnt);
      if (FUSED) stash[p][lane] = cnt;
    }
    if (FUSED && site < a.n_sites)
      eval_site(
          a.n_pops, fa.ploidy, [&](int p) { return stash[p][lane]; }, fa.n_sets, fa.sets, site, a.n_sites, fa.tgt_freq,
          fa.flags, nullptr, fa.sparse_freq != 0);
  }
}

// ------------------------------------------------------------------------------------------
// window_bounds
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void window_bounds_kernel(const int32_t* __restrict__ pos,
                                                             int64_t n_sites, int32_t n_windows,
                                                             const int64_t* __restrict__ ws,
                                                             const int64_t* __restrict__ we,
                                                             int32_t* __restrict__ lo,
                                                             int32_t* __restrict__ hi) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_windows) return;
  const int64_t s = ws[w], e = we[w];
  int64_t a = 0, b = n_sites;  // first index with pos >= s
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (static_cast<int64_t>(pos[m]) < s) a = m + 1; else b = m;
  }
  const int64_t first = a;
  b = n_sites;                 // first index with pos > e
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (static_cast<int64_t>(pos[m]) <= e) a = m + 1; else b = m;
  }
  lo[w] = static_cast<int32_t>(first);
  hi[w] = static_cast<int32_t>(a < first ? first : a);
}

// ------------------------------------------------------------------------------------------
// window statistics.  Four launches, no atomics on shared words (a returning atomic on one
// address saturates at ~90 per microsecond, which at one reservation per window was 85 % of the
// old single-kernel version):
//   window_stats_wave   one WAVEFRONT per (window, set): U count, condition count, Q, Q-list size
//   window_stats_heavy  workgroup fallback for windows with > kWaveCap qualifying sites
//   window_scan         exclusive prefix sums of the list sizes -> CSR offsets + totals
//   window_lists        one wavefront per (window, set): candidate lists in ascending site order
// ------------------------------------------------------------------------------------------

struct WinArgs {
  int64_t n_sites;
  const double* tgt_freq;
  const uint8_t* flags;
  int32_t n_sets;
  int32_t n_windows;
  const int32_t* lo;
  const int32_t* hi;
  const int32_t* pos;
  sai_window_record* records;
  int64_t* cdd_off;
  int32_t* cdd_u;
  int64_t cap_u;
  int32_t* cdd_q;
  int64_t cap_q;
  int64_t* cdd_total;
  double quantile[SAI_MAX_SETS];
};

constexpr int kWinThreads = 256;
constexpr int kWaveCap = 256;       // qualifying sites a wave keeps in LDS; more -> heavy kernel
constexpr int kSelCap = 4096;       // values the heavy kernel keeps in LDS (32 KiB); beyond: re-read
constexpr int32_t kHeavyMark = -1;  // records[].n_cdd_q value that hands a window to the fallback

__device__ __forceinline__ double eff_freq(const double* tgt_freq, uint8_t f, int64_t i) {
  const double v = tgt_freq[i];
  return (f & 4) ? 1.0 - v : v;
}

// numpy 'linear' quantile from the two neighbouring order statistics (numpy _quantile/_lerp):
// virtual index v = (n-1)*q; a + (b-a)*g, or b - (b-a)*(1-g) when g >= 0.5.
__device__ __forceinline__ double numpy_lerp(double x0, double x1, double v, double fl_v) {
  const double g = v - fl_v;
  const double d = x1 - x0;
  return (g >= 0.5) ? x1 - d * (1.0 - g) : x0 + d * g;
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving
  // accesses across the point where lanes start reading what other lanes of the wave wrote.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware block order: consecutive blockIdx values round-robin over the 8 XCDs; give each XCD a
// contiguous run of (overlapping) windows so their shared sites stay in one L2.
__device__ __forceinline__ int xcd_contiguous(int b, int n_blocks) {
  const int per = n_blocks >> 3;
  return (per > 0 && b < per * 8) ? (b & 7) * per + (b >> 3) : b;
}

__global__ __launch_bounds__(256) void window_stats_wave_kernel(WinArgs a) {
  __shared__ double sh_vals[4][kWaveCap];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int w = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wv;
  if (w >= a.n_windows) return;  // whole wave
  const int set = blockIdx.y;
  const int lo = a.lo[w], hi = a.hi[w];
  const uint8_t* fl = a.flags + static_cast<int64_t>(set) * a.n_sites;
  const int64_t ridx = static_cast<int64_t>(set) * a.n_windows + w;
  double* vals = sh_vals[wv];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  // pass 1: counts + compaction of the qualifying effective frequencies into LDS
  uint32_t n_c = 0, n_u = 0;
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const uint8_t f = i < hi ? fl[i] : static_cast<uint8_t>(0);
    const bool c = (f & 1u) != 0;
    const unsigned long long bc = __ballot(c);
    if (c) {
      const uint32_t slot = n_c + __popcll(bc & lt_mask);
      if (slot < kWaveCap) vals[slot] = eff_freq(a.tgt_freq, f, i);
    }
    n_c += __popcll(bc);
    n_u += __popcll(__ballot((f & 2u) != 0));
  }
  double q = std::numeric_limits<double>::quiet_NaN();
  uint32_t n_q = 0;
  if (n_c > kWaveCap) {
    n_q = static_cast<uint32_t>(kHeavyMark);  // uniform: the workgroup kernel finishes this window
  } else if (n_c > 0) {
    wave_lds_fence();
    const double v = static_cast<double>(n_c - 1) * a.quantile[set];
    const bool take_max = v >= static_cast<double>(n_c - 1);  // at/after the last index: maximum
    const double fl_v = floor(v);
    const uint32_t k0 = take_max ? n_c - 1 : static_cast<uint32_t>(fl_v);
    const uint32_t k1 = take_max ? n_c - 1 : k0 + 1;
    double x0 = 0.0, x1 = 0.0;
    // rank counting: rank(e) = #{j: v_j < v_e or (v_j == v_e and j < e)} is a permutation
    for (uint32_t e0 = 0; e0 < n_c; e0 += 64) {
      const uint32_t e = e0 + lane;
      const bool act = e < n_c;
      const double ve = act ? vals[e] : 0.0;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n_c; ++j) {
        const double vj = vals[j];  // same address in every lane: LDS broadcast
        rank += (vj < ve) || (vj == ve && j < e);
      }
      const unsigned long long h0 = __ballot(act && rank == k0);
      const unsigned long long h1 = __ballot(act && rank == k1);
      if (h0) x0 = __shfl(ve, __ffsll(static_cast<long long>(h0)) - 1, 64);
      if (h1) x1 = __shfl(ve, __ffsll(static_cast<long long>(h1)) - 1, 64);
    }
    q = take_max ? x0 : numpy_lerp(x0, x1, v, fl_v);
    for (uint32_t e0 = 0; e0 < n_c; e0 += 64) {
      const uint32_t e = e0 + lane;
      n_q += __popcll(__ballot(e < n_c && vals[e] >= q));
    }
  }
  if (lane == 0) {
    sai_window_record rec;
    rec.n_sites = hi - lo;
    rec.u_count = static_cast<int32_t>(n_u);
    rec.n_cond = static_cast<int32_t>(n_c);
    rec.n_cdd_q = static_cast<int32_t>(n_q);
    rec.q = q;
    a.records[ridx] = rec;
  }
}

// ---- heavy fallback ------------------------------------------------------------------------

struct WinShared {
  double vals[kSelCap];
  uint32_t hist[256];
  uint32_t wave_tot[4];
  uint32_t red[4];
  uint32_t n_stored;
  uint32_t digit;
  uint32_t k_rem;
};

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// k-th smallest (0-based) of the selected values: MSB-first radix select on the f64 bit pattern
// (all selected values are finite and >= 0, so the unsigned order of the bits is the numeric order).
template <bool IN_LDS>
__device__ double select_kth(WinShared& sh, const double* tgt_freq, const uint8_t* fl, int lo, int hi,
                             uint32_t n_sel, uint32_t k, int tid) {
  unsigned long long prefix = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    sh.hist[tid] = 0;
    __syncthreads();
    const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
    if (IN_LDS) {
      for (uint32_t i = tid; i < n_sel; i += kWinThreads) {
        const unsigned long long key = __double_as_longlong(sh.vals[i]);
        if ((key & himask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1u);
      }
    } else {
      for (int i = lo + tid; i < hi; i += kWinThreads) {
        const uint8_t f = fl[i];
        if (f & 1) {
          const unsigned long long key = __double_as_longlong(eff_freq(tgt_freq, f, i));
          if ((key & himask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1u);
        }
      }
    }
    __syncthreads();
    // inclusive scan of the 256 bins: shuffles inside each wave, wave totals through LDS
    const uint32_t h = sh.hist[tid];
    uint32_t inc = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if ((tid & 63) >= o) inc += t;
    }
    if ((tid & 63) == 63) sh.wave_tot[tid >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) before += sh.wave_tot[wv];
    inc += before;
    const uint32_t exc = inc - h;
    if (h != 0 && k >= exc && k < inc) {  // exactly one bin satisfies this
      sh.digit = tid;
      sh.k_rem = k - exc;
    }
    __syncthreads();
    prefix |= static_cast<unsigned long long>(sh.digit) << shift;
    k = sh.k_rem;
    __syncthreads();
  }
  return __longlong_as_double(static_cast<long long>(prefix));
}

// The grid covers every window; those not marked by the wave kernel exit at once.
__global__ __launch_bounds__(kWinThreads) void window_stats_heavy_kernel(WinArgs a) {
  __shared__ WinShared sh;
  const int tid = threadIdx.x;
  const int w = blockIdx.x;
  const int set = blockIdx.y;
  const int64_t ridx = static_cast<int64_t>(set) * a.n_windows + w;
  if (a.records[ridx].n_cdd_q != kHeavyMark) return;  // uniform over the workgroup
  const int lo = a.lo[w], hi = a.hi[w];
  const uint8_t* fl = a.flags + static_cast<int64_t>(set) * a.n_sites;
  const uint32_t n_c = static_cast<uint32_t>(a.records[ridx].n_cond);
  const bool in_lds = n_c <= kSelCap;
  if (tid == 0) sh.n_stored = 0;
  __syncthreads();
  if (in_lds) {
    for (int i = lo + tid; i < hi; i += kWinThreads) {
      const uint8_t f = fl[i];
      if (f & 1u) sh.vals[atomicAdd(&sh.n_stored, 1u)] = eff_freq(a.tgt_freq, f, i);
    }
  }
  __syncthreads();
  const double v = static_cast<double>(n_c - 1) * a.quantile[set];
  double q;
  if (v >= static_cast<double>(n_c - 1)) {
    q = in_lds ? select_kth<true>(sh, a.tgt_freq, fl, lo, hi, n_c, n_c - 1, tid)
               : select_kth<false>(sh, a.tgt_freq, fl, lo, hi, n_c, n_c - 1, tid);
  } else {
    const double fl_v = floor(v);
    const uint32_t k = static_cast<uint32_t>(fl_v);
    const double x0 = in_lds ? select_kth<true>(sh, a.tgt_freq, fl, lo, hi, n_c, k, tid)
                             : select_kth<false>(sh, a.tgt_freq, fl, lo, hi, n_c, k, tid);
    const double x1 = in_lds ? select_kth<true>(sh, a.tgt_freq, fl, lo, hi, n_c, k + 1, tid)
                             : select_kth<false>(sh, a.tgt_freq, fl, lo, hi, n_c, k + 1, tid);
    q = numpy_lerp(x0, x1, v, fl_v);
  }
  uint32_t c_q = 0;
  for (int i = lo + tid; i < hi; i += kWinThreads) {
    const uint8_t f = fl[i];
    if ((f & 1u) && eff_freq(a.tgt_freq, f, i) >= q) ++c_q;
  }
  const uint32_t n_q = block_sum(c_q, sh.red, tid);
  if (tid == 0) {
    a.records[ridx].n_cdd_q = static_cast<int32_t>(n_q);
    a.records[ridx].q = q;
  }
}

// ---- CSR offsets ---------------------------------------------------------------------------

// One 1024-thread workgroup: exclusive prefix sums of u_count and n_cdd_q over the records in
// (set, window) order.  cdd_off[2r] / cdd_off[2r+1] = start of record r's U / Q list (or -1 when
// the list would not fit its buffer); cdd_total[0..1] = entries needed in all.  Records are taken
// 4 x 1024 at a time (coalesced, all loads of a batch in flight together); each row of 1024 is
// scanned with wave shuffles + one LDS hop, the running totals carry over in registers.
__global__ __launch_bounds__(1024) void window_scan_kernel(WinArgs a) {
  __shared__ long long wave_tot[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = static_cast<int64_t>(a.n_sets) * a.n_windows;
  long long carry_u = 0, carry_q = 0;
  constexpr int kBatch = 4;
  for (int64_t base = 0; base < n; base += 1024 * kBatch) {
    long long nu[kBatch], nq[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t r = base + k * 1024 + tid;
      nu[k] = r < n ? a.records[r].u_count : 0;
      nq[k] = r < n ? a.records[r].n_cdd_q : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (base + k * 1024 >= n) break;  // uniform
      long long iu = nu[k], iq = nq[k];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long long tu = __shfl_up(iu, o, 64), tq = __shfl_up(iq, o, 64);
        if (lane >= o) { iu += tu; iq += tq; }
      }
      __syncthreads();  // the previous row's wave totals have been consumed
      if (lane == 63) { wave_tot[0][wave] = iu; wave_tot[1][wave] = iq; }
      __syncthreads();
      long long before_u = 0, before_q = 0, all_u = 0, all_q = 0;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const long long tu = wave_tot[0][v], tq = wave_tot[1][v];
        if (v < wave) { before_u += tu; before_q += tq; }
        all_u += tu;
        all_q += tq;
      }
      const int64_t r = base + k * 1024 + tid;
      if (r < n) {
        const long long ou = carry_u + before_u + iu - nu[k], oq = carry_q + before_q + iq - nq[k];
        a.cdd_off[2 * r + 0] = (ou + nu[k] <= a.cap_u) ? ou : -1;
        a.cdd_off[2 * r + 1] = (oq + nq[k] <= a.cap_q) ? oq : -1;
      }
      carry_u += all_u;
      carry_q += all_q;
    }
  }
  if (tid == 0) {
    a.cdd_total[0] = carry_u;
    a.cdd_total[1] = carry_q;
  }
}

// ---- candidate lists -----------------------------------------------------------------------

__global__ __launch_bounds__(256) void window_lists_kernel(WinArgs a) {
  const int lane = threadIdx.x & 63;
  const int w = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (w >= a.n_windows) return;
  const int set = blockIdx.y;
  const int64_t ridx = static_cast<int64_t>(set) * a.n_windows + w;
  const sai_window_record rec = a.records[ridx];
  const long long off_u = a.cdd_off[2 * ridx + 0], off_q = a.cdd_off[2 * ridx + 1];
  const bool write_u = rec.u_count > 0 && off_u >= 0 && a.cdd_u != nullptr;
  const bool write_q = rec.n_cdd_q > 0 && off_q >= 0 && a.cdd_q != nullptr;
  if (!write_u && !write_q) return;
  const int lo = a.lo[w], hi = a.hi[w];
  const uint8_t* fl = a.flags + static_cast<int64_t>(set) * a.n_sites;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const double q = rec.q;
  uint32_t done_u = 0, done_q = 0;
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const uint8_t f = i < hi ? fl[i] : static_cast<uint8_t>(0);
    const bool pu = (f & 2u) != 0;
    const bool pq = write_q && (f & 1u) && eff_freq(a.tgt_freq, f, i) >= q;
    const unsigned long long mu = __ballot(pu);
    const unsigned long long mq = __ballot(pq);
    if (mu | mq) {
      const int32_t out = (pu || pq) ? (a.pos ? a.pos[i] : i) : 0;
      if (write_u && pu) a.cdd_u[off_u + done_u + __popcll(mu & lt_mask)] = out;
      if (pq) a.cdd_q[off_q + done_q + __popcll(mq & lt_mask)] = out;
      done_u += __popcll(mu);
      done_q += __popcll(mq);
    }
  }
}

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

// numpy's *_pairwise_sum for n <= 128 (loops_utils.h.src): plain loop below 8 elements, else eight
// running sums, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail.
template <typename E>
__device__ __forceinline__ double pairwise_leaf(const E& e, int off, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += e(off + i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = e(off + j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += e(off + i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += e(off + i);
  return res;
}

// recursion of numpy's pairwise sum unrolled to the depth an 8192-element piece can reach
template <int DEPTH, typename E>
struct PairwiseNode {
  static __device__ __noinline__ double run(const E& e, int off, int n) {
    if (n <= 128) return pairwise_leaf(e, off, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return PairwiseNode<DEPTH - 1, E>::run(e, off, n2) + PairwiseNode<DEPTH - 1, E>::run(e, off + n2, n - n2);
  }
};
template <typename E>
struct PairwiseNode<0, E> {
  static __device__ __noinline__ double run(const E& e, int off, int n) { return pairwise_leaf(e, off, n); }
};

// np.sum of elements e(off .. off+n): 8192-element pieces (the ufunc buffer), added in order
template <typename E>
__device__ __forceinline__ double numpy_sum(const E& e, int off, int n) {
  double res = 0.0;
  for (int o = 0; o < n; o += 8192) res += PairwiseNode<6, E>::run(e, off + o, min(8192, n - o));
  return res;
}

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
  const int lo = a.lo[w], hi = a.hi[w];
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

// ------------------------------------------------------------------------------------------
// synth-v1: counter-based synthetic data (identical on host and device)
// ------------------------------------------------------------------------------------------

#define SAI_HD __host__ __device__ __forceinline__

SAI_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

SAI_HD uint64_t stream_key(uint64_t seed, int32_t chrom, int32_t stream) {
  return mix64(seed ^ (static_cast<uint64_t>(static_cast<uint32_t>(chrom)) << 40) ^
               (static_cast<uint64_t>(static_cast<uint32_t>(stream)) << 32));
}

SAI_HD int32_t synth_gap(uint64_t seed, int32_t chrom, int64_t site) {
  const uint64_t h = mix64(stream_key(seed, chrom, 0) + static_cast<uint64_t>(site));
  return 1 + static_cast<int32_t>(static_cast<uint32_t>(h >> 32) % 49u);
}

struct SiteModel {
  uint64_t key;        // per (site, population stream) hashing key
  uint32_t threshold;  // allele is ALT when a 32-bit uniform < threshold
  int32_t fixed;       // -1: draw; else dosage forced to this value
};

// pop_stream: 0 = ref, 1 = tgt, >= 2 = sources
SAI_HD SiteModel site_model(uint64_t seed, int32_t chrom, int64_t site, int32_t pop_stream, int32_t ploidy) {
  const uint64_t hs = mix64(stream_key(seed, chrom, 1) + static_cast<uint64_t>(site));
  const bool intro = (static_cast<uint32_t>(hs >> 32) % 1000u) == 0u;
  const double u = static_cast<double>(static_cast<uint32_t>(hs)) * (1.0 / 4294967296.0);
  SiteModel m;
  m.key = mix64(stream_key(seed, chrom, 2 + pop_stream) + static_cast<uint64_t>(site));
  m.fixed = -1;
  double p;
  if (intro) {
    if (pop_stream == 0) { m.fixed = 0; p = 0.0; }
    else if (pop_stream == 1) { p = 0.2 + 0.7 * u; }
    else { m.fixed = ploidy; p = 1.0; }
  } else {
    const double u2 = u * u;
    p = u2 * u2;
  }
  m.threshold = static_cast<uint32_t>(p * 4294967296.0 >= 4294967295.0 ? 4294967295.0 : p * 4294967296.0);
  return m;
}

SAI_HD int8_t synth_genotype(const SiteModel& m, int32_t ind, int32_t ploidy, uint32_t miss_thr) {
  if (miss_thr != 0u) {
    const uint64_t hm = mix64(m.key ^ 0xD1B54A32D192ED03ull ^ (static_cast<uint64_t>(static_cast<uint32_t>(ind)) << 1));
    if (static_cast<uint32_t>(hm >> 32) < miss_thr) return static_cast<int8_t>(-ploidy);
  }
  if (m.fixed >= 0) return static_cast<int8_t>(m.fixed);
  int d = 0;
  for (int a = 0; a < ploidy; a += 2) {
    const uint64_t h = mix64(m.key + static_cast<uint64_t>(static_cast<uint32_t>(ind)) +
                             (static_cast<uint64_t>(a >> 1) << 32));
    d += static_cast<uint32_t>(h) < m.threshold;
    if (a + 1 < ploidy) d += static_cast<uint32_t>(h >> 32) < m.threshold;
  }
  return static_cast<int8_t>(d);
}

SAI_HD uint32_t miss_threshold(int32_t missing_per_million) {
  return static_cast<uint32_t>((static_cast<uint64_t>(missing_per_million) << 32) / 1000000ull);
}

__global__ __launch_bounds__(256) void synth_fill_kernel(uint64_t seed, int32_t chrom, int64_t site0,
                                                          int64_t n_sites, int32_t pop_stream, int32_t n_ind,
                                                          int32_t ploidy, uint32_t miss_thr, int8_t* tiles) {
  __shared__ SiteModel models[kTile];
  const int64_t tile = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < kTile) models[tid] = site_model(seed, chrom, site0 + tile * kTile + tid, pop_stream, ploidy);
  __syncthreads();
  const int part = tid & 3;
  for (int ind = tid >> 2; ind < n_ind; ind += 64) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int s = part * 16 + j * 4 + k;
        int8_t g = 0;
        if (tile * kTile + s < n_sites) g = synth_genotype(models[s], ind, ploidy, miss_thr);
        v |= static_cast<uint32_t>(static_cast<uint8_t>(g)) << (8 * k);
      }
      w[j] = v;
    }
    *reinterpret_cast<uint4*>(tiles + (tile * n_ind + ind) * kTile + part * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__global__ __launch_bounds__(256) void synth_gaps_kernel(uint64_t seed, int32_t chrom, int64_t site0,
                                                          int64_t n_sites, int32_t* gaps) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_sites) gaps[i] = synth_gap(seed, chrom, site0 + i);
}


// ------------------------------------------------------------------------------------------
// stream_read probe: the plainest streaming read in the access pattern site_counts uses -- one
// wave per workgroup walks contiguous 125 KiB runs, 8 non-temporal 1 KiB wave loads in flight,
// XOR-reduced to one word.  It is the on-box read ceiling the site_counts rate is compared with
// (a thread-strided grid loop reads ~8 % slower on MI355X than per-wave contiguous runs).
// ------------------------------------------------------------------------------------------

constexpr int64_t kProbeRunVecs = 8000;  // 125 KiB, the size of one C3 tile (ref + tgt rows)

__global__ __launch_bounds__(64) void stream_read_kernel(const u32x4* __restrict__ src, int64_t n_vec,
                                                          uint32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  const int64_t n_runs = n_vec / kProbeRunVecs;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int64_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
    const u32x4* base = src + run * kProbeRunVecs + lane;
    for (int it = 0; it < kProbeRunVecs / 64; it += 5) {  // 125 wave loads in 25 groups of 5
      u32x4 v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
#pragma unroll
      for (int u = 0; u < 5; ++u) acc ^= v[u];
    }
  }
  for (int64_t i = n_runs * kProbeRunVecs + static_cast<int64_t>(blockIdx.x) * 64 + lane; i < n_vec;
       i += static_cast<int64_t>(gridDim.x) * 64)
    acc ^= __builtin_nontemporal_load(src + i);
  uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if (lane == 0) out[blockIdx.x] = v;  // one word per wave; thousands of atomics on one address would
                                       // add ~5 % to the time this kernel exists to measure
}

__global__ __launch_bounds__(256) void stream_read_fold_kernel(const uint32_t* __restrict__ partials, int n,
                                                                uint32_t* __restrict__ xor_out) {
  __shared__ uint32_t sh[4];
  uint32_t v = 0;
  for (int i = threadIdx.x; i < n; i += 256) v ^= partials[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *xor_out ^= sh[0] ^ sh[1] ^ sh[2] ^ sh[3];
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

// shared with vcf_ingest.cpp (not part of the public header)
int sai_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int sai_abi_version(void) { return SAI_ABI_VERSION; }
const char* sai_build_arch(void) { return "gfx950"; }
const char* sai_last_error(void) { return g_err; }

int sai_device_count(int* count_out) {
  if (!count_out) return fail(SAI_ERR_ARG, "count_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count_out = 0;
    return fail(SAI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count_out = n;
  return SAI_OK;
}

int sai_ctx_create(int device, sai_ctx** ctx_out) {
  if (!ctx_out) return fail(SAI_ERR_ARG, "ctx_out is NULL");
  *ctx_out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(SAI_ERR_NO_DEVICE, "no HIP device is visible; libsaihip has no CPU fallback");
  if (device < 0 || device >= n) return fail(SAI_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
  SAI_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SAI_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SAI_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
  sai_ctx* c = new (std::nothrow) sai_ctx;
  if (!c) return fail(SAI_ERR_HIP, "out of host memory");
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  c->probe_partials = nullptr;
  if (hipMalloc(&c->probe_partials, sizeof(uint32_t) * c->n_cu * kProbeWavesPerCu) != hipSuccess) {
    delete c;
    return fail(SAI_ERR_HIP, "hipMalloc of the context scratch failed");
  }
  *ctx_out = c;
  return SAI_OK;
}

int sai_ctx_destroy(sai_ctx* ctx) {
  if (ctx && ctx->probe_partials) (void)hipFree(ctx->probe_partials);
  delete ctx;
  return SAI_OK;
}

int64_t sai_tiled_bytes(int64_t n_sites, int32_t n_ind) {
  if (n_sites < 0 || n_ind < 0) return -1;
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  return n_tiles * static_cast<int64_t>(n_ind) * kTile;
}

int sai_tile_from_site_major(sai_ctx* ctx, const int8_t* src, int64_t n_sites, int32_t n_ind,
                             int64_t row_stride, int8_t* dst, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_ind < 0) return fail(SAI_ERR_ARG, "negative size");
  if (n_sites == 0 || n_ind == 0) return SAI_OK;
  if (!src || !dst) return fail(SAI_ERR_ARG, "NULL buffer");
  if (row_stride < n_ind) return fail(SAI_ERR_ARG, "row_stride %lld < n_ind %d", (long long)row_stride, n_ind);
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  const int64_t n_blk = (static_cast<int64_t>(n_ind) + kTile - 1) / kTile;
  if (n_tiles > 0x7FFFFFFFll || n_blk > 65535) return fail(SAI_ERR_UNSUPPORTED, "block too large for one launch");
  dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(n_blk));
  hipLaunchKernelGGL(tile_from_site_major_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     n_sites, n_ind, row_stride, dst);
  return check_launch("tile_from_site_major");
}

static int check_sets(int32_t n_sets, const sai_params* sets, int32_t n_src);

// shared by sai_site_counts (n_sets == 0) and sai_site_pass
static int launch_site_counts(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                              int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                              uint8_t* flags, void* stream) {
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
  fa.tgt_freq = tgt_freq;
  fa.flags = flags;
  for (int s = 0; s < n_sets; ++s) fa.sets[s] = sets_host[s];
  // 16 single-wave workgroups per CU = 4 waves per SIMD: enough to saturate HBM (measured flat from
  // 8 to 611 per CU) while leaving registers and LDS on every SIMD for the small kernels that the
  // pipelined scorer runs on a second stream under this one
  const int64_t max_grid = static_cast<int64_t>(ctx->n_cu) * kStreamWavesPerCu;
  const dim3 grid(static_cast<unsigned>(a.n_tiles < max_grid ? a.n_tiles : max_grid));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_sets > 0) {
    if (multi) hipLaunchKernelGGL((site_counts_kernel<true, true>), grid, dim3(64), 0, st, a, fa);
    else hipLaunchKernelGGL((site_counts_kernel<false, true>), grid, dim3(64), 0, st, a, fa);
  } else {
    if (multi) hipLaunchKernelGGL((site_counts_kernel<true, false>), grid, dim3(64), 0, st, a, fa);
    else hipLaunchKernelGGL((site_counts_kernel<false, false>), grid, dim3(64), 0, st, a, fa);
  }
  return check_launch("site_counts");
}

int sai_site_counts(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                    void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (!counts && n_sites > 0) return fail(SAI_ERR_ARG, "counts is NULL");
  return launch_site_counts(ctx, n_sites, n_pops, pops, counts, 0, nullptr, SAI_FREQ_DENSE, nullptr, nullptr, stream);
}

int sai_site_pass(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                  int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq, uint8_t* flags,
                  void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (freq_mode != SAI_FREQ_DENSE && freq_mode != SAI_FREQ_CANDIDATES) return fail(SAI_ERR_ARG, "bad freq_mode %d", freq_mode);
  if (n_pops < 2) return fail(SAI_ERR_ARG, "n_pops must be >= 2 (ref, tgt, sources)");
  if (n_sets > kFusedSets)
    return fail(SAI_ERR_UNSUPPORTED, "sai_site_pass carries at most %d parameter sets; use sai_site_counts + sai_site_flags",
                kFusedSets);
  if (int rc = check_sets(n_sets, sets_host, n_pops - 2)) return rc;
  if (pops)
    for (int p = 0; p < n_pops && p < kMaxPops; ++p)
      if (pops[p].ploidy <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
  if (n_sites > 0 && (!tgt_freq || !flags)) return fail(SAI_ERR_ARG, "NULL buffer");
  return launch_site_counts(ctx, n_sites, n_pops, pops, counts, n_sets, sets_host, freq_mode, tgt_freq, flags, stream);
}

static int check_sets(int32_t n_sets, const sai_params* sets, int32_t n_src) {
  if (n_sets < 1 || n_sets > SAI_MAX_SETS) return fail(SAI_ERR_ARG, "n_sets must be 1..%d", SAI_MAX_SETS);
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

int sai_site_flags(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const int32_t* ploidy_host,
                   const uint32_t* counts, int32_t n_sets, const sai_params* sets_host, double* tgt_freq,
                   uint8_t* flags, double* adj_freq, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 2 || n_pops > kMaxPops) return fail(SAI_ERR_ARG, "n_pops must be 2..%d (ref, tgt, sources)", kMaxPops);
  if (!ploidy_host) return fail(SAI_ERR_ARG, "ploidy_host is NULL");
  if (int rc = check_sets(n_sets, sets_host, n_pops - 2)) return rc;
  if (n_sites == 0) return SAI_OK;
  if (!counts || !tgt_freq || !flags) return fail(SAI_ERR_ARG, "NULL buffer");
  FlagArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_sites = n_sites;
  a.n_pops = n_pops;
  a.n_sets = n_sets;
  for (int p = 0; p < n_pops; ++p) {
    if (ploidy_host[p] <= 0) return fail(SAI_ERR_ARG, "ploidy[%d] must be positive", p);
    a.ploidy[p] = ploidy_host[p];
  }
  a.counts = reinterpret_cast<const uint2*>(counts);
  a.tgt_freq = tgt_freq;
  a.flags = flags;
  a.adj_freq = adj_freq;
  for (int s = 0; s < n_sets; ++s) a.sets[s] = sets_host[s];
  const unsigned grid = static_cast<unsigned>((n_sites + 255) / 256);
  hipLaunchKernelGGL(site_flags_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("site_flags");
}

int sai_window_bounds(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                      const int64_t* win_start, const int64_t* win_end, int32_t* lo, int32_t* hi, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll || n_windows < 0) return fail(SAI_ERR_ARG, "size out of range");
  if (n_windows == 0) return SAI_OK;
  if ((n_sites > 0 && !pos) || !win_start || !win_end || !lo || !hi) return fail(SAI_ERR_ARG, "NULL buffer");
  const unsigned grid = static_cast<unsigned>((n_windows + 255) / 256);
  hipLaunchKernelGGL(window_bounds_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), pos, n_sites,
                     n_windows, win_start, win_end, lo, hi);
  return check_launch("window_bounds");
}

int sai_window_stats(sai_ctx* ctx, int64_t n_sites, const double* tgt_freq, const uint8_t* flags, int32_t n_sets,
                     const sai_params* sets_host, int32_t n_windows, const int32_t* lo, const int32_t* hi,
                     const int32_t* pos, sai_window_record* records, int64_t* cdd_off, int32_t* cdd_u, int64_t cap_u,
                     int32_t* cdd_q, int64_t cap_q, int64_t* cdd_total, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll || n_windows < 0) return fail(SAI_ERR_ARG, "size out of range");
  if (int rc = check_sets(n_sets, sets_host, -1)) return rc;
  if (!cdd_total) return fail(SAI_ERR_ARG, "cdd_total is NULL");
  if (cap_u < 0 || cap_q < 0 || (cap_u > 0 && !cdd_u) || (cap_q > 0 && !cdd_q))
    return fail(SAI_ERR_ARG, "candidate buffers do not match their capacities");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_windows == 0) {
    SAI_HIP(hipMemsetAsync(cdd_total, 0, 2 * sizeof(int64_t), st));
    return SAI_OK;
  }
  if ((n_sites > 0 && (!tgt_freq || !flags)) || !lo || !hi || !records || !cdd_off)
    return fail(SAI_ERR_ARG, "NULL buffer");
  WinArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_sites = n_sites;
  a.tgt_freq = tgt_freq;
  a.flags = flags;
  a.n_sets = n_sets;
  a.n_windows = n_windows;
  a.lo = lo;
  a.hi = hi;
  a.pos = pos;
  a.records = records;
  a.cdd_off = cdd_off;
  a.cdd_u = cdd_u;
  a.cap_u = cap_u;
  a.cdd_q = cdd_q;
  a.cap_q = cap_q;
  a.cdd_total = cdd_total;
  for (int s = 0; s < n_sets; ++s) a.quantile[s] = sets_host[s].quantile;
  const dim3 wave_grid(static_cast<unsigned>((n_windows + 3) / 4), static_cast<unsigned>(n_sets));
  const dim3 block_grid(static_cast<unsigned>(n_windows), static_cast<unsigned>(n_sets));
  hipLaunchKernelGGL(window_stats_wave_kernel, wave_grid, dim3(256), 0, st, a);
  if (int rc = check_launch("window_stats_wave")) return rc;
  hipLaunchKernelGGL(window_stats_heavy_kernel, block_grid, dim3(kWinThreads), 0, st, a);
  if (int rc = check_launch("window_stats_heavy")) return rc;
  hipLaunchKernelGGL(window_scan_kernel, dim3(1), dim3(1024), 0, st, a);
  if (int rc = check_launch("window_scan")) return rc;
  hipLaunchKernelGGL(window_lists_kernel, wave_grid, dim3(256), 0, st, a);
  return check_launch("window_lists");
}

int sai_synth_fill(sai_ctx* ctx, uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t pop_stream,
                   int32_t n_ind, int32_t ploidy, int32_t missing_per_million, int8_t* tiles, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || site0 < 0 || n_ind < 0 || pop_stream < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (ploidy < 1 || ploidy > 8) return fail(SAI_ERR_ARG, "ploidy must be 1..8");
  if (missing_per_million < 0 || missing_per_million > 1000000) return fail(SAI_ERR_ARG, "missing_per_million out of range");
  if (n_sites == 0 || n_ind == 0) return SAI_OK;
  if (!tiles) return fail(SAI_ERR_ARG, "NULL buffer");
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  if (n_tiles > 0x7FFFFFFFll) return fail(SAI_ERR_UNSUPPORTED, "too many tiles for one launch");
  hipLaunchKernelGGL(synth_fill_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy,
                     miss_threshold(missing_per_million), tiles);
  return check_launch("synth_fill");
}

int sai_synth_fill_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t pop_stream,
                        int32_t n_ind, int32_t ploidy, int32_t missing_per_million, int8_t* out) {
  if (n_sites < 0 || site0 < 0 || n_ind < 0 || pop_stream < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (ploidy < 1 || ploidy > 8) return fail(SAI_ERR_ARG, "ploidy must be 1..8");
  if (missing_per_million < 0 || missing_per_million > 1000000) return fail(SAI_ERR_ARG, "missing_per_million out of range");
  if (n_sites == 0 || n_ind == 0) return SAI_OK;
  if (!out) return fail(SAI_ERR_ARG, "NULL buffer");
  const uint32_t mt = miss_threshold(missing_per_million);
  for (int64_t s = 0; s < n_sites; ++s) {
    const SiteModel m = site_model(seed, chrom, site0 + s, pop_stream, ploidy);
    int8_t* row = out + s * n_ind;
    for (int32_t i = 0; i < n_ind; ++i) row[i] = synth_genotype(m, i, ploidy, mt);
  }
  return SAI_OK;
}

int sai_synth_gaps_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t* gaps) {
  if (n_sites < 0 || site0 < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (n_sites > 0 && !gaps) return fail(SAI_ERR_ARG, "NULL buffer");
  for (int64_t i = 0; i < n_sites; ++i) gaps[i] = synth_gap(seed, chrom, site0 + i);
  return SAI_OK;
}

int sai_synth_gaps(sai_ctx* ctx, uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t* gaps,
                   void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || site0 < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (n_sites == 0) return SAI_OK;
  if (!gaps) return fail(SAI_ERR_ARG, "NULL buffer");
  const unsigned grid = static_cast<unsigned>((n_sites + 255) / 256);
  hipLaunchKernelGGL(synth_gaps_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), seed, chrom,
                     site0, n_sites, gaps);
  return check_launch("synth_gaps");
}

int sai_probe_stream_read(sai_ctx* ctx, const void* buf, int64_t n_bytes, uint32_t* xor_out, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_bytes < 0 || (n_bytes & 15)) return fail(SAI_ERR_ARG, "n_bytes must be a non-negative multiple of 16");
  if (!xor_out || (n_bytes > 0 && !buf)) return fail(SAI_ERR_ARG, "NULL buffer");
  if (reinterpret_cast<uintptr_t>(buf) & 15u) return fail(SAI_ERR_ARG, "buf must be 16-byte aligned");
  if (n_bytes == 0) return SAI_OK;
  const unsigned grid = static_cast<unsigned>(ctx->n_cu) * kProbeWavesPerCu;
  hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const u32x4*>(buf), n_bytes / 16, ctx->probe_partials);
  hipLaunchKernelGGL(stream_read_fold_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                     ctx->probe_partials, static_cast<int>(grid), xor_out);
  return check_launch("stream_read");
}

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
  const int64_t max_grid = static_cast<int64_t>(ctx->n_cu) * kStreamWavesPerCu;
  const dim3 grid(static_cast<unsigned>(a.n_tiles < max_grid ? a.n_tiles : max_grid));
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

int64_t sai_packed2_bytes(int64_t n_sites, int32_t n_ind) {
  if (n_sites < 0 || n_ind < 0 || n_ind > kPackedMaxInd) return -1;
  const int64_t n_tiles = (n_sites + kTile - 1) / kTile;
  return n_tiles * packed2_groups(n_ind) * 1024;  // whole 1 KiB blocks: 64 sites x 64 individuals
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
  hipLaunchKernelGGL(pack2_from_tiles_kernel, grid, dim3(256), 0, st, tiles, n_sites, n_ind, packed2_groups(n_ind),
                     reinterpret_cast<uint32_t*>(packed), n_unrepresentable);
  return check_launch("pack2_from_tiles");
}

int sai_site_pass_packed2(sai_ctx* ctx, int64_t n_sites, int32_t n_pops, const sai_pop* pops, uint32_t* counts,
                          int32_t n_sets, const sai_params* sets_host, int32_t freq_mode, double* tgt_freq,
                          uint8_t* flags, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (freq_mode != SAI_FREQ_DENSE && freq_mode != SAI_FREQ_CANDIDATES) return fail(SAI_ERR_ARG, "bad freq_mode %d", freq_mode);
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll) return fail(SAI_ERR_ARG, "n_sites out of range");
  if (n_pops < 1 || n_pops > kMaxPops) return fail(SAI_ERR_ARG, "n_pops must be 1..%d", kMaxPops);
  if (!pops) return fail(SAI_ERR_ARG, "pops is NULL");
  if (n_sets < 0 || n_sets > kFusedSets) return fail(SAI_ERR_UNSUPPORTED, "at most %d parameter sets", kFusedSets);
  if (n_sets > 0) {
    if (n_pops < 2) return fail(SAI_ERR_ARG, "n_pops must be >= 2 (ref, tgt, sources)");
    if (int rc = check_sets(n_sets, sets_host, n_pops - 2)) return rc;
    if (n_sites > 0 && (!tgt_freq || !flags)) return fail(SAI_ERR_ARG, "NULL buffer");
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
    a.pop[p].data = reinterpret_cast<const u32x4*>(pops[p].tiles);
    a.pop[p].n_ind = pops[p].n_ind;
    a.pop[p].n_groups = packed2_groups(pops[p].n_ind);
    fa.ploidy[p] = pops[p].ploidy;
  }
  a.counts = reinterpret_cast<uint2*>(counts);
  fa.n_sets = n_sets;
  fa.sparse_freq = freq_mode == SAI_FREQ_CANDIDATES;
  fa.tgt_freq = tgt_freq;
  fa.flags = flags;
  for (int s = 0; s < n_sets; ++s) fa.sets[s] = sets_host[s];
  const int64_t max_grid = static_cast<int64_t>(ctx->n_cu) * kStreamWavesPerCu;
  const dim3 grid(static_cast<unsigned>(a.n_tiles < max_grid ? a.n_tiles : max_grid));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_sets > 0) hipLaunchKernelGGL((site_counts_packed2_kernel<true>), grid, dim3(64), 0, st, a, fa);
  else hipLaunchKernelGGL((site_counts_packed2_kernel<false>), grid, dim3(64), 0, st, a, fa);
  return check_launch("site_counts_packed2");
}

}  // extern "C"
