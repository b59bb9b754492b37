// The streaming loop of the int8 tiled layout, shared by site_counts and site_absdiff: 16-byte
// non-temporal loads, SWAR byte accumulation, the butterfly reduce-scatter.
#pragma once

#include "common.hpp"

// kernel arguments of the int8 site pass (site_pass.hip, site_pass_dd.hip)
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
__device__ __forceinline__ void store_counts_nt(uint2* dst, uint2 v) {
  __builtin_nontemporal_store(u32x2{v.x, v.y}, reinterpret_cast<u32x2*>(dst));
}

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

// A group of wave loads at once.  Missing calls are what the reference's reader filters out by default
// (utils.py:175-179), and a dosage is rarely 64 or more: one OR over the group's 16 words shows (bit 7 / bit 6 of a
// byte) whether any lane met either.  If none did -- the wave votes -- the four rows add up BYTEWISE without a
// carry (4 x 63 = 252) and only the sum is split into the 16-bit fields: 9 vector operations per wave load instead
// of 44.  Otherwise the group goes through acc_vec as it is.  A narrow population's tile is made of the per-tile
// work either way; this is what a wide one's is made of.
constexpr int kGroup = 4;
__device__ __forceinline__ void acc_group(const u32x4 (&v)[kGroup], uint32_t (&lo)[4], uint32_t (&hi)[4],
                                          uint32_t (&ms)[4]) {
  uint32_t seen = 0;
#pragma unroll
  for (int u = 0; u < kGroup; ++u) seen |= v[u].x | v[u].y | v[u].z | v[u].w;
  if (__ballot((seen & 0xC0C0C0C0u) != 0u) == 0ull) {  // wave-uniform
    const uint32_t s[4] = {v[0].x + v[1].x + v[2].x + v[3].x, v[0].y + v[1].y + v[2].y + v[3].y,
                           v[0].z + v[1].z + v[2].z + v[3].z, v[0].w + v[1].w + v[2].w + v[3].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] += s[j] & 0x00FF00FFu;
      hi[j] += (s[j] >> 8) & 0x00FF00FFu;
    }
  } else {
#pragma unroll
    for (int u = 0; u < kGroup; ++u) acc_vec(v[u], lo, hi, ms);
  }
}

// one wave load: no missing call in it (the wave votes) and its bytes are the dosages themselves
__device__ __forceinline__ void acc_one(const u32x4& v, uint32_t (&lo)[4], uint32_t (&hi)[4], uint32_t (&ms)[4]) {
  if (__ballot(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0u) == 0ull) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] += w[j] & 0x00FF00FFu;
      hi[j] += (w[j] >> 8) & 0x00FF00FFu;
    }
  } else {
    acc_vec(v, lo, hi, ms);
  }
}

template <int N, int MASK>
__device__ __forceinline__ void reduce_scatter_step(uint32_t (&a)[16], int lane) {
  constexpr int H = N / 2;
  if constexpr (MASK == 32 || MASK == 16) {
    // gfx950's half-swaps do the exchange of a pair in one operation: v_permlane32_swap trades the upper 32 lanes of its
    // first operand for the lower 32 of its second (v_permlane16_swap: odd 16-lane rows for even ones), after which
    // the two registers hold, lane by lane, exactly the two values this lane has to add
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const auto pair = MASK == 32 ? __builtin_amdgcn_permlane32_swap(a[k], a[k + H], false, false)
                                   : __builtin_amdgcn_permlane16_swap(a[k], a[k + H], false, false);
      a[k] = pair[0] + pair[1];
    }
    return;
  }
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const uint32_t send = up ? a[k] : a[k + H];
    const uint32_t keep = up ? a[k + H] : a[k];
    a[k] = keep + __shfl_xor(send, MASK, 64);
  }
}

constexpr int kChunkIters = 248;  // iterations (rows per lane) the 16-/8-bit fields can absorb
constexpr int kUnroll = 4;        // wave loads (1 KiB each) in flight per group
static_assert(kChunkIters % kUnroll == 0 && kChunkIters + kUnroll <= 255, "8-bit missing fields overflow");
static_assert(kUnroll == kGroup, "acc_group adds up kUnroll rows bytewise");

// Accumulate iterations [it, full_end) of full 16-row groups plus, when it is the last one, the
// partial group, into the packed fields lo/hi (16-bit dosage sums) and ms (8-bit missing counts).
// Loads are non-temporal: every genotype byte is read exactly once.  (Measured alternatives --
// 8 loads per group, ping-pong prefetch of the next group, default cache policy, 4 waves per
// workgroup, 64-register builds with 8 waves per SIMD -- all landed within 2 % of this form, and a
// default-policy load 8 % below it.  Groups of 8 loads were measured again on top of the bytewise sums of
// acc_group -- 125 registers: C3 0.866 against 0.877 of peak, C5 0.73 against 0.82, c2x22 0.72 against 0.79.)
__device__ __forceinline__ void accumulate_rows(const u32x4* base, int& it, int full_end, int n_full, int n_iter,
                                                int n_ind, int r, uint32_t (&lo)[4], uint32_t (&hi)[4],
                                                uint32_t (&ms)[4]) {
  for (; it + kUnroll <= full_end; it += kUnroll) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(base + (it + u) * 64);
    acc_group(v, lo, hi, ms);
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
    acc_one(v, lo, hi, ms);
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
    }
    acc_group(v, lo, hi, ms);
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
