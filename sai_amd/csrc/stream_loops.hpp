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
constexpr int kUnroll = 4;        // wave loads (1 KiB each) in flight per group
static_assert(kChunkIters % kUnroll == 0 && kChunkIters + kUnroll <= 255, "8-bit missing fields overflow");

// Accumulate iterations [it, full_end) of full 16-row groups plus, when it is the last one, the
// partial group, into the packed fields lo/hi (16-bit dosage sums) and ms (8-bit missing counts).
// Loads are non-temporal: every genotype byte is read exactly once.  (Measured alternatives --
// 8 loads per group, ping-pong prefetch of the next group, default cache policy, 4 waves per
// workgroup, 64-register builds with 8 waves per SIMD -- all landed within 2 % of this form, and a
// default-policy load 8 % below it: the kernel sits at the rate a plain streaming read reaches.)
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
