// Window stage: site-index ranges of the windows, U / Q records and candidate lists.

#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// window_bounds
// ------------------------------------------------------------------------------------------

// First index in [a, n) whose position is >= key (strict = false) or > key (strict = true); n when
// there is none.  G lanes search together (G+1)-ary: every step probes G evenly spaced positions,
// a ballot counts how many lie below the key, and the range shrinks (G+1)-fold.  G = 8 needs 8
// dependent loads for 10^7 sites where a binary search needs 24; wider groups need fewer steps
// but issue so many more probes that the texture path, not latency, bounds the kernel (G = 64:
// 4 steps, 33 us for 10^4 windows; G = 8: the fastest measured).
template <int G>
__device__ __forceinline__ int64_t group_search(const int32_t* __restrict__ pos, int64_t a, int64_t n, int64_t key,
                                                bool strict, int lane) {
  const int sub = lane % G, shift = lane - sub;  // lane within its group, first lane of the group
  int64_t b = n;                                 // the answer is in [a, b]
  while (__ballot(b > a) != 0ull) {              // groups finish after different numbers of steps
    const int64_t len = b - a;
    const bool active = len > 0, small = len <= G;
    const int64_t p = small ? a + sub : a + (len * (sub + 1)) / (G + 1);  // < b in both forms
    bool below = false;
    if (active && (!small || sub < len)) {
      const int64_t v = pos[p];
      below = strict ? v <= key : v < key;
    }
    // positions ascend: the probes below the key are a prefix of the group's lanes
    const int cnt = __popcll((__ballot(below) >> shift) & ((1ull << G) - 1ull));
    if (!active) continue;
    if (small) {
      a += cnt;
      b = a;
    } else {
      const int64_t na = cnt == 0 ? a : a + (len * cnt) / (G + 1) + 1;
      b = cnt == G ? b : a + (len * (cnt + 1)) / (G + 1);
      a = na;
    }
  }
  return a;
}

constexpr int kBoundsGroup = 8;      // lanes per search, many windows
constexpr int kBoundsWideGroup = 32;  // lanes per search, few windows: both searches of a window side by side in one wave
constexpr int kBoundsWideMax = 32768;

// lo = first site with pos >= start, hi = first site with pos > end, both searched inside the
// window's segment [seg_lo[w], seg_hi[w]) of the block (the whole block when seg_lo is NULL): a
// block that holds several chromosomes back to back has positions that ascend only per segment.
// PAIR = false: G lanes per window, lo then hi (hi from lo onward).  PAIR = true: 2 G lanes per window,
// one group per bound at the same time -- with G = 32 a wave per window and 4 dependent loads for 10^6
// sites instead of 14.  Stand-alone the narrow form is the faster one (fewer probes: the texture path
// bounds the kernel, 14 against 33 us for 10^4 windows), but the pipelined scorer runs this kernel under
// the next step's genotype stream, where every dependent load takes microseconds and the chain of the
// windows stage has to fit under a SHORT pass (C2: 75 us): there the wide form is used.
template <int G, bool PAIR>
__global__ __launch_bounds__(256) void window_bounds_kernel(const int32_t* __restrict__ pos,
                                                             int64_t n_sites, int32_t n_windows,
                                                             const int64_t* __restrict__ ws,
                                                             const int64_t* __restrict__ we,
                                                             const int32_t* __restrict__ seg_lo,
                                                             const int32_t* __restrict__ seg_hi,
                                                             int32_t* __restrict__ lo,
                                                             int32_t* __restrict__ hi) {
  const int lane = threadIdx.x & 63;
  const int slot = (blockIdx.x * 256 + threadIdx.x) / G;  // one search group
  const int w = PAIR ? slot >> 1 : slot;
  const bool upper = PAIR && (slot & 1);  // this group looks for hi
  const bool live = w < n_windows;  // dead groups search an empty range: no loads, no stores
  int64_t a = 0, b = live ? n_sites : 0;
  if (live && seg_lo) {  // clamped into the block, so a bad segment can never turn into a wild load
    a = min(max(static_cast<int64_t>(seg_lo[w]), int64_t{0}), n_sites);
    b = min(max(static_cast<int64_t>(seg_hi[w]), a), n_sites);
  }
  if (PAIR) {
    const int64_t at = group_search<G>(pos, a, b, live ? (upper ? we[w] : ws[w]) : 0, upper, lane);
    // hi is never below lo (a window whose end lies before its start is empty), as in the form that searches
    // hi from lo onward: the window's two groups sit side by side in the wave
    const int base = lane - lane % (2 * G);
    const int64_t first = __shfl(at, base, 64), last = __shfl(at, base + G, 64);
    if (live && lane == base) {
      lo[w] = static_cast<int32_t>(first);
      hi[w] = static_cast<int32_t>(last > first ? last : first);
    }
  } else {
    const int64_t first = group_search<G>(pos, a, b, live ? ws[w] : 0, false, lane);
    const int64_t last = group_search<G>(pos, first, live ? b : first, live ? we[w] : 0, true, lane);
    if (live && lane % G == 0) {
      lo[w] = static_cast<int32_t>(first);
      hi[w] = static_cast<int32_t>(last);
    }
  }
}

// ------------------------------------------------------------------------------------------
// window statistics.  Four launches, no atomics on shared words (a returning atomic on one
// address saturates at ~90 per microsecond, which at one reservation per window was 85 % of the
// first single-kernel version), and every window's data is read ONCE per kernel:
//   window_stats   >= 4 sets: one WORKGROUP per window -- the window's plane rows and stored target
//                  frequencies are brought into LDS once, each of the four waves then answers sets (U
//                  count, condition count, Q, Q-list size) from LDS on its own; fewer sets: one WAVE per
//                  window, every word read where it lies.  No workgroup barrier after the fill.
//   window_scan    exclusive prefix sums of the list sizes -> CSR offsets + totals (two launches)
//   window_lists   the same two forms: candidate lists in ascending site order
// Round 3 ran one wavefront per (window, set) and a second workgroup kernel for the heavy pairs, all
// reading global memory: C5's 18 sets x 2x window overlap x 64-byte lines re-read ~190 MB of planes
// and frequencies as 1.36 GB per step (profiles/history/r03g_c5_pmc_summary.csv), under the next step's
// genotype stream.
// ------------------------------------------------------------------------------------------

struct WinArgs {
  int64_t n_sites;
  const double* tgt_freq;
  const uint64_t* planes;  // flag planes of this call's sets (saihip.h): rows of `stride` words per tile
  int64_t stride;
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
  int32_t with_inv;  // the rows carry inverted words (some set lacks ancestral alleles)
  double quantile[SAI_MAX_SETS];
  double x[SAI_MAX_SETS];  // U's threshold on the effective target frequency (u_statistic.py:92)
};

constexpr int kWinThreads = 256;
constexpr int kWinWaves = kWinThreads / 64;
constexpr int kWaveCap = 256;  // qualifying sites a wave ranks directly; more -> digit histogram + one gathered bin
// Words of a tile's row (saihip.h): 0 = "any" (sites whose frequency is stored), 1 + s = condition of
// set s, 1 + n + s = inverted for set s (present only with a.with_inv).
constexpr int kAny = 0;
__device__ __forceinline__ int cond_word(int set) { return 1 + set; }
__device__ __forceinline__ int inv_word(const int32_t with_inv, int n_sets, int set) { return with_inv ? 1 + n_sets + set : -1; }

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

// lo / hi come from the caller's device arrays: clamped into the block, so that a bad range can never
// turn into a load beyond the planes or the per-site arrays
__device__ __forceinline__ int clamp_site(int v, int64_t n_sites) {
  return v < 0 ? 0 : (v > n_sites ? static_cast<int>(n_sites) : v);
}

// bits of tile t that lie inside the site range [lo, hi)
__device__ __forceinline__ uint64_t range_mask(int t, int lo, int hi) {
  const int64_t base = static_cast<int64_t>(t) * kTile;
  const int64_t a = lo > base ? lo - base : 0, b = hi - base < kTile ? hi - base : kTile;
  if (b <= a) return 0ull;
  const uint64_t below_b = b >= kTile ? ~0ull : (1ull << b) - 1ull;
  return below_b & (~0ull << a);
}

// word `v` of lane `src` (wave-uniform index) for every lane
__device__ __forceinline__ uint64_t read_lane64(uint64_t v, int src) {
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), src);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// ---- a window's own data, read once per workgroup ---------------------------------------------
//
// A window's sites are tiles [lo / 64, ceil(hi / 64)) of the planes: a C3 / C5 window (2 000 sites) is 32
// or 33 rows of 1 + n_sets (+ n_sets) used words, and between a few and ~700 stored target frequencies
// (the sites some set of the call selects).  Both fit a few KiB of LDS; a window they do not fit (a
// dense real-data window, a dense frequency array) is served from global memory by the same code.

constexpr int kRowWords = 1024;  // plane words of a window kept in LDS (8 KiB)
constexpr int kFreqCap = 1024;   // stored target frequencies of a window kept in LDS (8 KiB)
constexpr int kLdsTiles = 128;   // rows whose running count of stored frequencies is kept

struct WinLds {
  uint64_t rows[kRowWords];  // [tile - t0][used word]
  double freq[kFreqCap];     // the tiles' stored frequencies back to back
  uint32_t pre[kLdsTiles + 1];  // pre[j] = stored frequencies of the tiles before t0 + j
};

struct WinSrc {
  const uint64_t* g_rows;
  const double* g_freq;
  int64_t g_stride;
  const WinLds* lds;
  int t0;
  int used;  // words of a row that carry planes of this call: the LDS row pitch
  bool rows_in_lds, freq_in_lds;

  __device__ __forceinline__ uint64_t word(int t, int k) const {
    return rows_in_lds ? lds->rows[(t - t0) * used + k] : g_rows[static_cast<int64_t>(t) * g_stride + k];
  }
  // the stored frequency number `rank` of tile t (rank = popcount of the tile's "any" bits below the site)
  __device__ __forceinline__ double freq(int t, int rank) const {
    return freq_in_lds ? lds->freq[lds->pre[t - t0] + rank] : g_freq[static_cast<int64_t>(t) * kTile + rank];
  }
};

// the window's data where it lies in global memory
__device__ __forceinline__ WinSrc global_window(const WinArgs& a, int lo) {
  WinSrc s;
  s.g_rows = a.planes;
  s.g_freq = a.tgt_freq;
  s.g_stride = a.stride;
  s.lds = nullptr;
  s.t0 = lo >> 6;
  s.used = 1 + a.n_sets * (a.with_inv ? 2 : 1);
  s.rows_in_lds = false;
  s.freq_in_lds = false;
  return s;
}

// Called by the whole workgroup (it synchronises): rows, and the stored frequencies when asked for.
__device__ __forceinline__ WinSrc load_window(const WinArgs& a, int lo, int hi, WinLds& sh, int tid, bool want_freq) {
  WinSrc s = global_window(a, lo);
  s.lds = &sh;
  const int nt = hi > lo ? ((hi + kTile - 1) >> 6) - s.t0 : 0;
  if (nt <= 0 || nt > kLdsTiles || nt * s.used > kRowWords) return s;  // uniform over the workgroup
  s.rows_in_lds = true;
  const int lane = tid & 63, wv = tid >> 6;
  const uint64_t* src = a.planes + static_cast<int64_t>(s.t0) * a.stride;
  for (int i = tid; i < nt * s.used; i += kWinThreads) {  // a row's used words are contiguous: coalesced per row
    const int j = i / s.used, k = i - j * s.used;
    sh.rows[i] = src[static_cast<int64_t>(j) * a.stride + k];
  }
  __syncthreads();
  if (!want_freq) return s;
  if (wv == 0) {  // running count of the "any" bits: one wave, 64 rows per round
    uint32_t running = 0;
    for (int j0 = 0; j0 < nt; j0 += 64) {
      const int j = j0 + lane;
      const uint32_t cnt = j < nt ? __popcll(sh.rows[j * s.used + kAny]) : 0u;
      uint32_t inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (j < nt) sh.pre[j] = running + inc - cnt;
      running += __shfl(inc, 63, 64);
    }
    if (lane == 0) sh.pre[nt] = running;
  }
  __syncthreads();
  if (sh.pre[nt] > kFreqCap) return s;  // uniform
  s.freq_in_lds = true;
  for (int j = wv; j < nt; j += kWinWaves) {  // a tile's stored frequencies: one coalesced load of its first popcount slots
    const uint32_t b = sh.pre[j], cnt = sh.pre[j + 1] - b;
    if (static_cast<uint32_t>(lane) < cnt) sh.freq[b + lane] = a.tgt_freq[static_cast<int64_t>(s.t0 + j) * kTile + lane];
  }
  __syncthreads();
  return s;
}

constexpr int kTileBatch = 4;  // tiles whose per-site values are fetched together

// The per-site side of a chunk of 64 tiles.  Lane = tile while the plane words are fetched; the
// tiles that hold a marked site (bit of `nonempty`, wave-uniform) are then visited one by one with
// lane = SITE: on_tiles(n, tile_lane[], ...) gets up to kTileBatch of them at a time so that the
// per-site loads of a batch are issued before any is consumed.  A sparse window (C3: two condition
// sites in 2 000) visits one or two tiles, a dense one (C5's loose sets: 600 in 2 000) all 32.
template <typename F>
__device__ __forceinline__ void for_nonempty_tiles(unsigned long long nonempty, F&& on_batch) {
  while (nonempty) {  // wave-uniform
    int tl[kTileBatch];
    int n = 0;
#pragma unroll
    for (int u = 0; u < kTileBatch; ++u) {
      tl[u] = nonempty ? __ffsll(static_cast<long long>(nonempty)) - 1 : 0;
      if (nonempty) {
        ++n;
        nonempty &= nonempty - 1ull;
      }
    }
    on_batch(n, tl);
  }
}

// on_chunk(t, live, mask) is called by the whole wave for every 64 tiles of the window (lane = tile).
template <typename F>
__device__ __forceinline__ void walk_tiles(int lo, int hi, int lane, F&& on_chunk) {
  if (hi <= lo) return;
  const int t0 = lo >> 6, t1 = (hi + kTile - 1) >> 6;
  for (int tb = t0; tb < t1; tb += 64) {
    const int t = tb + lane;
    on_chunk(t, t < t1, t < t1 ? range_mask(t, lo, hi) : 0ull);
  }
}

// ---- sets with more than kWaveCap qualifying sites: the wave's radix select ----------------------
//
// Loose source conditions (C5's ("=0", "=0") and (">=0", ">=0") sets select ~630 of a window's 2 000
// sites) are finished by the same wave.  The k-th smallest value is found by a radix select ON THE
// VALUE: all values lie in [0, 1], so digit l of a value is floor(frac_l * 256) with frac_0 = v,
// frac_{l+1} = frac_l * 256 - digit_l -- multiplications by a power of two and subtractions of the
// integer part, all exact in binary floating point -- which makes the digits a monotone, lossless code
// of the value.  One histogram pass over the set's values (re-read through the window's planes: they
// sit in LDS) narrows the candidates to one of 257 bins (bin 256 holds exactly 1.0); frequencies
// k / (called * ploidy) near a high quantile separate within one or two levels, then the members of the
// bin are gathered (ballot-ranked, no atomics) and ranked directly.  Round 3 gave such a set to a
// 256-thread workgroup of its own (1 025 bins, ~25 workgroup barriers per set): under the next step's
// genotype stream a barrier couples four waves that each wait microseconds for their turn.

constexpr int kDigitBits = 8;
constexpr int kBins = (1 << kDigitBits) + 1;  // digits 0..255, and 256 for the value 1.0
// 15 x 8 bits: a frequency is >= 1 / (n_called * ploidy) > 2^-55 (n_ind <= 2^24, ploidy < 2^31), so
// its 53 mantissa bits end above 2^-108 -- 120 fractional bits tell any two distinct doubles apart,
// and members that still share a bin after the last level are the same number
constexpr int kMaxLevels = 15;
constexpr int kListCap = 1024;  // qualifying sites of a set whose LDS frequency slots a wave lists (16 bits each)
static_assert(kFreqCap <= 0x8000, "a list entry keeps the frequency's LDS slot in 15 bits");

struct WaveLds {
  double vals[kWaveCap];      // the set's first qualifying values (all of them for a light set); the select's gathered bin
  uint32_t hist[kBins + 63];  // padded so that 64 lanes x 5 bins stay inside
  uint32_t n_small;
};

// The digits chosen so far, as the number they spell: prefix = floor(v * 256^level) for every value v still
// on the path (a floor of a double: exact), scale = 256^level.
struct Path {
  double prefix;
  double scale;
};

// digit `level` of v, or -1 when v left the chosen path at an earlier level.  v * scale and its multiple by
// 256 are scalings by powers of two, the floors of doubles are doubles, and the difference is the digit
// itself (0..255, or 256 for v = 1.0 at level 0): every step is exact.
__device__ __forceinline__ int digit_on_path(double v, const Path& path) {
  const double s = v * path.scale;
  if (path.scale != 1.0 && floor(s) != path.prefix) return -1;  // at level 0 every value is on the path (1.0: digit 256)
  return static_cast<int>(floor(s * static_cast<double>(1 << kDigitBits)) - path.prefix * static_cast<double>(1 << kDigitBits));
}

// A set's planes as a wave walks them when it needs every qualifying site: eight lanes per tile, each
// owning eight sites (one byte of the tile's condition word), eight tiles per round -- a 2 000-site window
// is five rounds, against 33 with lane = site.  on_byte(t, c, rank0, any_bits, inv_bits) is called by the
// WHOLE wave once per round (c = 0 in lanes without a condition site): bit b of c has its frequency at
// rank rank0 + popcount(any_bits below b) of tile t.  Bytes ascend with the lane, so (lane, bit) order
// is site order.
template <typename F>
__device__ __forceinline__ void walk_bytes(const WinSrc& src, int ci, int ii, int lo, int hi, int lane, F&& on_byte) {
  const int sub = lane & 7;
  // lane = tile first: ONE load per plane brings 64 tiles' words (the window), the rounds below take
  // theirs from the lanes that hold them -- no round waits for a load of its own
  walk_tiles(lo, hi, lane, [&](int t, bool live, uint64_t mask) {
    const uint64_t cw = live ? src.word(t, ci) & mask : 0ull;
    const uint64_t aw = live ? src.word(t, kAny) : 0ull;
    const uint64_t iw = (live && ii >= 0) ? src.word(t, ii) : 0ull;
    const unsigned long long nonempty = __ballot(cw != 0ull);
    if (nonempty == 0ull) return;  // no condition site among these 4 096: the usual case for a tight set
    const int tb = t - lane;
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
      if (((nonempty >> (8 * r)) & 0xFFull) == 0ull) continue;  // wave-uniform
      const int from = 8 * r + (lane >> 3);
      const uint64_t c64 = __shfl(cw, from, 64), a64 = __shfl(aw, from, 64), i64 = __shfl(iw, from, 64);
      const uint32_t c = static_cast<uint32_t>(c64 >> (8 * sub)) & 0xFFu;
      on_byte(tb + from, c, __popcll(a64 & ((1ull << (8 * sub)) - 1ull)), static_cast<uint32_t>(a64 >> (8 * sub)) & 0xFFu,
              static_cast<uint32_t>(i64 >> (8 * sub)) & 0xFFu);
    }
  });
}

// How the passes of a heavy set see its values: the wave's list of LDS frequency slots when the build
// pass could make one (the window's frequencies are in LDS and the set selects at most kListCap sites:
// ten rounds of 64 for C5's 630), else the planes again.  use(e) runs per lane, possibly under a
// divergent mask: no wave-wide operation inside.
struct SetValues {
  const WinSrc* src;
  const uint16_t* list;  // slot of each qualifying site's frequency in WinLds::freq, bit 15 = inverted; or NULL
  int ci, ii, lo, hi;
  uint32_t n_c;
};

// the eight sites of a lane's byte: all frequencies first (independent reads), then fn(b, e) per condition bit
template <typename F>
__device__ __forceinline__ void for_byte_values(const WinSrc& src, int t, uint32_t c, int rank0, uint32_t ab, uint32_t iv, F&& fn) {
  double v[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) v[b] = ((c >> b) & 1u) ? src.freq(t, rank0 + __popc(ab & ((1u << b) - 1u))) : 0.0;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if ((c >> b) & 1u) fn(b, ((iv >> b) & 1u) ? 1.0 - v[b] : v[b]);
}

constexpr int kListUnroll = 4;

template <typename F>
__device__ __forceinline__ void for_each_selected(const SetValues& sv, int lane, F&& use) {
  if (sv.list) {
    for (uint32_t i0 = 0; i0 < sv.n_c; i0 += 64 * kListUnroll) {  // four entries per lane in flight
      uint32_t entry[kListUnroll];
      double v[kListUnroll];
#pragma unroll
      for (int u = 0; u < kListUnroll; ++u) {
        const uint32_t i = i0 + u * 64 + lane;
        entry[u] = i < sv.n_c ? sv.list[i] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int u = 0; u < kListUnroll; ++u) v[u] = entry[u] != 0xFFFFFFFFu ? sv.src->lds->freq[entry[u] & 0x7FFFu] : 0.0;
#pragma unroll
      for (int u = 0; u < kListUnroll; ++u)
        if (entry[u] != 0xFFFFFFFFu) use((entry[u] & 0x8000u) ? 1.0 - v[u] : v[u]);
    }
  } else {
    walk_bytes(*sv.src, sv.ci, sv.ii, sv.lo, sv.hi, lane, [&](int t, uint32_t c, int rank0, uint32_t ab, uint32_t iv) {
      for_byte_values(*sv.src, t, c, rank0, ab, iv, [&](int, double e) { use(e); });
    });
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// first radix digit of a value in [0, 1] (digit_on_path at level 0)
__device__ __forceinline__ int first_digit(double e) {
  const int d = static_cast<int>(e * static_cast<double>(1 << kDigitBits));
  return d > (1 << kDigitBits) ? (1 << kDigitBits) : d;
}

// The bin of the wave's histogram that holds rank k: {digit, rank inside the bin, members}.  Lane l owns
// bins 5 l .. 5 l + 4; exclusive scan of the lanes' sums.
struct BinOfRank {
  int digit;
  uint32_t rank_in_bin, members;
};
__device__ __forceinline__ BinOfRank bin_of_rank(const WaveLds& sh, uint32_t k, int lane) {
  uint32_t h[5], mine_sum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    h[j] = sh.hist[lane * 5 + j];
    mine_sum += h[j];
  }
  uint32_t inc = mine_sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  uint32_t before = inc - mine_sum;
  int my_digit = -1;
  uint32_t my_rem = 0, my_count = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (h[j] != 0 && k >= before && k < before + h[j]) {  // exactly one bin of one lane
      my_digit = lane * 5 + j;
      my_rem = k - before;
      my_count = h[j];
    }
    before += h[j];
  }
  const int owner = __ffsll(static_cast<long long>(__ballot(my_digit >= 0))) - 1;
  return BinOfRank{__shfl(my_digit, owner, 64), static_cast<uint32_t>(__shfl(my_rem, owner, 64)),
                   static_cast<uint32_t>(__shfl(my_count, owner, 64))};
}

// the values of ranks k0 and k1 among vals[0, n) (ties broken by slot: ranks are a permutation); a rank
// >= n leaves its output untouched
__device__ __forceinline__ void two_ranks(const double* vals, uint32_t n, uint32_t k0, uint32_t k1, int lane, double& x0, double& x1) {
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e = e0 + lane;
    const bool act = e < n;
    const double ve = act ? vals[e] : 0.0;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; ++j) {
      const double vj = vals[j];  // same address in every lane: LDS broadcast
      rank += (vj < ve) || (vj == ve && j < e);
    }
    const unsigned long long h0 = __ballot(act && rank == k0);
    const unsigned long long h1 = __ballot(act && rank == k1);
    if (h0) x0 = __shfl(ve, __ffsll(static_cast<long long>(h0)) - 1, 64);
    if (h1) x1 = __shfl(ve, __ffsll(static_cast<long long>(h1)) - 1, 64);
  }
}

__device__ __forceinline__ double wave_min(double m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(m, o, 64);
    m = other < m ? other : m;
  }
  return m;
}

// k-th smallest (0-based) of the set's qualifying values, by one wave
__device__ __forceinline__ double wave_select_kth(WaveLds& sh, const SetValues& sv, uint32_t k, int lane) {
  Path path{0.0, 1.0};
  for (int level = 0; level < kMaxLevels; ++level) {
#pragma unroll
    for (int j = 0; j < 5; ++j) sh.hist[j * 64 + lane] = 0u;  // 320 >= kBins
    if (lane == 0) sh.n_small = 0u;
    wave_lds_fence();
    for_each_selected(sv, lane, [&](double e) {
      const int d = digit_on_path(e, path);
      if (d >= 0) atomicAdd(&sh.hist[d], 1u);
    });
    wave_lds_fence();
    const BinOfRank bin = bin_of_rank(sh, k, lane);
    const int digit = bin.digit;
    k = bin.rank_in_bin;
    const uint32_t members = bin.members;
    if (members <= kWaveCap || level == kMaxLevels - 1) {
      // gather the bin (at the last level all its members are the same number) and rank directly; the order
      // of the members does not matter: ties are broken by slot, whichever value ends up with rank k is the
      // k-th smallest
      for_each_selected(sv, lane, [&](double e) {
        if (digit_on_path(e, path) == digit) {
          const uint32_t slot = atomicAdd(&sh.n_small, 1u);
          if (slot < kWaveCap) sh.vals[slot] = e;
        }
      });
      wave_lds_fence();
      if (members > kWaveCap) return sh.vals[0];  // last level: all equal
      double found = 0.0, unused = 0.0;
      two_ranks(sh.vals, members, k, k, lane, found, unused);
      wave_lds_fence();
      return found;
    }
    path.prefix = path.prefix * static_cast<double>(1 << kDigitBits) + static_cast<double>(digit);
    path.scale *= static_cast<double>(1 << kDigitBits);
  }
  return 0.0;  // not reached
}

// One set of one window answered by one wave.  Build pass: counts, the qualifying effective frequencies
// into the wave's LDS slice (site order; U's tgt > x is taken from the same values), the histogram of
// their first radix digit and -- when the window's frequencies are in LDS -- the list of their slots.
// Then Q = numpy's `linear` quantile: by rank counting over the slice, or, above kWaveCap qualifying
// sites, from the histogram: ONE more pass gathers the bin that holds the wanted rank (and the smallest
// value above it), and the order statistics, Q and the size of the Q list follow from the bin's members
// and the histogram's counts.  Only a bin with more than kWaveCap members goes through the level-by-level
// select and its extra passes.
__device__ __forceinline__ void wave_set(const WinArgs& a, const WinSrc& src, WaveLds& sh, uint16_t* list, int set, int lo,
                                         int hi, int lane, int64_t ridx) {
  const int ci = cond_word(set), ii = inv_word(a.with_inv, a.n_sets, set);
  const double x = a.x[set];
  double* vals = sh.vals;
  const bool can_list = list != nullptr && src.freq_in_lds;
#pragma unroll
  for (int j = 0; j < 5; ++j) sh.hist[j * 64 + lane] = 0u;  // 320 >= kBins
  if (lane == 0) sh.n_small = 0u;
  wave_lds_fence();
  uint32_t n_c = 0, u_mine = 0;
  walk_bytes(src, ci, ii, lo, hi, lane, [&](int t, uint32_t c, int rank0, uint32_t ab, uint32_t iv) {
    const uint32_t cnt = __popc(c);
    uint32_t inc = cnt;  // the lanes' bytes are in site order: an exclusive scan places every site
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    const uint32_t base = n_c + inc - cnt;
    n_c += __shfl(inc, 63, 64);
    const uint32_t slot0 = (can_list && c) ? src.lds->pre[t - src.t0] + rank0 : 0u;
    for_byte_values(src, t, c, rank0, ab, iv, [&](int b, double e) {
      const uint32_t below = (1u << b) - 1u;
      const uint32_t slot = base + __popc(c & below);
      if (slot < kWaveCap) vals[slot] = e;
      if (can_list && slot < kListCap)
        list[slot] = static_cast<uint16_t>((slot0 + __popc(ab & below)) | (((iv >> b) & 1u) ? 0x8000u : 0u));
      atomicAdd(&sh.hist[first_digit(e)], 1u);
      u_mine += e > x ? 1u : 0u;
    });
  });
  const uint32_t n_u = wave_sum(u_mine);
  double q = std::numeric_limits<double>::quiet_NaN();
  uint32_t n_q = 0;
  if (n_c > 0) {
    wave_lds_fence();
    const double v = static_cast<double>(n_c - 1) * a.quantile[set];
    const bool take_max = v >= static_cast<double>(n_c - 1);  // at/after the last index: maximum
    const double fl_v = floor(v);
    const uint32_t k0 = take_max ? n_c - 1 : static_cast<uint32_t>(fl_v);
    if (n_c <= kWaveCap) {
      double x0 = 0.0, x1 = 0.0;
      two_ranks(vals, n_c, k0, take_max ? k0 : k0 + 1, lane, x0, x1);
      q = take_max ? x0 : numpy_lerp(x0, x1, v, fl_v);
      for (uint32_t e0 = 0; e0 < n_c; e0 += 64) {
        const uint32_t e = e0 + lane;
        n_q += __popcll(__ballot(e < n_c && vals[e] >= q));
      }
    } else {
      const SetValues sv{&src, (can_list && n_c <= kListCap) ? list : nullptr, ci, ii, lo, hi, n_c};
      const BinOfRank bin = bin_of_rank(sh, k0, lane);
      if (bin.members <= kWaveCap) {
        // gather the bin's members (any order) and find the smallest value of the bins above it
        double next_mine = std::numeric_limits<double>::infinity();
        for_each_selected(sv, lane, [&](double e) {
          const int d = first_digit(e);
          if (d == bin.digit) {
            vals[atomicAdd(&sh.n_small, 1u)] = e;  // exactly bin.members of them
          } else if (d > bin.digit && e < next_mine) {
            next_mine = e;
          }
        });
        const double next_bin = wave_min(next_mine);
        wave_lds_fence();
        double x0 = 0.0, x1 = next_bin;  // the next order statistic is the next bin's smallest when the bin ends at k0
        two_ranks(vals, bin.members, bin.rank_in_bin, take_max ? bin.rank_in_bin : bin.rank_in_bin + 1, lane, x0, x1);
        q = take_max ? x0 : numpy_lerp(x0, x1, v, fl_v);
        // every value of a higher bin is >= x1 >= q; of the bin's own members those that reach q
        n_q = n_c - (k0 - bin.rank_in_bin) - bin.members;
        for (uint32_t e0 = 0; e0 < bin.members; e0 += 64) {
          const uint32_t e = e0 + lane;
          n_q += __popcll(__ballot(e < bin.members && vals[e] >= q));
        }
      } else {
        const double x0 = wave_select_kth(sh, sv, k0, lane);
        q = x0;
        if (!take_max) {
          // the next order statistic: x0 again when more than k0 + 1 values are <= x0, else the smallest
          // value above x0 -- one pass instead of a second selection
          uint32_t le_mine = 0;
          double m_gt = std::numeric_limits<double>::infinity();
          for_each_selected(sv, lane, [&](double e) {
            le_mine += e <= x0 ? 1u : 0u;
            if (e > x0 && e < m_gt) m_gt = e;
          });
          const uint32_t n_le = wave_sum(le_mine);
          m_gt = wave_min(m_gt);
          q = numpy_lerp(x0, n_le > k0 + 1 ? x0 : m_gt, v, fl_v);
        }
        uint32_t q_mine = 0;
        for_each_selected(sv, lane, [&](double e) { q_mine += e >= q ? 1u : 0u; });
        n_q = wave_sum(q_mine);
      }
    }
    wave_lds_fence();  // the slice is reused by this wave's next set
  }
  if (lane == 0) {
    sai_window_record rec;
    rec.n_sites = hi > lo ? hi - lo : 0;
    rec.u_count = static_cast<int32_t>(n_u);
    rec.n_cond = static_cast<int32_t>(n_c);
    rec.n_cdd_q = static_cast<int32_t>(n_q);
    rec.q = q;
    a.records[ridx] = rec;
  }
}

// SHARED: one workgroup per window; the window's data comes into LDS once and the four waves share the
// sets (calls with >= kWinWaves sets: C5's sweep).  !SHARED: one WAVE per window, every word read where it
// lies (one to three sets read a plane word once to three times; C2 / C3 / C4), and no workgroup barrier.
template <bool SHARED>
struct StatsLds {
  WinLds win;
  WaveLds wave[kWinWaves];
  uint16_t list[kWinWaves][kListCap];
};
template <>
struct StatsLds<false> {
  WaveLds wave[kWinWaves];
};

template <bool SHARED>
__global__ __launch_bounds__(kWinThreads) void window_stats_kernel(WinArgs a) {
  __shared__ StatsLds<SHARED> sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // the compiler cannot see that it is uniform over the wave
  // each XCD works on a contiguous run of (overlapping) windows
  const int w = SHARED ? xcd_contiguous(blockIdx.x, gridDim.x) : xcd_contiguous(blockIdx.x, gridDim.x) * kWinWaves + wv;
  if (w >= a.n_windows) return;  // SHARED: the whole workgroup; else the wave (which meets no barrier below)
  const int lo = clamp_site(a.lo[w], a.n_sites), hi = clamp_site(a.hi[w], a.n_sites);
  WinSrc src;
  uint16_t* list = nullptr;
  if constexpr (SHARED) {
    src = load_window(a, lo, hi, sh.win, tid, true);
    list = sh.list[wv];
  } else {
    src = global_window(a, lo);
  }
  for (int set = SHARED ? wv : 0; set < a.n_sets; set += SHARED ? kWinWaves : 1)
    wave_set(a, src, sh.wave[wv], list, set, lo, hi, lane, static_cast<int64_t>(set) * a.n_windows + w);
}

// ---- CSR offsets ---------------------------------------------------------------------------

// Exclusive prefix sums of u_count and n_cdd_q over the records in (set, window) order:
// cdd_off[2r] / cdd_off[2r+1] = start of record r's U / Q list (or -1 when the list would not fit
// its buffer); cdd_total[0..1] = entries needed in all.  Two small launches: every workgroup adds up
// its 1024 records (window_scan_partials), then every workgroup sums the partials before it -- at
// most a few hundred numbers -- and scans its own records (window_scan_apply).  The first version
// was ONE workgroup walking all records: 20 us for C3's 9 999 records but 0.2 ms for a 16-set chunk
// of C5 and 1.4 ms for C4's 110 017 windows on one GPU, the longest kernel of the windows stage.
// Workgroups are 256 threads with few registers on purpose: the pipelined scorer runs this stage
// under the next step's site pass, whose 4 waves per SIMD leave one wave slot per SIMD free.
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanBlock = kScanThreads * kScanPerThread;  // records per workgroup

__device__ __forceinline__ long long block_sum_ll(long long v, long long* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kScanThreads) void window_scan_partials_kernel(WinArgs a, long long* partials) {
  __shared__ long long red[2][kScanThreads / 64];
  const int tid = threadIdx.x;
  const int64_t n = static_cast<int64_t>(a.n_sets) * a.n_windows;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanBlock;
  long long su = 0, sq = 0;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {  // lane-consecutive records: coalesced 24-byte loads
    const int64_t r = base + k * kScanThreads + tid;
    if (r < n) {
      su += a.records[r].u_count;
      sq += a.records[r].n_cdd_q;
    }
  }
  const long long tu = block_sum_ll(su, red[0], tid);
  const long long tq = block_sum_ll(sq, red[1], tid);
  if (tid == 0) {
    partials[2 * blockIdx.x] = tu;
    partials[2 * blockIdx.x + 1] = tq;
  }
}

__global__ __launch_bounds__(kScanThreads) void window_scan_apply_kernel(WinArgs a, const long long* partials) {
  __shared__ long long red[2][kScanThreads / 64];
  __shared__ long long wave_tot[2][kScanThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = static_cast<int64_t>(a.n_sets) * a.n_windows;
  // everything before this workgroup
  long long bu = 0, bq = 0;
  for (int j = tid; j < static_cast<int>(blockIdx.x); j += kScanThreads) {
    bu += partials[2 * j];
    bq += partials[2 * j + 1];
  }
  const long long base_u = block_sum_ll(bu, red[0], tid);
  const long long base_q = block_sum_ll(bq, red[1], tid);
  // thread t owns records first .. first+3 (consecutive), so that one shuffle scan orders the block
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kScanBlock + static_cast<int64_t>(tid) * kScanPerThread;
  int32_t nu[kScanPerThread], nq[kScanPerThread];
  long long su = 0, sq = 0;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {
    const int64_t r = first + k;
    nu[k] = r < n ? a.records[r].u_count : 0;
    nq[k] = r < n ? a.records[r].n_cdd_q : 0;
    su += nu[k];
    sq += nq[k];
  }
  long long iu = su, iq = sq;  // inclusive scan over the threads of a wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long tu = __shfl_up(iu, o, 64), tq = __shfl_up(iq, o, 64);
    if (lane >= o) { iu += tu; iq += tq; }
  }
  if (lane == 63) { wave_tot[0][wave] = iu; wave_tot[1][wave] = iq; }
  __syncthreads();
  long long ou = base_u + iu - su, oq = base_q + iq - sq;
  long long block_u = 0, block_q = 0;
#pragma unroll
  for (int v = 0; v < kScanThreads / 64; ++v) {
    if (v < wave) { ou += wave_tot[0][v]; oq += wave_tot[1][v]; }
    block_u += wave_tot[0][v];
    block_q += wave_tot[1][v];
  }
  longlong2* out = reinterpret_cast<longlong2*>(a.cdd_off);
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {
    const int64_t r = first + k;
    if (r < n) out[r] = make_longlong2((ou + nu[k] <= a.cap_u) ? ou : -1, (oq + nq[k] <= a.cap_q) ? oq : -1);
    ou += nu[k];
    oq += nq[k];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    a.cdd_total[0] = base_u + block_u;
    a.cdd_total[1] = base_q + block_q;
  }
}

// ---- candidate lists -----------------------------------------------------------------------

template <bool SHARED>
struct ListLds {
  WinLds win;
  int32_t any_list;
};
template <>
struct ListLds<false> {
  int32_t any_list;
};

// The candidate lists of the window's sets, in ascending site order: a tile's U sites are the condition
// bits whose effective frequency exceeds x, its Q sites those whose frequency reaches q (a ballot each);
// each leaves at `offset + done + popcount(word & lanes below)`.  SHARED / !SHARED as in window_stats.
template <bool SHARED>
__global__ __launch_bounds__(kWinThreads) void window_lists_kernel(WinArgs a) {
  __shared__ ListLds<SHARED> sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // the compiler cannot see that it is uniform over the wave
  const int w = SHARED ? xcd_contiguous(blockIdx.x, gridDim.x) : xcd_contiguous(blockIdx.x, gridDim.x) * kWinWaves + wv;
  if constexpr (SHARED) {
    if (w >= a.n_windows) return;
    // nothing to write for any set of this window: leave before touching the planes
    if (tid == 0) sh.any_list = 0;
    __syncthreads();
    if (tid < a.n_sets) {
      const int64_t r = static_cast<int64_t>(tid) * a.n_windows + w;
      const sai_window_record rec = a.records[r];
      if ((rec.u_count > 0 && a.cdd_off[2 * r] >= 0 && a.cdd_u != nullptr) || (rec.n_cdd_q > 0 && a.cdd_off[2 * r + 1] >= 0 && a.cdd_q != nullptr))
        sh.any_list = 1;
    }
    __syncthreads();
    if (!sh.any_list) return;  // uniform
  } else {
    if (w >= a.n_windows) return;
  }
  int lo = 0, hi = 0;
  WinSrc src;
  if constexpr (SHARED) {
    lo = clamp_site(a.lo[w], a.n_sites);
    hi = clamp_site(a.hi[w], a.n_sites);
    src = load_window(a, lo, hi, sh.win, tid, true);
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int set = SHARED ? wv : 0; set < a.n_sets; set += SHARED ? kWinWaves : 1) {
    const int64_t ridx = static_cast<int64_t>(set) * a.n_windows + w;
    const sai_window_record rec = a.records[ridx];
    const long long off_u = a.cdd_off[2 * ridx + 0], off_q = a.cdd_off[2 * ridx + 1];
    const bool write_u = rec.u_count > 0 && off_u >= 0 && a.cdd_u != nullptr;
    const bool write_q = rec.n_cdd_q > 0 && off_q >= 0 && a.cdd_q != nullptr;
    if (!write_u && !write_q) continue;  // C3: most windows have no U candidate, a window without condition sites no Q list
    if constexpr (!SHARED) {
      if (hi == 0 && lo == 0) {
        lo = clamp_site(a.lo[w], a.n_sites);
        hi = clamp_site(a.hi[w], a.n_sites);
        src = global_window(a, lo);
      }
    }
    const int ci = cond_word(set), ii = inv_word(a.with_inv, a.n_sets, set);
    const double q = rec.q, x = a.x[set];
    uint32_t done_u = 0, done_q = 0;
    walk_tiles(lo, hi, lane, [&](int t, bool live, uint64_t mask) {
      const uint64_t c = live ? src.word(t, ci) & mask : 0ull;
      const unsigned long long nonempty = __ballot(c != 0ull);
      if (nonempty == 0ull) return;
      const uint64_t iv = (c && ii >= 0) ? src.word(t, ii) : 0ull;
      const uint64_t an = c ? src.word(t, kAny) : 0ull;
      const int tb = t - lane;
      // lane = site: a tile's marked sites leave in site order, placed by the popcount of the lanes below
      for_nonempty_tiles(nonempty, [&](int n, const int (&tl)[kTileBatch]) {
        uint64_t cw[kTileBatch], iw[kTileBatch], aw[kTileBatch];
        bool mu[kTileBatch], mq[kTileBatch];
#pragma unroll
        for (int u = 0; u < kTileBatch; ++u) {
          cw[u] = u < n ? read_lane64(c, tl[u]) : 0ull;
          iw[u] = u < n ? read_lane64(iv, tl[u]) : 0ull;
          aw[u] = u < n ? read_lane64(an, tl[u]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kTileBatch; ++u) {
          const bool mine_c = (cw[u] >> lane) & 1ull;
          const double v = mine_c ? src.freq(tb + tl[u], __popcll(aw[u] & lt_mask)) : 0.0;
          const double e = ((iw[u] >> lane) & 1ull) ? 1.0 - v : v;
          mu[u] = write_u && mine_c && e > x;
          mq[u] = write_q && mine_c && e >= q;
        }
#pragma unroll
        for (int u = 0; u < kTileBatch; ++u) {
          const unsigned long long uw = __ballot(mu[u]), qw = __ballot(mq[u]);
          if (uw | qw) {  // uniform: positions are read only for the sites that leave
            const int64_t site = static_cast<int64_t>(tb + tl[u]) * kTile + lane;
            const int32_t p = (mu[u] || mq[u]) ? (a.pos ? a.pos[site] : static_cast<int32_t>(site)) : 0;
            if (mu[u]) a.cdd_u[off_u + done_u + __popcll(uw & lt_mask)] = p;
            if (mq[u]) a.cdd_q[off_q + done_q + __popcll(qw & lt_mask)] = p;
            done_u += __popcll(uw);
            done_q += __popcll(qw);
          }
        }
      });
    });
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int64_t sai_window_total_words(int32_t n_sets, int32_t n_windows) {
  if (n_sets < 0 || n_windows < 0) return -1;
  const int64_t n_rec = static_cast<int64_t>(n_sets) * n_windows;
  return 2 + 2 * ((n_rec + kScanBlock - 1) / kScanBlock);
}

static int launch_window_bounds(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                                const int64_t* win_start, const int64_t* win_end, const int32_t* seg_lo,
                                const int32_t* seg_hi, int32_t* lo, int32_t* hi, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_sites < 0 || n_sites >= 0x7FFFFFFFll || n_windows < 0) return fail(SAI_ERR_ARG, "size out of range");
  if (n_windows == 0) return SAI_OK;
  if ((n_sites > 0 && !pos) || !win_start || !win_end || !lo || !hi) return fail(SAI_ERR_ARG, "NULL buffer");
  static const int wide_max = [] {  // SAI_BOUNDS_WIDE_MAX: tuning knob for sweeps
    const char* e = std::getenv("SAI_BOUNDS_WIDE_MAX");
    return e ? std::atoi(e) : kBoundsWideMax;
  }();
  if (n_windows <= wide_max) {
    const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(n_windows) * 2 * kBoundsWideGroup + 255) / 256);
    hipLaunchKernelGGL((window_bounds_kernel<kBoundsWideGroup, true>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       pos, n_sites, n_windows, win_start, win_end, seg_lo, seg_hi, lo, hi);
  } else {
    const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(n_windows) * kBoundsGroup + 255) / 256);
    hipLaunchKernelGGL((window_bounds_kernel<kBoundsGroup, false>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       pos, n_sites, n_windows, win_start, win_end, seg_lo, seg_hi, lo, hi);
  }
  return check_launch("window_bounds");
}

int sai_window_bounds(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                      const int64_t* win_start, const int64_t* win_end, int32_t* lo, int32_t* hi, void* stream) {
  return launch_window_bounds(ctx, pos, n_sites, n_windows, win_start, win_end, nullptr, nullptr, lo, hi, stream);
}

int sai_window_bounds_seg(sai_ctx* ctx, const int32_t* pos, int64_t n_sites, int32_t n_windows,
                          const int64_t* win_start, const int64_t* win_end, const int32_t* seg_lo,
                          const int32_t* seg_hi, int32_t* lo, int32_t* hi, void* stream) {
  if (n_windows > 0 && (!seg_lo || !seg_hi)) return fail(SAI_ERR_ARG, "NULL segment bounds");
  return launch_window_bounds(ctx, pos, n_sites, n_windows, win_start, win_end, seg_lo, seg_hi, lo, hi, stream);
}

int sai_window_stats(sai_ctx* ctx, int64_t n_sites, const double* tgt_freq, const uint64_t* planes,
                     int64_t plane_stride, int32_t n_sets, const sai_params* sets_host, int32_t n_windows, const int32_t* lo, const int32_t* hi,
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
  if ((n_sites > 0 && (!tgt_freq || !planes)) || !lo || !hi || !records || !cdd_off)
    return fail(SAI_ERR_ARG, "NULL buffer");
  if (int rc = check_plane_stride(plane_stride, n_sets)) return rc;
  WinArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_sites = n_sites;
  a.tgt_freq = tgt_freq;
  a.planes = planes;
  a.stride = plane_stride;
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
  a.with_inv = sets_with_inverted(n_sets, sets_host);
  for (int s = 0; s < n_sets; ++s) {
    a.quantile[s] = sets_host[s].quantile;
    a.x[s] = sets_host[s].x;
  }
  // a call with at least kWinWaves sets shares a window's data between the waves of a workgroup; fewer sets:
  // one wave per window, nothing staged
  const bool shared = n_sets >= kWinWaves;
  const dim3 win_grid(shared ? static_cast<unsigned>(n_windows) : static_cast<unsigned>((n_windows + kWinWaves - 1) / kWinWaves));
  if (shared) hipLaunchKernelGGL(window_stats_kernel<true>, win_grid, dim3(kWinThreads), 0, st, a);
  else hipLaunchKernelGGL(window_stats_kernel<false>, win_grid, dim3(kWinThreads), 0, st, a);
  if (int rc = check_launch("window_stats")) return rc;
  const int64_t n_rec = static_cast<int64_t>(n_sets) * n_windows;
  const unsigned scan_grid = static_cast<unsigned>((n_rec + kScanBlock - 1) / kScanBlock);
  long long* partials = reinterpret_cast<long long*>(cdd_total) + 2;  // the caller's scratch behind the two totals
  hipLaunchKernelGGL(window_scan_partials_kernel, dim3(scan_grid), dim3(kScanThreads), 0, st, a, partials);
  if (int rc = check_launch("window_scan_partials")) return rc;
  hipLaunchKernelGGL(window_scan_apply_kernel, dim3(scan_grid), dim3(kScanThreads), 0, st, a, partials);
  if (int rc = check_launch("window_scan_apply")) return rc;
  if (shared) hipLaunchKernelGGL(window_lists_kernel<true>, win_grid, dim3(kWinThreads), 0, st, a);
  else hipLaunchKernelGGL(window_lists_kernel<false>, win_grid, dim3(kWinThreads), 0, st, a);
  return check_launch("window_lists");
}

}  // extern "C"
