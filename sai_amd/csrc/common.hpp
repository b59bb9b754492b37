// Shared by every translation unit of libsaihip: error reporting, the context, launch helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>

#include "saihip.h"

// core.hip; also called by vcf_ingest.cpp.  Stores the thread's message, returns `code`.
extern "C" int sai_set_error(int code, const char* fmt, ...);
#define fail sai_set_error

#define SAI_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(SAI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                    \
  } while (0)

constexpr int kTile = SAI_TILE_SITES;
constexpr int kMaxPops = 2 + SAI_FUSED_SRC;  // populations of a streaming pass
constexpr int kBigPops = 2 + SAI_MAX_SRC;    // populations of the stand-alone per-site decision (sai_site_flags)
constexpr int kProbeWavesPerCu = 32;
// Grid of the streaming passes: 16 single-wave workgroups per CU = 4 waves per SIMD, enough to
// saturate HBM (measured flat from 8 to 611 per CU) while leaving registers and LDS on every SIMD
// for the small kernels the pipelined scorer runs on a second stream under them.
constexpr int kStreamWavesPerCu = 16;

struct sai_ctx {
  int device;
  int n_cu;
  uint32_t* probe_partials;  // n_cu * kProbeWavesPerCu words (stream-read probe)
  // scratch of sai_single_window, grown on demand, freed by sai_ctx_destroy
  void* sw_dev;
  size_t sw_dev_cap;
  void* sw_host;  // pinned
  size_t sw_host_cap;
  // events the NEXT site-pass launch of this ctx carries in its dispatch packet (sai_plan_set_pass_events);
  // taken -- and cleared -- by that launch
  hipEvent_t next_start;
  hipEvent_t next_stop;
};

// launch a streaming site-pass kernel, with the ctx's pending events in the dispatch packet when there are any
template <typename K, typename... Args>
inline void launch_pass(sai_ctx* ctx, K kernel, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  hipEvent_t start = ctx->next_start, stop = ctx->next_stop;
  ctx->next_start = ctx->next_stop = nullptr;
  if (start || stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, start, stop, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

inline int enter(sai_ctx* ctx) {
  if (!ctx) return fail(SAI_ERR_ARG, "ctx is NULL");
  SAI_HIP(hipSetDevice(ctx->device));
  return SAI_OK;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SAI_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return SAI_OK;
}

// site_pass.hip: argument checks of a parameter-set array (n_src < 0: any number of sources)
int check_sets(int32_t n_sets, const sai_params* sets, int32_t n_src, int32_t max_sets = SAI_MAX_SETS);
// site_pass.hip: a flag-plane row of plane_stride words holds n_sets sets
int check_plane_stride(int64_t plane_stride, int32_t n_sets);
// the rows of these sets carry inverted words (some set lacks ancestral alleles): decided the same way
// by the call that writes the rows and the call that reads them
inline bool sets_with_inverted(int32_t n_sets, const sai_params* sets) {
  for (int32_t s = 0; s < n_sets; ++s)
    if (sets[s].anc_allele_available == 0) return true;
  return false;
}

inline int stream_waves_override() {
  static const int waves_per_cu = [] {  // SAI_STREAM_WAVES_PER_CU: tuning knob for sweeps
    const char* e = std::getenv("SAI_STREAM_WAVES_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 0;
  }();
  return waves_per_cu;
}

inline unsigned stream_grid(const sai_ctx* ctx, int64_t n_tiles, int waves_per_cu = kStreamWavesPerCu) {
  if (stream_waves_override()) waves_per_cu = stream_waves_override();
  const int64_t max_grid = static_cast<int64_t>(ctx->n_cu) * waves_per_cu;  // grid-stride beyond this
  return static_cast<unsigned>(n_tiles < max_grid ? n_tiles : max_grid);
}

// Waves per CU of the int8 / packed2 site pass (same-box sweeps: profiles/history/r04_waves_per_cu.txt,
// profiles/history/r04_shape_sweep.txt, profiles/r05_c5_grid.txt).  The pass is HBM-bound from 8 waves per CU on when its
// populations are wide, so the grid is chosen for what runs NEXT to it (the windows stage of the step before, on
// a second stream) and for the pass's own tail:
//  * with many parameter sets (C5's 18) every tile ends in the sets' evaluation, during which a wave loads
//    nothing, and the stage next to the pass is a large one: 12 waves per CU -- three per SIMD, which leaves the
//    stage's waves half of the register file.  C5 pipelined, one box: 3.09 ms per step at 12, 3.41-3.46 at 16
//    (round 4's 64-register form at 16 with the set-by-set decision: 3.15);
//  * with one to three sets, WIDE populations (>= 2 000 individuals per site in at most three populations:
//    C3, C4) and a long pass, 8 -- two waves per SIMD, multiples of four only: 9 and 10 are slower than
//    either -- is a little faster: C3 2.924 against 2.966 ms, 2000 / 2000 / 2 individuals 2.99 against 3.06,
//    three chromosomes of C4 4.37 against 4.47-4.57, packed2 0.79-0.82 against 0.85.  Narrower populations
//    need their waves (a population's rows are over before a wave has many loads in flight, and every
//    population boundary drains them): 250 / 250 / 2 is 7 % slower at 8, 100 / 100 / 1 / 1 26 %; a fourth
//    population (two sources) makes 8 and 16 equal at best;
//  * a short pass (C2: 15 625 tiles, under four per wave at 16) needs its waves for the ramp and the
//    tail: 16 (0.088 ms per step against 0.109 at 8).
constexpr int kManySets = 4;  // from this many parameter sets on the pass takes 12 waves per CU
inline int many_sets() {
  static const int n = [] {  // SAI_MANY_SETS: tuning knob for sweeps (99 = never)
    const char* e = std::getenv("SAI_MANY_SETS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : kManySets;
  }();
  return n;
}
inline int site_pass_waves_per_cu(const sai_ctx* ctx, int64_t n_tiles, int32_t n_sets, int32_t n_pops, int64_t individuals) {
  if (n_sets >= many_sets()) return 12;
  const bool wide = individuals >= 2000 && n_pops <= 3;
  return wide && n_tiles >= static_cast<int64_t>(ctx->n_cu) * 8 * 32 ? 8 : kStreamWavesPerCu;
}

// The site pass with DD's terms riding along (site_pass_dd.hip) is built for three waves per SIMD (one or two
// source individuals) or two (three or four), and takes them: unlike the plain pass it has arithmetic to hide
// behind its loads (C3 shape, two source individuals: 3.26 ms at 12 waves per CU, 3.45 at 8; three source
// individuals at a grid their registers do not hold at once: 4.02 against 3.67 -- profiles/r05_dd_pass.txt).
inline int dd_pass_waves_per_cu(const sai_ctx*, int64_t, int64_t, int n_rows) { return n_rows > 2 ? 8 : 12; }

// XCD-aware block order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a
// contiguous run of (overlapping) windows so their shared sites stay in one L2.
__device__ __forceinline__ int xcd_contiguous(int b, int n_blocks) {
  const int per = n_blocks >> 3;
  return (per > 0 && b < per * 8) ? (b & 7) * per + (b >> 3) : b;
}
