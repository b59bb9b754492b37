// Stand-alone microbenchmark: which streaming-read form gets closest to the HBM ceiling on this box?
// hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: grid-stride, UNROLL independent loads spread over the grid
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_gridstride(const u32x4* __restrict__ src, int64_t n, uint32_t* out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x12345678u) out[0] = r;
}

// B: each wave streams its own contiguous chunk (like site_counts: 512 KB per wave), UNROLL KiB in flight
template <int UNROLL, bool NT>
__global__ __launch_bounds__(64) void k_wavechunk(const u32x4* __restrict__ src, int64_t n, int64_t chunk_vecs, uint32_t* out) {
  const int lane = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const int64_t n_chunks = n / chunk_vecs;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const u32x4* base = src + c * chunk_vecs + lane;
    for (int64_t it = 0; it + UNROLL * 64 <= chunk_vecs; it += UNROLL * 64) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(base + it + u * 64) : base[it + u * 64];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
  }
  uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x12345678u) out[0] = r;
}

// C: LDS-DMA (global_load_lds_dwordx4) into a per-wave ring, then ds_read back
template <int SLOTS, int AUX>
__global__ __launch_bounds__(64) void k_glds(const u32x4* __restrict__ src, int64_t n, int64_t chunk_vecs, uint32_t* out) {
  __shared__ u32x4 ring[SLOTS][64];
  const int lane = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const int64_t n_chunks = n / chunk_vecs;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const u32x4* base = src + c * chunk_vecs + lane;
    const int64_t n_it = chunk_vecs / 64;
    for (int64_t it0 = 0; it0 < n_it; it0 += SLOTS) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (it0 + s) * 64),
                                         (__attribute__((address_space(3))) void*)&ring[s][0], 16, 0, AUX);
      __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) etc.
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) acc ^= ring[s][lane];
    }
  }
  uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x12345678u) out[0] = r;
}

// D: wavechunk + the SWAR byte accumulation of site_counts (10 VALU per word)
__device__ __forceinline__ void acc_word(uint32_t w, uint32_t& lo, uint32_t& hi, uint32_t& ms) {
  const uint32_t neg = w & 0x80808080u, m1 = neg >> 7;
  ms += m1;
  const uint32_t mask = (neg - m1) | neg, val = w & ~mask;
  lo += val & 0x00FF00FFu;
  hi += (val >> 8) & 0x00FF00FFu;
}
template <int UNROLL, int MINW>
__global__ __launch_bounds__(64, MINW) void k_swar(const u32x4* __restrict__ src, int64_t n, int64_t chunk_vecs, uint32_t* out) {
  const int lane = threadIdx.x;
  uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
  const int64_t n_chunks = n / chunk_vecs;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const u32x4* base = src + c * chunk_vecs + lane;
    for (int64_t it = 0; it + UNROLL * 64 <= chunk_vecs; it += UNROLL * 64) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(base + it + u * 64);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc_word(v[u].x, lo[0], hi[0], ms[0]); acc_word(v[u].y, lo[1], hi[1], ms[1]);
        acc_word(v[u].z, lo[2], hi[2], ms[2]); acc_word(v[u].w, lo[3], hi[3], ms[3]);
      }
    }
  }
  uint32_t r = lo[0] ^ lo[1] ^ lo[2] ^ lo[3] ^ hi[0] ^ hi[1] ^ hi[2] ^ hi[3] ^ ms[0] ^ ms[1] ^ ms[2] ^ ms[3];
  if (r == 0x12345678u) out[0] = r;
}

__global__ void k_fill_random(uint32_t* p, int64_t n_words) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_words; i += stride) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    // genotype-like bytes: mostly 0, some 1 and 2
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) { uint32_t r = (z >> (b * 16)) & 0xFFFF; w |= (r < 6000 ? 2u : r < 20000 ? 1u : 0u) << (8 * b); }
    p[i] = w;
  }
}

template <typename F>
double time_ms(F&& launch, int reps = 8) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  double best = 1e9;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

// E: the "lane = site" layout: inside a tile, groups of 16 individuals, each group stored as
// [64 sites][16 individuals] (1 KiB, one wave load), a last narrower group of 4 / 8 / 16 columns.
// Every lane owns one site: v_dot4_u32_u8 sums its 16 bytes, no cross-lane step, no LDS.
__device__ __forceinline__ void acc16(const u32x4& v, uint32_t& sum, uint32_t& called) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t sel = ~(w[j] >> 7) & 0x01010101u;  // 1 per called byte
    sum = __builtin_amdgcn_udot4(w[j], sel, sum, false);
    called += __builtin_popcount(sel);
  }
}

template <int U, int WRITE>
__global__ __launch_bounds__(64) void k_dot4(const int8_t* __restrict__ ref, const int8_t* __restrict__ tgt, int64_t n_tiles,
                                             int n_ind, int64_t tile_stride, double* __restrict__ freq, uint8_t* __restrict__ flags) {
  const int lane = threadIdx.x;
  const int G = n_ind >> 4, rem = n_ind & 15;
  const int kp = rem == 0 ? 0 : rem <= 4 ? 4 : rem <= 8 ? 8 : 16;
  uint32_t held = 0; int n_held = 0; int64_t held_tile[8];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t sum[2], called[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int8_t* tb = (p ? tgt : ref) + tile * tile_stride;
      const u32x4* base = reinterpret_cast<const u32x4*>(tb) + lane;
      uint32_t s = 0, c = 0;
      int g = 0;
      for (; g + U <= G; g += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + (g + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) acc16(v[u], s, c);
      }
      {  // tail: < U full groups (clamped, masked) and the narrow group, one batch
        u32x4 v[U];
        u32x4 t = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
        for (int u = 0; u < U - 1; ++u) v[u] = __builtin_nontemporal_load(base + min(g + u, G - 1) * 64);
        const int8_t* nb = tb + (int64_t)G * 1024;
        if (kp == 8) { const u32x2 q = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(nb) + lane); t.x = q.x; t.y = q.y; }
        else if (kp == 4) t.x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(nb) + lane);
        else if (kp == 16) t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(nb) + lane);
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
          if (g + u >= G) v[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
          acc16(v[u], s, c);
        }
        acc16(t, s, c);
      }
      sum[p] = s; called[p] = c;
    }
    const int64_t site = tile * 64 + lane;
    const double fr = called[0] ? (double)sum[0] / ((double)called[0] * 2.0) : __builtin_nan("");
    const double ft = called[1] ? (double)sum[1] / ((double)called[1] * 2.0) : __builtin_nan("");
    const uint8_t fl = (fr < 0.01 && ft > 0.5) ? 1 : 0;
    if (WRITE == 0) { if (fr == 123.0) { freq[site] = ft; flags[site] = fl; } }
    else if (WRITE == 1) { freq[site] = ft; flags[site] = fl; }
    else if (WRITE == 2) { flags[site] = fl; if (fr == 123.0) freq[site] = ft; }
    else if (WRITE == 3) { freq[site] = ft; if (fr == 123.0) flags[site] = fl; }
    else if (WRITE == 4) { __builtin_nontemporal_store(ft, freq + site); __builtin_nontemporal_store(fl, flags + site); }
    else if (WRITE == 5) { flags[site] = fl; if (fl) freq[site] = ft; }
    else if (WRITE == 7 || WRITE == 8) {  // hold the flag bytes of 4 tiles in a register, store them back to back
      constexpr int K = 4;
      held |= (uint32_t)fl << (8 * n_held);
      held_tile[n_held] = tile;
      if (++n_held == K) {
#pragma unroll
        for (int k = 0; k < K; ++k) flags[held_tile[k] * 64 + lane] = (held >> (8 * k)) & 0xFF;
        held = 0; n_held = 0;
      }
      if (fl) freq[site] = ft;
    }
    else if (WRITE == 9) { __builtin_nontemporal_store(fl, flags + site); if (fl) freq[site] = ft; }
    else if (WRITE == 6) { const uint64_t m = __ballot(fl); if (lane == 0) reinterpret_cast<uint64_t*>(flags)[tile] = m; if (fl) freq[site] = ft; }
  }
}

// F: cache-policy bits of the load instruction (gfx950: sc0 / sc1 / nt), 5 wave loads in flight
template <int POL>
__global__ __launch_bounds__(64) void k_policy(const u32x4* __restrict__ src, int64_t n, uint32_t* out) {
  const int lane = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const int64_t n_runs = n / 8000;
  for (int64_t c = blockIdx.x; c < n_runs; c += gridDim.x) {
    const u32x4* base = src + c * 8000 + lane;
    for (int it = 0; it < 125; it += 5) {
      u32x4 v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const u32x4* p = base + (it + u) * 64;
        if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v[u]) : "v"(p) : "memory");
        if (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v[u]) : "v"(p) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 5; ++u) acc ^= v[u];
    }
  }
  uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x12345678u) out[0] = r;
}

int main(int argc, char** argv) {
  // usage: stream_probe [section ...]   sections: stride chunk glds swar writes policy (default: all)
  auto want = [&](const char* name) {
    if (argc < 2) return true;
    for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], name)) return true;
    return false;
  };
  const int64_t n_tiles = 156250;                // the C3 block: 10^7 sites
  const int64_t pop_bytes = n_tiles * 64000ll;   // 1000 individuals x 64 sites per tile
  int8_t* big; double* freq; uint8_t* flags; uint32_t* out;
  CK(hipMalloc(&big, 2 * pop_bytes + 4096)); CK(hipMalloc(&freq, n_tiles * 64 * 8)); CK(hipMalloc(&flags, n_tiles * 64)); CK(hipMalloc(&out, 4));
  hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)big, 2 * pop_bytes / 4);
  CK(hipDeviceSynchronize());
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cu = pr.multiProcessorCount;
  const double bytes = 2.0 * pop_bytes;
  const int64_t n = 2 * pop_bytes / 16;
  const u32x4* src = (const u32x4*)big;
  auto rep = [&](const char* name, double ms) { printf("%-56s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
  if (want("stride")) {  // thread-strided grid loops
#define A(U, NT, G) rep("gridstride U=" #U " nt=" #NT " grid=cu*" #G, time_ms([&] { hipLaunchKernelGGL((k_gridstride<U, NT>), dim3(cu * G), dim3(256), 0, 0, src, n, out); }))
    A(4, true, 8); A(4, false, 8); A(8, true, 8); A(8, true, 16);
  }
  if (want("chunk")) {  // one wave per contiguous run (chunk sizes divide 20.0 GB exactly with these unrolls)
#define B(U, NT, G, CH) rep("wavechunk " #CH "x16B U=" #U " nt=" #NT " grid=cu*" #G, time_ms([&] { hipLaunchKernelGGL((k_wavechunk<U, NT>), dim3(cu * G), dim3(64), 0, 0, src, n, (int64_t)(CH), out); }))
    B(4, true, 32, 8000); B(4, true, 32, 16000); B(5, true, 32, 8000); B(8, true, 16, 16000); B(8, true, 32, 16000); B(8, false, 32, 16000);
  }
  if (want("glds")) {  // LDS-DMA ring instead of registers
#define Cc(S, AUX, G) rep("glds ring slots=" #S " aux=" #AUX " grid=cu*" #G, time_ms([&] { hipLaunchKernelGGL((k_glds<S, AUX>), dim3(cu * G), dim3(64), 0, 0, src, n, (int64_t)16000, out); }))
    Cc(8, 0, 16); Cc(8, 2, 16); Cc(16, 2, 16); Cc(4, 2, 32);
  }
  if (want("swar")) {  // + the byte arithmetic of site_counts
#define D(U, W, G, CH) rep("swar U=" #U " minwaves=" #W " grid=cu*" #G " run=" #CH "x16B", time_ms([&] { hipLaunchKernelGGL((k_swar<U, W>), dim3(cu * G), dim3(64), 0, 0, src, n, (int64_t)(CH), out); }))
    D(4, 1, 16, 16000); D(4, 1, 32, 16000); D(4, 1, 64, 16000); D(8, 1, 32, 16000); D(4, 8, 32, 16000);
  }
  if (want("writes")) {  // lane = site layout, per-site outputs: 0 none, 1 freq+flags, 2 flags, 3 freq, 4 both nt,
                         // 5 flags + freq where flagged, 6 ballot-packed flags, 7 flags of 4 tiles batched, 9 nt flags + sparse freq
#define E(U, G, W) rep("dot4 U=" #U " grid=cu*" #G " write=" #W, time_ms([&] { hipLaunchKernelGGL((k_dot4<U, W>), dim3(cu * G), dim3(64), 0, 0, big, big + pop_bytes, n_tiles, 1000, (int64_t)64000, freq, flags); }))
    E(8, 32, 0); E(8, 32, 1); E(8, 32, 2); E(8, 32, 3); E(8, 32, 4); E(8, 32, 5); E(8, 32, 6); E(8, 32, 7); E(8, 32, 9); E(16, 32, 0); E(4, 32, 0);
  }
  if (want("policy")) {  // cache-policy bits of the load
#define P(POL, NAME) rep("policy " NAME " U=5 run=8000x16B grid=cu*32", time_ms([&] { hipLaunchKernelGGL((k_policy<POL>), dim3(cu * 32), dim3(64), 0, 0, src, n, out); }))
    P(0, "default"); P(1, "nt"); P(2, "sc0"); P(3, "sc1"); P(4, "sc0 sc1"); P(5, "sc0 nt"); P(6, "sc1 nt"); P(7, "sc0 sc1 nt");
  }
  return 0;
}
