// synth-v1 synthetic data generator and the stream-read bandwidth probe (no reference counterpart).

#include "common.hpp"
#include "synth_core.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// synth-v1: counter-based synthetic data (identical on host and device)
// ------------------------------------------------------------------------------------------

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

}  // extern "C"
