// libsaihip: sai's sliding-window U/Q statistics as hand-written HIP for MI355X (gfx950, CDNA4).
//
// Translation units (see DESIGN.md for the roofline of each kernel):
//   host_core.cpp  errors, version, host side of the synthetic generator (plain C++, also built
//                  alone with the sanitizers)
//   core.hip       context, tile_from_site_major (ingest: [site][ind] int8 -> tiled SoA)
//   site_pass.hip  site_counts (the HBM-bound byte reduction), site_flags, the fused site pass
//   packed2.hip    the optional 2-bit layout and its site pass
//   windows.hip    window_bounds, window statistics (U count, numpy-'linear' quantile, lists)
//   single_window.hip  one window per call: the plugin classes' U / Q evaluation in one entry
//   fourpop.hip    fd / df / Danc / Dplus: frequencies and numpy-ordered pattern sums
//   dd.hip         DD: per-site city-block terms and their window means
//   synth.hip      counter-based synthetic data (synth-v1) and the stream-read probe
//   vcf_ingest.cpp host-side VCF / BED reader
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared (see __graft_entry__.py).
// -ffp-contract=off is part of the contract: the f64 arithmetic must round exactly like numpy's
// (separate multiply and add in the quantile lerp, IEEE division for the frequencies).

#include "common.hpp"

namespace {

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

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

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

int sai_device_identity(int device, char* bus_id_out, int32_t bus_id_capacity, char* uuid_hex_out, int32_t uuid_capacity) {
  if (!bus_id_out || bus_id_capacity < 16) return fail(SAI_ERR_ARG, "bus_id_out needs room for 16 bytes");
  if (uuid_hex_out && uuid_capacity < 33) return fail(SAI_ERR_ARG, "uuid_hex_out needs room for 33 bytes");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
    return fail(SAI_ERR_NO_DEVICE, "device %d is not one of the %d visible HIP devices", device, n);
  SAI_HIP(hipDeviceGetPCIBusId(bus_id_out, bus_id_capacity, device));
  if (uuid_hex_out) {
    hipUUID id;
    SAI_HIP(hipDeviceGetUuid(&id, device));
    for (int i = 0; i < 16; ++i) std::snprintf(uuid_hex_out + 2 * i, 3, "%02x", static_cast<unsigned>(static_cast<unsigned char>(id.bytes[i])));
  }
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
  c->sw_dev = c->sw_host = nullptr;
  c->sw_dev_cap = c->sw_host_cap = 0;
  c->next_start = c->next_stop = nullptr;
  if (hipMalloc(&c->probe_partials, sizeof(uint32_t) * c->n_cu * kProbeWavesPerCu) != hipSuccess) {
    delete c;
    return fail(SAI_ERR_HIP, "hipMalloc of the context scratch failed");
  }
  *ctx_out = c;
  return SAI_OK;
}

int sai_event_create(sai_ctx* ctx, void** event_out) {
  if (int rc = enter(ctx)) return rc;
  if (!event_out) return fail(SAI_ERR_ARG, "event_out is NULL");
  hipEvent_t e = nullptr;
  SAI_HIP(hipEventCreate(&e));
  *event_out = e;
  return SAI_OK;
}

int sai_event_destroy(void* event) {
  if (event) SAI_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return SAI_OK;
}

int sai_event_synchronize(void* event) {
  if (!event) return fail(SAI_ERR_ARG, "event is NULL");
  SAI_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return SAI_OK;
}

int sai_event_query(void* event, int32_t* done_out) {
  if (!event || !done_out) return fail(SAI_ERR_ARG, "NULL argument");
  const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
  if (e != hipSuccess && e != hipErrorNotReady) return fail(SAI_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
  *done_out = e == hipSuccess ? 1 : 0;
  return SAI_OK;
}

int sai_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out) {
  if (!start_event || !stop_event || !ms_out) return fail(SAI_ERR_ARG, "NULL argument");
  SAI_HIP(hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)));
  return SAI_OK;
}

int sai_ctx_destroy(sai_ctx* ctx) {
  if (ctx && ctx->probe_partials) (void)hipFree(ctx->probe_partials);
  if (ctx && ctx->sw_dev) (void)hipFree(ctx->sw_dev);
  if (ctx && ctx->sw_host) (void)hipHostFree(ctx->sw_host);
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

}  // extern "C"
