// sai_text_line_starts / sai_text_line_heads: the line structure of a text batch, found on the GPU.
//
// After sai_inflate_bgzf the text of a bgzip VCF lies in HBM.  The host needs very little of it to
// index the records (sai_vcf_index_heads, vcf_ingest.cpp): where every line starts and its fixed
// columns CHROM .. FORMAT -- some tens of bytes of a line that is kilobytes long.  These kernels find
// the newlines (count per 4 KiB block, scan, scatter), measure how far the ninth tab of every line
// lies, and gather the first `head_bytes` of every line into a dense array, so that a few MB cross
// PCIe instead of the whole text.

#include "common.hpp"

namespace {

constexpr int kBlockBytes = 4096;  // text per workgroup of 256 threads (16 bytes each)
constexpr int kScanMax = 4096;     // how far into a line the ninth tab is searched

struct LineArgs {
  const uint8_t* text;
  const uint8_t* aligned;  // text rounded down to 16 bytes
  int mis;                 // text - aligned
  int64_t n_bytes;
  int64_t n_blocks;
  int32_t* block_count;  // n_blocks + 1 (exclusive offsets after the scan)
  int64_t* line_start;   // capacity + 1
  int64_t capacity;
  int32_t* line_info;  // capacity: fixed_len | ends_with_cr << 31
  int32_t* info;       // [0] = newlines, [1] = max fixed_len, [2] = 1 when capacity was too small
};

// Newlines among the 16 bytes at `at` of the 16-byte-aligned view (`text` minus its misalignment `mis`);
// bytes outside [mis, mis + n) do not count.  One aligned 16-byte load per thread.
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t* aligned, int mis, int64_t n, int64_t at) {
  const int64_t lo = mis, hi = mis + n;
  if (at + 16 <= lo || at >= hi) return 0;
  const u32x4 w = *reinterpret_cast<const u32x4*>(aligned + at);
  const uint32_t v[4] = {w.x, w.y, w.z, w.w};
  uint32_t mask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (((v[j] >> (8 * b)) & 0xFFu) == '\n') mask |= 1u << (4 * j + b);
  if (at < lo) mask &= ~0u << (lo - at);
  if (at + 16 > hi) mask &= (1u << (hi - at)) - 1u;
  return mask;
}

__global__ __launch_bounds__(256) void count_newlines_kernel(LineArgs a) {
  __shared__ int wave_sum[4];
  const int64_t at = static_cast<int64_t>(blockIdx.x) * kBlockBytes + threadIdx.x * 16;
  int c = __popc(newline_mask16(a.aligned, a.mis, a.n_bytes, at));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) a.block_count[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// exclusive scan of the block counts by one workgroup (a batch has at most a few 10^4 blocks)
__global__ __launch_bounds__(1024) void scan_blocks_kernel(LineArgs a) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (a.n_blocks + 1023) / 1024;
  const int64_t lo = t * per, hi = lo + per < a.n_blocks ? lo + per : a.n_blocks;
  int64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += a.block_count[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int64_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = t ? part[t - 1] : 0;
  for (int64_t i = lo; i < hi; ++i) {
    const int c = a.block_count[i];
    a.block_count[i] = static_cast<int32_t>(run);
    run += c;
  }
  if (t == 1023) {
    a.block_count[a.n_blocks] = static_cast<int32_t>(part[1023]);
    a.info[0] = static_cast<int32_t>(part[1023]);
    a.info[1] = 0;
    a.info[2] = part[1023] + 1 > a.capacity ? 1 : 0;
  }
  if (t == 0) a.line_start[0] = 0;
}

__global__ __launch_bounds__(256) void write_starts_kernel(LineArgs a) {
  __shared__ int wave_sum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t at = static_cast<int64_t>(blockIdx.x) * kBlockBytes + threadIdx.x * 16;
  const uint32_t mask = newline_mask16(a.aligned, a.mis, a.n_bytes, at);
  const int c = __popc(mask);
  int incl = c;  // inclusive prefix over the wavefront
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  int64_t idx = a.block_count[blockIdx.x] + (incl - c);
  for (int w = 0; w < wave; ++w) idx += wave_sum[w];
  uint32_t m = mask;
  while (m) {
    const int k = __ffs(m) - 1;
    m &= m - 1;
    if (idx + 1 <= a.capacity) a.line_start[idx + 1] = at + k + 1 - a.mis;
    ++idx;
  }
}

// per complete line: bytes up to and including the ninth tab (the fixed columns), 1 for a '#' line
__global__ __launch_bounds__(256) void line_fixed_kernel(LineArgs a) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n_lines = a.info[0] < a.capacity ? a.info[0] : a.capacity;
  if (i >= n_lines) return;
  const int64_t s = a.line_start[i], e = a.line_start[i + 1] - 1;  // [s, e) without the newline
  const int cr = (e > s && a.text[e - 1] == '\r') ? 1 : 0;
  const int64_t le = e - cr;
  int fixed = 1;
  if (le > s && a.text[s] != '#') {
    const int64_t stop = le - s < kScanMax ? le : s + kScanMax;
    int tabs = 0;
    int64_t p = s;
    for (; p < stop; ++p)
      if (a.text[p] == '\t' && ++tabs == 9) break;
    if (tabs == 9) fixed = static_cast<int>(p - s) + 1;
    else fixed = (stop == le) ? static_cast<int>(le - s) + 1 : kScanMax + 1;  // fewer than ten columns / out of reach
  }
  a.line_info[i] = static_cast<int32_t>(static_cast<uint32_t>(fixed) | (static_cast<uint32_t>(cr) << 31));
  atomicMax(&a.info[1], fixed);
}

struct HeadArgs {
  const uint8_t* text;
  int64_t n_bytes;
  const int64_t* line_start;
  int64_t n_lines;
  int32_t head_bytes;  // multiple of 4
  uint8_t* heads;
};

__global__ __launch_bounds__(256) void gather_heads_kernel(HeadArgs a) {
  const int words = a.head_bytes >> 2;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t line = t / words;
  if (line >= a.n_lines) return;
  const int w = static_cast<int>(t - line * words);
  const int64_t s = a.line_start[line] + 4 * w, e = a.line_start[line + 1];  // the newline included
  uint32_t v = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint8_t c = s + b < e ? a.text[s + b] : static_cast<uint8_t>('\n');
    v |= static_cast<uint32_t>(c) << (8 * b);
  }
  reinterpret_cast<uint32_t*>(a.heads)[line * words + w] = v;
}

}  // namespace

extern "C" {

int sai_text_line_starts(sai_ctx* ctx, const char* text, int64_t n_bytes, int64_t line_capacity, int64_t* line_start,
                         int32_t* line_info, int32_t* block_scratch, int32_t* info, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_bytes < 0 || line_capacity < 1) return fail(SAI_ERR_ARG, "bad size");
  if (!line_start || !line_info || !block_scratch || !info || (n_bytes > 0 && !text)) return fail(SAI_ERR_ARG, "NULL buffer");
  if (n_bytes >= (int64_t(1) << 40)) return fail(SAI_ERR_UNSUPPORTED, "text batch too large");
  LineArgs a;
  a.text = reinterpret_cast<const uint8_t*>(text);
  a.mis = static_cast<int>(reinterpret_cast<uintptr_t>(text) & 15u);
  a.aligned = a.text - a.mis;
  a.n_bytes = n_bytes;
  a.n_blocks = n_bytes ? (n_bytes + a.mis + kBlockBytes - 1) / kBlockBytes : 0;
  a.block_count = block_scratch;
  a.line_start = line_start;
  a.capacity = line_capacity;
  a.line_info = line_info;
  a.info = info;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.n_blocks > 0) hipLaunchKernelGGL(count_newlines_kernel, dim3(static_cast<unsigned>(a.n_blocks)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, a);
  if (a.n_blocks > 0) {
    hipLaunchKernelGGL(write_starts_kernel, dim3(static_cast<unsigned>(a.n_blocks)), dim3(256), 0, st, a);
    const int64_t max_lines = line_capacity < n_bytes ? line_capacity : n_bytes;
    hipLaunchKernelGGL(line_fixed_kernel, dim3(static_cast<unsigned>((max_lines + 255) / 256)), dim3(256), 0, st, a);
  }
  return check_launch("text_line_starts");
}

int sai_text_line_heads(sai_ctx* ctx, const char* text, int64_t n_bytes, const int64_t* line_start, int64_t n_lines,
                        int32_t head_bytes, void* heads, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_lines < 0 || n_bytes < 0 || head_bytes < 4 || (head_bytes & 3)) return fail(SAI_ERR_ARG, "head_bytes must be a positive multiple of 4");
  if (n_lines == 0) return SAI_OK;
  if (!text || !line_start || !heads) return fail(SAI_ERR_ARG, "NULL buffer");
  if (reinterpret_cast<uintptr_t>(heads) & 3u) return fail(SAI_ERR_ARG, "heads must be 4-byte aligned");
  HeadArgs a;
  a.text = reinterpret_cast<const uint8_t*>(text);
  a.n_bytes = n_bytes;
  a.line_start = line_start;
  a.n_lines = n_lines;
  a.head_bytes = head_bytes;
  a.heads = static_cast<uint8_t*>(heads);
  const int64_t threads = n_lines * (head_bytes >> 2);
  hipLaunchKernelGGL(gather_heads_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("text_line_heads");
}

}  // extern "C"
