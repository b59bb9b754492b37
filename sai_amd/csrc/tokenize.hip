// sai_tokenize_gt: the sample columns of VCF record lines -> int8 dosages, on the GPU.
//
// The host side (sai_vcf_stream_*, vcf_ingest.cpp) has indexed the lines: where the first sample
// column starts, how long the rest of the line is, which FORMAT sub-field is GT, whether the line is
// flipped by the ancestral-allele rule.  One wavefront owns one line and walks it 256 bytes per
// step (one aligned 32-bit word per lane): four ballots mark the tabs, a prefix popcount gives every
// byte its column number, and the lane that holds the first byte of a selected column parses that
// field -- the same state machine as parse_lines() of the host reader, so both give the same bytes
// (tests/test_ingest_device.py compares them on every awkward file of the host reader's tests).

#include "common.hpp"

namespace {

struct TokArgs {
  const char* text;
  int64_t n_text;  // bytes that may be read (the caller pads the buffer to a multiple of 4)
  int64_t n_lines;
  const int64_t* off;
  const int32_t* len;
  const uint8_t* flip;
  const uint8_t* gi;
  int32_t n_cols;
  const int32_t* slot_of_col;
  int32_t n_out;
  const int32_t* ploidy;
  int8_t* out;
  int32_t* status;
};

// One sample field starting at g (le = end of the line): skip to the GT sub-field, read up to `pl`
// alleles, pad with missing ones.  Returns false where the host reader reports an error.
__device__ __forceinline__ bool parse_field(const char* g, const char* le, int gi, int pl, int* dosage, int* flipped) {
  for (int k = 0; k < gi; ++k) {
    while (g < le && *g != ':' && *g != '\t') ++g;
    if (g < le && *g == ':') ++g;
  }
  int n = 0, d = 0, fd = 0;
  for (;;) {
    const char ch = g < le ? *g : '\t';
    int a;
    if (ch == '.') {
      a = -1;
      ++g;
    } else if (ch >= '0' && ch <= '9') {
      a = 0;
      do {
        a = a * 10 + (*g++ - '0');
        if (a > 100000) return false;  // far outside int8 already; keeps the product from wrapping
      } while (g < le && *g >= '0' && *g <= '9');
    } else if (ch == '|' || ch == '/' || ch == ':' || ch == '\t') {
      a = -1;  // empty allele
    } else {
      return false;  // unparsable genotype
    }
    if (n < pl) {  // alleles beyond the ploidy asked for are ignored
      d += a;
      fd += a >= 1 ? a - 1 : 1 - a;
      ++n;
    }
    if (g < le && (*g == '|' || *g == '/')) {
      ++g;
      continue;
    }
    break;
  }
  for (; n < pl; ++n) {  // fewer alleles than the ploidy asked for: padded with missing
    d -= 1;
    fd += 2;
  }
  if (d > 127 || fd > 127 || d < -128) return false;
  *dosage = d;
  *flipped = fd;
  return true;
}

constexpr int kTokWaves = 4;

__global__ __launch_bounds__(64 * kTokWaves) void tokenize_gt_kernel(TokArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t line = static_cast<int64_t>(blockIdx.x) * kTokWaves + (threadIdx.x >> 6);
  if (line >= a.n_lines) return;  // whole wave
  const int64_t off = a.off[line];
  const int64_t end = off + a.len[line];
  const char* le = a.text + end;
  const int gi = a.gi[line];
  const bool flip = a.flip[line] != 0;
  int8_t* row = a.out + line * a.n_out;
  const unsigned long long below = (1ull << lane) - 1ull;
  bool bad = false;
  int col_base = 0;              // tabs before this step = column of its first byte
  unsigned long long carry = 0;  // bit 0: the byte before this step's first byte is a tab

  auto take = [&](int64_t p, int col) {  // the field that starts at byte p is sample column `col`
    if (col >= a.n_cols) return;
    const int slot = a.slot_of_col[col];
    if (slot < 0) return;
    int d = 0, fd = 0;
    if (!parse_field(a.text + p, le, gi, a.ploidy[slot], &d, &fd)) {
      bad = true;
      return;
    }
    row[slot] = static_cast<int8_t>(flip ? fd : d);
  };

  for (int64_t c = off & ~int64_t{3}; c < end; c += 256) {
    const int64_t wpos = c + 4 * lane;
    uint32_t w = 0;
    if (wpos + 4 <= a.n_text) w = *reinterpret_cast<const uint32_t*>(a.text + wpos);
    unsigned long long m[4];
    bool tab[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t p = wpos + k;
      tab[k] = p >= off && p < end && ((w >> (8 * k)) & 0xFFu) == 9u;
      m[k] = __ballot(tab[k]);
    }
    const int before = __popcll(m[0] & below) + __popcll(m[1] & below) + __popcll(m[2] & below) + __popcll(m[3] & below);
    int own = 0;  // tabs in this lane's word before byte k
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t p = wpos + k;
      if (p >= off && p < end) {
        bool starts;
        if (p == off) starts = true;
        else if (k > 0) starts = tab[k - 1];
        else starts = lane > 0 ? ((m[3] >> (lane - 1)) & 1ull) != 0 : (carry & 1ull) != 0;
        if (starts) take(p, col_base + before + own);
      }
      own += tab[k] ? 1 : 0;
    }
    col_base += __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
    carry = m[3] >> 63;
  }
  // a line whose sample section is empty, or ends in a tab, closes with one EMPTY field at `end`
  if (lane == 0) {
    const bool open_end = end == off || a.text[end - 1] == '\t';
    if (open_end) take(end, col_base);
  }
  // the host reader walks columns 0..max_col and refuses a line that has fewer
  const int n_fields = col_base + 1;
  if (a.n_cols > n_fields) bad = true;
  const unsigned long long any_bad = __ballot(bad);
  if (lane == 0) a.status[line] = any_bad ? 1 : 0;
}

}  // namespace

extern "C" int sai_tokenize_gt(sai_ctx* ctx, const char* text, int64_t n_text_bytes, int64_t n_lines, const int64_t* line_off,
                               const int32_t* line_len, const uint8_t* line_flip, const uint8_t* line_gi, int32_t n_cols,
                               const int32_t* slot_of_col, int32_t n_out, const int32_t* ploidy_of_slot, int8_t* out,
                               int32_t* status, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_lines < 0 || n_text_bytes < 0 || n_cols < 0 || n_out < 1) return fail(SAI_ERR_ARG, "size out of range");
  if (n_lines == 0) return SAI_OK;
  if (!text || !line_off || !line_len || !line_flip || !line_gi || !out || !status || !ploidy_of_slot ||
      (n_cols > 0 && !slot_of_col))
    return fail(SAI_ERR_ARG, "NULL buffer");
  if (reinterpret_cast<uintptr_t>(text) & 3u) return fail(SAI_ERR_ARG, "text must be 4-byte aligned");
  if (n_text_bytes & 3) return fail(SAI_ERR_ARG, "n_text_bytes must be a multiple of 4 (pad the buffer)");
  const int64_t grid = (n_lines + kTokWaves - 1) / kTokWaves;
  if (grid > 0x7FFFFFFFll) return fail(SAI_ERR_UNSUPPORTED, "too many lines for one launch");
  TokArgs a;
  a.text = text;
  a.n_text = n_text_bytes;
  a.n_lines = n_lines;
  a.off = line_off;
  a.len = line_len;
  a.flip = line_flip;
  a.gi = line_gi;
  a.n_cols = n_cols;
  a.slot_of_col = slot_of_col;
  a.n_out = n_out;
  a.ploidy = ploidy_of_slot;
  a.out = out;
  a.status = status;
  hipLaunchKernelGGL(tokenize_gt_kernel, dim3(static_cast<unsigned>(grid)), dim3(64 * kTokWaves), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("tokenize_gt");
}
