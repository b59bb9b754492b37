// Narrowing of a reference-style genotype matrix to int8 (sai_narrow_to_int8): the reference holds
// int64 [sites][individuals] (utils.py:410), the device layout is int8.  Host side, memory-bound: one
// pass that reads the wide matrix once and writes an eighth of it, fanned out over persistent worker
// threads (the per-window plugin route calls this two or three times per window -- 32 MB of int64 per
// C3 window -- so a thread is not created per call, and a 16 MB matrix is cut into 8 pieces, not the 2
// of round 2, which left it at 6 GB/s).

#include <algorithm>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <type_traits>

#include "host_threads.hpp"
#include "saihip.h"

extern "C" int sai_set_error(int code, const char* fmt, ...);  // host_core.cpp

namespace {

template <typename T>
bool narrow_rows(const char* src, int64_t row_stride, int64_t r0, int64_t r1, int64_t n_cols, int8_t* dst) {
  bool too_big = false;
  for (int64_t r = r0; r < r1; ++r) {
    const T* row = reinterpret_cast<const T*>(src + r * row_stride);
    int8_t* out = dst + r * n_cols;
    T hi = 0;  // running maximum: one compare per element instead of a branch
    for (int64_t c = 0; c < n_cols; ++c) {
      const T v = row[c];
      hi = v > hi ? v : hi;
      if constexpr (std::is_signed<T>::value) out[c] = static_cast<int8_t>(v < static_cast<T>(-128) ? static_cast<T>(-128) : v);
      else out[c] = static_cast<int8_t>(v);
    }
    too_big = too_big || hi > static_cast<T>(127);
  }
  return !too_big;
}


// The reference's own dtype gets a body per vector width of the host, picked once at run time (the
// generic build is SSE2: a fraction of the AVX-512 rate on the same core).
__attribute__((always_inline)) inline bool narrow_i64_body(const char* src, int64_t row_stride, int64_t r0, int64_t r1,
                                                           int64_t n_cols, int8_t* dst) {
  int64_t hi_all = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t* row = reinterpret_cast<const int64_t*>(src + r * row_stride);
    int8_t* out = dst + r * n_cols;
    int64_t hi = 0;
    for (int64_t c = 0; c < n_cols; ++c) {
      const int64_t v = row[c];
      hi = v > hi ? v : hi;
      out[c] = static_cast<int8_t>(v < -128 ? -128 : v);
    }
    hi_all = hi > hi_all ? hi : hi_all;
  }
  return hi_all <= 127;
}
__attribute__((target("avx512f,avx512bw,avx512vl"))) bool narrow_i64_avx512(const char* s, int64_t st, int64_t r0, int64_t r1, int64_t n, int8_t* d) {
  return narrow_i64_body(s, st, r0, r1, n, d);
}
__attribute__((target("avx2"))) bool narrow_i64_avx2(const char* s, int64_t st, int64_t r0, int64_t r1, int64_t n, int8_t* d) {
  return narrow_i64_body(s, st, r0, r1, n, d);
}
bool narrow_i64_generic(const char* s, int64_t st, int64_t r0, int64_t r1, int64_t n, int8_t* d) {
  return narrow_i64_body(s, st, r0, r1, n, d);
}
using NarrowFn = bool (*)(const char*, int64_t, int64_t, int64_t, int64_t, int8_t*);
NarrowFn pick_narrow_i64() {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) return narrow_i64_avx512;
  if (__builtin_cpu_supports("avx2")) return narrow_i64_avx2;
  return narrow_i64_generic;
}
bool narrow_rows_i64(const char* src, int64_t row_stride, int64_t r0, int64_t r1, int64_t n_cols, int8_t* dst) {
  static const NarrowFn fn = pick_narrow_i64();
  return fn(src, row_stride, r0, r1, n_cols, dst);
}

constexpr int kPoolThreads = 16;
constexpr int64_t kElementsPerThread = 1 << 18;  // 2 MB of int64 per piece

// One pool per process, created on first use; run() is serialised (callers come from one Python
// thread per process, but nothing here relies on that).
ProcessPool& pool() {
  static ProcessPool p(kPoolThreads);
  return p;
}

int narrow_impl(const void* src, int32_t itemsize, int32_t is_signed, int64_t n_rows, int64_t n_cols,
                int64_t row_stride_bytes, int8_t* dst, int32_t n_threads) {
  if (n_rows < 0 || n_cols < 0) return sai_set_error(SAI_ERR_ARG, "negative shape");
  if (n_rows == 0 || n_cols == 0) return SAI_OK;
  if (!src || !dst) return sai_set_error(SAI_ERR_ARG, "NULL buffer");
  if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return sai_set_error(SAI_ERR_ARG, "itemsize must be 1, 2, 4 or 8");
  const int64_t total = n_rows * n_cols;
  const int nt = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({static_cast<int64_t>(n_threads), static_cast<int64_t>(kPoolThreads), n_rows, 1 + total / kElementsPerThread})));
  char ok[kPoolThreads];
  std::fill(ok, ok + kPoolThreads, 1);
  const char* base = static_cast<const char*>(src);
  auto work = [&](int t) {  // narrow_rows touches only the caller's buffers: nothing throws in here
    const int64_t r0 = n_rows * t / nt, r1 = n_rows * (t + 1) / nt;
    bool good = true;
    switch (itemsize * 2 + (is_signed ? 1 : 0)) {
      case 3: good = narrow_rows<int8_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 2: good = narrow_rows<uint8_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 5: good = narrow_rows<int16_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 4: good = narrow_rows<uint16_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 9: good = narrow_rows<int32_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 8: good = narrow_rows<uint32_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      case 17: good = narrow_rows_i64(base, row_stride_bytes, r0, r1, n_cols, dst); break;
      default: good = narrow_rows<uint64_t>(base, row_stride_bytes, r0, r1, n_cols, dst); break;
    }
    ok[t] = good ? 1 : 0;
  };
  if (nt == 1 || !pool().usable()) {  // a forked child has the pool object but not its threads
    for (int t = 0; t < nt; ++t) work(t);
  } else {
    pool().run(nt, work);
  }
  for (int t = 0; t < nt; ++t)
    if (!ok[t]) return sai_set_error(SAI_ERR_UNSUPPORTED, "dosage above 127 is not representable in the int8 device layout");
  return SAI_OK;
}

}  // namespace

extern "C" int sai_narrow_to_int8(const void* src, int32_t itemsize, int32_t is_signed, int64_t n_rows, int64_t n_cols,
                                  int64_t row_stride_bytes, int8_t* dst, int32_t n_threads) {
  try {  // no C++ exception crosses the C ABI (thread creation of the pool may fail)
    return narrow_impl(src, itemsize, is_signed, n_rows, n_cols, row_stride_bytes, dst, n_threads);
  } catch (const std::bad_alloc&) {
    return sai_set_error(SAI_ERR_HIP, "sai_narrow_to_int8: out of host memory");
  } catch (const std::exception& e) {
    return sai_set_error(SAI_ERR_HIP, "sai_narrow_to_int8: %s", e.what());
  }
}
