// Output text of `sai score` straight from the numeric window results (host only): the TSV rows and
// the .U.log / .Q.log rows FeaturePreprocessor.process_items writes (reference:
// sai/preprocessors/feature_preprocessor.py:193-258), formatted without building a Python object per
// window.  Numbers print as Python's str() prints them -- ints in decimal, doubles with the shortest
// digits that round-trip, fixed notation for 1e-4 <= |x| < 1e16 and d.ddde+XX otherwise, "nan",
// "inf" -- which tests/test_text_out.py checks against str() itself on millions of doubles.

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>

#include <cerrno>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include "host_threads.hpp"

#include "saihip.h"

extern "C" int sai_set_error(int code, const char* fmt, ...);

namespace {

// str(float) / str(numpy.float64): repr with the shortest round-trip digits
void append_double(std::string& out, double v) {
  if (v != v) { out += "nan"; return; }
  if (std::isinf(v)) { out += v < 0 ? "-inf" : "inf"; return; }
  char buf[40];
  const auto res = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
  const char* p = buf;
  const char* end = res.ptr;
  if (*p == '-') { out += '-'; ++p; }
  // d[.ddd]e[+-]XX
  const char* e = p;
  while (e < end && *e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (const char* q = p; q < e; ++q)
    if (*q != '.') digits[nd++] = *q;
  int exp10 = 0;
  {
    const char* q = e + 1;
    const bool neg = *q == '-';
    if (*q == '+' || *q == '-') ++q;
    for (; q < end; ++q) exp10 = exp10 * 10 + (*q - '0');
    if (neg) exp10 = -exp10;
  }
  if (exp10 >= -4 && exp10 < 16) {  // fixed notation
    if (exp10 >= 0) {
      for (int i = 0; i <= exp10; ++i) out += i < nd ? digits[i] : '0';
      out += '.';
      if (nd > exp10 + 1) out.append(digits + exp10 + 1, static_cast<size_t>(nd - exp10 - 1));
      else out += '0';
    } else {
      out += "0.";
      out.append(static_cast<size_t>(-exp10 - 1), '0');
      out.append(digits, static_cast<size_t>(nd));
    }
  } else {  // to_chars' scientific form is Python's: d[.ddd]e+XX with at least two exponent digits
    out.append(p, static_cast<size_t>(end - p));
  }
}

void append_int(std::string& out, long long v) {
  char buf[24];
  const auto res = std::to_chars(buf, buf + sizeof(buf), v);
  out.append(buf, static_cast<size_t>(res.ptr - buf));
}

// Rows of different windows are independent: above a few thousand windows the range is cut into pieces
// that persistent workers format into strings of their own, joined in order afterwards (10^4 windows of
// C3 with both logs: 1.5 ms on one core, the largest host item of the product path after the GPU pass).
// `rows(w0, w1, out)` appends the rows of windows [w0, w1) and returns a status.
constexpr int kTextThreads = 8;
// A worker is worth waking for a few hundred rows (~0.3 us per row and log): with 1 024 a window range of a
// region written in parts (FeaturePreprocessor.score_and_write: 2 300-3 800 windows per call) ran on two or three
// of the eight workers
constexpr int32_t kRowsPerPiece = 256;

ProcessPool& text_pool() {
  static ProcessPool p(kTextThreads);
  return p;
}

template <typename Rows>
int format_rows(int32_t n_windows, std::string& out, Rows&& rows) {
  const int nt = static_cast<int>(std::min<int64_t>(kTextThreads, n_windows / kRowsPerPiece));
  if (nt <= 1 || !text_pool().usable()) return rows(0, n_windows, out);  // (a forked child: no worker threads)
  std::string part[kTextThreads];
  int rc[kTextThreads] = {0};
  bool threw[kTextThreads] = {false};
  {
    text_pool().run(nt, [&](int t) {
      const int32_t w0 = static_cast<int32_t>(static_cast<int64_t>(n_windows) * t / nt);
      const int32_t w1 = static_cast<int32_t>(static_cast<int64_t>(n_windows) * (t + 1) / nt);
      try {
        rc[t] = rows(w0, w1, part[t]);
      } catch (...) {  // a worker must not throw: the caller reports it as out of memory
        threw[t] = true;
      }
    });
  }
  size_t total = out.size();
  for (int t = 0; t < nt; ++t) {
    if (threw[t]) throw std::bad_alloc();
    if (rc[t]) return rc[t];  // the piece's own sai_set_error message went to its thread: repeat the class here
    total += part[t].size();
  }
  out.reserve(total);
  for (int t = 0; t < nt; ++t) out += part[t];
  return SAI_OK;
}

// The TSV rows of windows [w0, w1) (feature_preprocessor.py:232-236).
struct ScoreRows {
  const std::string& chr;
  const std::string& pops;
  const int64_t* windows;
  const int32_t* nsnps;
  int32_t n_cols;
  const sai_text_column* cols;
  void append(int32_t w_begin, int32_t w_end, std::string& out) const {
    out.reserve(out.size() + static_cast<size_t>(w_end - w_begin) * (48 + 12 * static_cast<size_t>(n_cols)));
    for (int32_t w = w_begin; w < w_end; ++w) {
      out += chr;
      out += '\t';
      append_int(out, windows[2 * w]);
      out += '\t';
      append_int(out, windows[2 * w + 1]);
      out += '\t';
      out += pops;
      out += '\t';
      append_int(out, nsnps[w]);
      const bool empty = nsnps[w] == 0;  // no site in the window: every statistic prints nan
      for (int32_t c = 0; c < n_cols; ++c) {
        out += '\t';
        if (empty) { out += "nan"; continue; }
        const char* p = static_cast<const char*>(cols[c].data) + static_cast<int64_t>(w) * cols[c].stride_bytes;
        if (cols[c].kind == SAI_TEXT_I32) {
          int32_t v;
          std::memcpy(&v, p, sizeof(v));
          append_int(out, v);
        } else {
          double v;
          std::memcpy(&v, p, sizeof(v));
          append_double(out, v);
        }
      }
      out += '\n';
    }
  }
};

// The .U.log / .Q.log rows of windows [w0, w1) (feature_preprocessor.py:241-258).
struct LogRows {
  const std::string& chr;
  const int64_t* windows;
  const void* counts;
  int64_t count_stride_bytes;
  const int64_t* offsets;
  int64_t offset_stride_words;
  const void* positions;
  int32_t position_bytes;
  void append(int32_t w_begin, int32_t w_end, std::string& out) const {
    for (int32_t w = w_begin; w < w_end; ++w) {
      out += chr;
      out += '\t';
      append_int(out, windows[2 * w]);
      out += '\t';
      append_int(out, windows[2 * w + 1]);
      out += '\t';
      int32_t n;
      std::memcpy(&n, static_cast<const char*>(counts) + static_cast<int64_t>(w) * count_stride_bytes, sizeof(n));
      if (n <= 0) {
        out += "NA\n";
        continue;
      }
      const int64_t o = offsets[static_cast<int64_t>(w) * offset_stride_words];
      for (int32_t k = 0; k < n; ++k) {
        if (k) out += ',';
        out += chr;
        out += ':';
        long long p;
        if (position_bytes == 4) {
          int32_t v;
          std::memcpy(&v, static_cast<const char*>(positions) + (o + k) * 4, 4);
          p = v;
        } else {
          int64_t v;
          std::memcpy(&v, static_cast<const char*>(positions) + (o + k) * 8, 8);
          p = v;
        }
        append_int(out, p);
      }
      out += '\n';
    }
  }
};

// A list may be NULL only when no window has an entry (checked up front: the pieces run on worker
// threads, whose error text is their own).
bool lists_present(int32_t n_windows, const void* counts, int64_t count_stride_bytes, const void* positions) {
  if (positions) return true;
  for (int32_t w = 0; w < n_windows; ++w) {
    int32_t n;
    std::memcpy(&n, static_cast<const char*>(counts) + static_cast<int64_t>(w) * count_stride_bytes, sizeof(n));
    if (n > 0) return false;
  }
  return true;
}

// writev() of the pieces of one output, in order, until everything is out.
int write_pieces(int fd, const std::string* const* piece, int n_pieces, int64_t* bytes) {
  iovec iov[kTextThreads];
  int n = 0;
  for (int t = 0; t < n_pieces; ++t) {
    if (piece[t]->empty()) continue;
    iov[n].iov_base = const_cast<char*>(piece[t]->data());
    iov[n].iov_len = piece[t]->size();
    *bytes += static_cast<int64_t>(piece[t]->size());
    ++n;
  }
  int at = 0;
  while (at < n) {
    const ssize_t got = ::writev(fd, iov + at, n - at);
    if (got < 0) {
      if (errno == EINTR) continue;
      return sai_set_error(SAI_ERR_ARG, "sai_write_window_rows: write to fd %d failed: %s", fd, std::strerror(errno));
    }
    size_t left = static_cast<size_t>(got);
    while (at < n && left >= iov[at].iov_len) left -= iov[at++].iov_len;
    if (at < n && left) {
      iov[at].iov_base = static_cast<char*>(iov[at].iov_base) + left;
      iov[at].iov_len -= left;
    }
  }
  return SAI_OK;
}

// What sai_write_window_rows keeps between calls: the pieces' strings (a later call appends into memory
// that is already mapped -- a fresh MB costs more in page faults than the formatting of its text) and the
// mutex that makes concurrent callers take turns.  A fork() while another thread is inside would hand the
// child a mutex that stays locked for ever: the child starts with a FRESH state (the old one is abandoned,
// not destroyed -- its mutex may be held, its strings half written).
struct WriterState {
  std::mutex busy;
  std::string piece[kTextThreads][1 + SAI_MAX_LOGS];
};
WriterState* g_writer = nullptr;
void writer_reset_in_child() { g_writer = new (std::nothrow) WriterState; }
WriterState* writer_state() {
  static std::once_flag once;
  std::call_once(once, [] {
    g_writer = new WriterState;
    pthread_atfork(nullptr, nullptr, writer_reset_in_child);
  });
  return g_writer;
}
constexpr size_t kKeepPieceBytes = size_t{1} << 20;  // larger pieces (a whole-genome call) are given back

// size of a regular file behind fd, or -1 (a pipe, a terminal: nothing to roll back to)
int64_t regular_file_size(int fd) {
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return -1;
  return static_cast<int64_t>(sb.st_size);
}

template <typename F>
int guarded_text(const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return sai_set_error(SAI_ERR_HIP, "%s: out of host memory", what);
  } catch (const std::exception& e) {
    return sai_set_error(SAI_ERR_HIP, "%s: %s", what, e.what());
  } catch (...) {
    return sai_set_error(SAI_ERR_HIP, "%s: unknown failure", what);
  }
}

}  // namespace

struct sai_text {
  std::string s;
};

extern "C" {

int sai_format_score_rows(const char* chr_name_host, const char* pop_columns_host, int32_t n_windows,
                          const int64_t* windows_host, const int32_t* nsnps_host, int32_t n_cols,
                          const sai_text_column* cols_host, sai_text** text_out) {
  return guarded_text("sai_format_score_rows", [&]() -> int {
    if (!chr_name_host || !pop_columns_host || !text_out) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    *text_out = nullptr;
    if (n_windows < 0 || n_cols < 0) return sai_set_error(SAI_ERR_ARG, "negative size");
    if (n_windows > 0 && (!windows_host || !nsnps_host)) return sai_set_error(SAI_ERR_ARG, "NULL buffer");
    for (int32_t c = 0; c < n_cols; ++c) {
      if (!cols_host || (cols_host[c].kind != SAI_TEXT_I32 && cols_host[c].kind != SAI_TEXT_F64))
        return sai_set_error(SAI_ERR_ARG, "column %d: bad kind", c);
      if (n_windows > 0 && !cols_host[c].data) return sai_set_error(SAI_ERR_ARG, "column %d: NULL data", c);
    }
    std::unique_ptr<sai_text> t(new sai_text);  // freed if an append below throws (guarded_text turns that into a status)
    const std::string chr(chr_name_host), pops(pop_columns_host);
    const ScoreRows rows{chr, pops, windows_host, nsnps_host, n_cols, cols_host};
    const int frc = format_rows(n_windows, t->s, [&](int32_t w_begin, int32_t w_end, std::string& out) -> int {
      rows.append(w_begin, w_end, out);
      return SAI_OK;
    });
    if (frc) return frc;
    *text_out = t.release();
    return SAI_OK;
  });
}

int sai_format_log_rows(const char* chr_name_host, int32_t n_windows, const int64_t* windows_host,
                        const void* counts_host, int64_t count_stride_bytes, const int64_t* offsets_host,
                        int64_t offset_stride_words, const void* positions_host, int32_t position_bytes,
                        sai_text** text_out) {
  return guarded_text("sai_format_log_rows", [&]() -> int {
    if (!chr_name_host || !text_out) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    *text_out = nullptr;
    if (n_windows < 0) return sai_set_error(SAI_ERR_ARG, "negative size");
    if (position_bytes != 4 && position_bytes != 8) return sai_set_error(SAI_ERR_ARG, "positions must be int32 or int64");
    if (n_windows > 0 && (!windows_host || !counts_host || !offsets_host)) return sai_set_error(SAI_ERR_ARG, "NULL buffer");
    std::unique_ptr<sai_text> t(new sai_text);  // freed if an append below throws (guarded_text turns that into a status)
    const std::string chr(chr_name_host);
    if (!lists_present(n_windows, counts_host, count_stride_bytes, positions_host)) return sai_set_error(SAI_ERR_ARG, "NULL candidate list");
    const LogRows rows{chr, windows_host, counts_host, count_stride_bytes, offsets_host, offset_stride_words, positions_host,
                       position_bytes};
    const int frc = format_rows(n_windows, t->s, [&](int32_t w_begin, int32_t w_end, std::string& out) -> int {
      rows.append(w_begin, w_end, out);
      return SAI_OK;
    });
    if (frc) return frc;
    *text_out = t.release();
    return SAI_OK;
  });
}

int sai_write_window_rows(const char* chr_name_host, const char* pop_columns_host, int32_t n_windows,
                          const int64_t* windows_host, const int32_t* nsnps_host, int32_t n_cols,
                          const sai_text_column* cols_host, int32_t tsv_fd, int32_t n_logs,
                          const sai_log_rows* logs_host, int64_t* bytes_out) {
  return guarded_text("sai_write_window_rows", [&]() -> int {
    if (!chr_name_host || !pop_columns_host) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    if (n_windows < 0 || n_cols < 0 || n_logs < 0 || n_logs > SAI_MAX_LOGS) return sai_set_error(SAI_ERR_ARG, "bad size");
    if (n_logs > 0 && !logs_host) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    if (n_windows > 0 && (!windows_host || (tsv_fd >= 0 && !nsnps_host))) return sai_set_error(SAI_ERR_ARG, "NULL buffer");
    if (tsv_fd >= 0)
      for (int32_t c = 0; c < n_cols; ++c) {
        if (!cols_host || (cols_host[c].kind != SAI_TEXT_I32 && cols_host[c].kind != SAI_TEXT_F64))
          return sai_set_error(SAI_ERR_ARG, "column %d: bad kind", c);
        if (n_windows > 0 && !cols_host[c].data) return sai_set_error(SAI_ERR_ARG, "column %d: NULL data", c);
      }
    for (int32_t k = 0; k < n_logs; ++k) {
      const sai_log_rows& lg = logs_host[k];
      if (lg.fd < 0) continue;
      if (lg.position_bytes != 4 && lg.position_bytes != 8) return sai_set_error(SAI_ERR_ARG, "positions must be int32 or int64");
      if (n_windows > 0 && (!lg.counts_host || !lg.offsets_host)) return sai_set_error(SAI_ERR_ARG, "NULL buffer");
      if (!lists_present(n_windows, lg.counts_host, lg.count_stride_bytes, lg.positions_host))
        return sai_set_error(SAI_ERR_ARG, "NULL candidate list");
    }
    for (int32_t k = 0; bytes_out && k <= n_logs; ++k) bytes_out[k] = 0;
    const std::string chr(chr_name_host), pops(pop_columns_host);
    const ScoreRows score{chr, pops, windows_host, nsnps_host, n_cols, cols_host};
    WriterState* ws = writer_state();
    if (!ws) return sai_set_error(SAI_ERR_HIP, "sai_write_window_rows: out of host memory");
    std::lock_guard<std::mutex> lk(ws->busy);
    auto& piece = ws->piece;
    int nt = static_cast<int>(std::min<int64_t>(kTextThreads, n_windows / kRowsPerPiece));
    if (nt < 1 || !text_pool().usable()) nt = 1;  // (a forked child: no worker threads)
    bool threw[kTextThreads] = {false};
    const auto job = [&](int t) {
      const int32_t w0 = static_cast<int32_t>(static_cast<int64_t>(n_windows) * t / nt);
      const int32_t w1 = static_cast<int32_t>(static_cast<int64_t>(n_windows) * (t + 1) / nt);
      try {
        for (int32_t k = 0; k <= n_logs; ++k) piece[t][k].clear();
        if (tsv_fd >= 0) score.append(w0, w1, piece[t][0]);
        for (int32_t k = 0; k < n_logs; ++k) {
          const sai_log_rows& lg = logs_host[k];
          if (lg.fd < 0) continue;
          const LogRows rows{chr, windows_host, lg.counts_host, lg.count_stride_bytes, lg.offsets_host, lg.offset_stride_words,
                             lg.positions_host, lg.position_bytes};
          rows.append(w0, w1, piece[t][1 + k]);
        }
      } catch (...) {  // a worker must not throw: reported as out of memory below
        threw[t] = true;
      }
    };
    if (nt > 1) text_pool().run(nt, job);
    else job(0);
    int rc = SAI_OK;
    for (int t = 0; t < nt; ++t)
      if (threw[t]) rc = sai_set_error(SAI_ERR_HIP, "sai_write_window_rows: out of host memory");
    // The three files of a run must agree on their last window: where each output ended before this call is
    // remembered, and a failed or short write takes every output back there (regular files; a pipe keeps
    // what it got).
    int64_t before[1 + SAI_MAX_LOGS];
    for (int32_t k = 0; k <= n_logs; ++k) {
      const int fd = k == 0 ? tsv_fd : logs_host[k - 1].fd;
      before[k] = fd >= 0 ? regular_file_size(fd) : -1;
    }
    for (int32_t k = 0; rc == SAI_OK && k <= n_logs; ++k) {
      const int fd = k == 0 ? tsv_fd : logs_host[k - 1].fd;
      if (fd < 0) continue;
      const std::string* of[kTextThreads];
      for (int t = 0; t < nt; ++t) of[t] = &piece[t][k];
      int64_t n = 0;
      rc = write_pieces(fd, of, nt, &n);
      if (bytes_out) bytes_out[k] = n;
    }
    if (rc != SAI_OK) {
      for (int32_t k = 0; k <= n_logs; ++k) {
        const int fd = k == 0 ? tsv_fd : logs_host[k - 1].fd;
        if (fd >= 0 && before[k] >= 0 && regular_file_size(fd) > before[k] && ftruncate(fd, static_cast<off_t>(before[k])) == 0)
          lseek(fd, 0, SEEK_END);
        if (bytes_out) bytes_out[k] = 0;
      }
    }
    for (int t = 0; t < nt; ++t)
      for (int32_t k = 0; k <= n_logs; ++k)
        if (piece[t][k].capacity() > kKeepPieceBytes) std::string().swap(piece[t][k]);
    return rc;
  });
}

int sai_format_doubles(const double* values_host, int64_t n, sai_text** text_out) {
  return guarded_text("sai_format_doubles", [&]() -> int {
    if (!text_out) return sai_set_error(SAI_ERR_ARG, "bad argument");
    *text_out = nullptr;
    if ((n > 0 && !values_host) || n < 0) return sai_set_error(SAI_ERR_ARG, "bad argument");
    std::unique_ptr<sai_text> t(new sai_text);  // freed if an append below throws (guarded_text turns that into a status)
    for (int64_t i = 0; i < n; ++i) {
      append_double(t->s, values_host[i]);
      t->s += '\n';
    }
    *text_out = t.release();
    return SAI_OK;
  });
}

const char* sai_text_data(const sai_text* text, int64_t* n_bytes) {
  if (!text) {
    if (n_bytes) *n_bytes = 0;
    return "";
  }
  if (n_bytes) *n_bytes = static_cast<int64_t>(text->s.size());
  return text->s.data();
}

int sai_text_free(sai_text* text) {
  delete text;
  return SAI_OK;
}

}  // extern "C"
