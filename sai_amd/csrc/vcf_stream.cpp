// sai_vcf_stream_*: the host reads / inflates a VCF and indexes its record lines while the genotype
// text crosses PCIe untouched for the GPU tokenizer (split out of vcf_ingest.cpp in round 3).

#include "ingest_index.hpp"

struct sai_vcf_stream {
  std::string path, chrom, anc_path;
  int64_t start = -1, end = -1;
  int n_threads = 1;
  std::vector<std::string> names;
  std::vector<int32_t> ploidy;
  char* bufs[2] = {nullptr, nullptr};
  size_t cap = 0;
  // producer state
  Selection sel;
  AncMap anc;
  bool header_seen = false;
  int64_t n_matched = 0, n_anc = 0;
  // hand-over: batch k lives in buffer k % 2
  std::mutex m;
  std::condition_variable cv;
  IndexOut batch[2];
  size_t batch_bytes[2] = {0, 0};
  int state[2] = {0, 0};  // 0 free, 1 full, 2 held by the consumer
  int64_t produced = 0, consumed = 0;
  int held = -1;
  bool finished = false, cancel = false;
  int rc = 0;
  std::string err;
  std::unique_ptr<WorkerPool> pool;  // the producer's indexing / copying workers
  std::thread producer;
};

namespace {

// Copy + index [p, endp) (whole lines) into the next free buffer(s).  Returns 0, or a negative status.
int stream_emit(sai_vcf_stream* st, const char* p, const char* endp, bool* done, bool* seen_chrom) {
  const int nt = std::max(1, st->n_threads);
  if (!st->pool) st->pool.reset(new WorkerPool(nt));
  std::vector<IndexOut> outs(static_cast<size_t>(nt));
  while (p < endp) {
    // the part of [p, endp) that fits a buffer, cut at a line boundary
    const char* cut = endp;
    if (static_cast<size_t>(endp - p) > st->cap) {
      cut = p + st->cap;
      while (cut > p && cut[-1] != '\n') --cut;
      if (cut == p) return sai_set_error(SAI_ERR_UNSUPPORTED, "%s: a line is longer than the staging buffer", st->path.c_str());
    }
    int b;
    {
      std::unique_lock<std::mutex> lk(st->m);
      b = static_cast<int>(st->produced % 2);
      st->cv.wait(lk, [&] { return st->state[b] == 0 || st->cancel; });
      if (st->cancel) return 1;
    }
    char* dst = st->bufs[b];
    const size_t total = static_cast<size_t>(cut - p);
    std::vector<const char*> edge(static_cast<size_t>(nt) + 1, cut);
    edge[0] = p;
    for (int t = 1; t < nt; ++t) {
      const char* guess = p + total * static_cast<size_t>(t) / static_cast<size_t>(nt);
      if (guess < edge[static_cast<size_t>(t) - 1]) guess = edge[static_cast<size_t>(t) - 1];
      const char* nl = static_cast<const char*>(memchr(guess, '\n', static_cast<size_t>(cut - guess)));
      edge[static_cast<size_t>(t)] = nl ? nl + 1 : cut;
    }
    for (auto& o : outs) o.clear();
    auto piece = [&](int t) {
      IndexOut& o = outs[static_cast<size_t>(t)];
      const char* a = edge[static_cast<size_t>(t)];
      const char* z = edge[static_cast<size_t>(t) + 1];
      if (a >= z) return;
      try {
        memcpy(dst + (a - p), a, static_cast<size_t>(z - a));  // the text goes to the pinned buffer as it is
        index_lines(a, z, p, st->chrom, st->start, st->end, st->anc, o);
      } catch (...) {
        o.failed = true;
      }
    };
    st->pool->run(nt, piece);
    IndexOut& out = st->batch[b];
    out.clear();
    for (auto& o : outs) {
      if (o.failed) return sai_set_error(SAI_ERR_HIP, "%s: indexing failed (out of memory)", st->path.c_str());
      if (!o.error.empty()) return sai_set_error(SAI_ERR_ARG, "%s: %s", st->path.c_str(), o.error.c_str());
      st->n_matched += o.matched;
      out.off.insert(out.off.end(), o.off.begin(), o.off.end());
      out.len.insert(out.len.end(), o.len.begin(), o.len.end());
      out.pos.insert(out.pos.end(), o.pos.begin(), o.pos.end());
      out.flip.insert(out.flip.end(), o.flip.begin(), o.flip.end());
      out.gi.insert(out.gi.end(), o.gi.begin(), o.gi.end());
      *seen_chrom = *seen_chrom || o.saw_chrom;
      if (o.beyond_stop || (*seen_chrom && o.last_line_other)) *done = true;
    }
    {
      std::lock_guard<std::mutex> lk(st->m);
      st->batch_bytes[b] = total;
      st->state[b] = 1;
      ++st->produced;
    }
    st->cv.notify_all();
    p = cut;
  }
  return 0;
}

// Uncompressed text: the batches are pread straight into the staging buffers (no intermediate copy)
// and indexed where they lie.
int stream_run_plain(sai_vcf_stream* st, const std::vector<const char*>& names) {
  const char* path = st->path.c_str();
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
  struct FdGuard { int fd; ~FdGuard() { close(fd); } } guard{fd};
  struct stat sb;
  if (fstat(fd, &sb) != 0) return sai_set_error(SAI_ERR_ARG, "cannot stat %s", path);
  const size_t total = static_cast<size_t>(sb.st_size);
  // header: read from the top until the #CHROM line has been seen
  size_t data_off = 0;
  {
    std::vector<char> head;
    size_t have = 0;
    while (!st->header_seen) {
      const size_t want = std::min(total - have, std::max<size_t>(size_t(1) << 20, have));
      if (want == 0) break;
      head.resize(have + want);
      size_t got_all = 0;
      while (got_all < want) {
        const ssize_t got = pread(fd, head.data() + have + got_all, want - got_all, static_cast<off_t>(have + got_all));
        if (got <= 0) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
        got_all += static_cast<size_t>(got);
      }
      have += want;
      const char* p = head.data() + data_off;
      const char* endp = head.data() + have;
      while (p < endp) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(endp - p)));
        if (!eol) {
          if (have < total) break;  // an incomplete line: read more
          eol = endp;
        }
        if (*p != '#') return sai_set_error(SAI_ERR_ARG, "%s: no #CHROM header line before the records", path);
        const bool is_chrom = eol - p > 6 && memcmp(p, "#CHROM", 6) == 0;
        if (is_chrom) {
          if (int hrc = parse_header(p, eol, path, static_cast<int32_t>(names.size()), names.data(), st->ploidy.data(), st->sel))
            return hrc;
        }
        p = eol < endp ? eol + 1 : endp;
        data_off = static_cast<size_t>(p - head.data());
        if (is_chrom) {
          std::lock_guard<std::mutex> lk(st->m);
          st->header_seen = true;
          break;
        }
      }
      if (have >= total) break;
    }
  }
  if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
  const int nt = std::max(1, st->n_threads);
  if (!st->pool) st->pool.reset(new WorkerPool(nt));
  std::vector<IndexOut> outs(static_cast<size_t>(nt));
  bool done = false, seen_chrom = false;
  size_t file_off = data_off;
  while (file_off < total && !done) {
    int b;
    {
      std::unique_lock<std::mutex> lk(st->m);
      b = static_cast<int>(st->produced % 2);
      st->cv.wait(lk, [&] { return st->state[b] == 0 || st->cancel; });
      if (st->cancel) return SAI_OK;
    }
    char* dst = st->bufs[b];
    const size_t want = std::min(st->cap, total - file_off);
    std::vector<char> bad(static_cast<size_t>(nt), 0);
    auto reader = [&](int t) {
      size_t lo = want * static_cast<size_t>(t) / static_cast<size_t>(nt), hi = want * static_cast<size_t>(t + 1) / static_cast<size_t>(nt);
      while (lo < hi) {
        const ssize_t got = pread(fd, dst + lo, hi - lo, static_cast<off_t>(file_off + lo));
        if (got <= 0) { bad[static_cast<size_t>(t)] = 1; return; }
        lo += static_cast<size_t>(got);
      }
    };
    st->pool->run(nt, reader);
    for (char x : bad)
      if (x) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
    size_t usable = want;
    if (file_off + want < total) {  // cut at the last complete line; the rest is read again next time
      while (usable > 0 && dst[usable - 1] != '\n') --usable;
      if (usable == 0) return sai_set_error(SAI_ERR_UNSUPPORTED, "%s: a line is longer than the staging buffer", path);
    }
    const char* p = dst;
    const char* cut = dst + usable;
    std::vector<const char*> edge(static_cast<size_t>(nt) + 1, cut);
    edge[0] = p;
    for (int t = 1; t < nt; ++t) {
      const char* guess = p + usable * static_cast<size_t>(t) / static_cast<size_t>(nt);
      if (guess < edge[static_cast<size_t>(t) - 1]) guess = edge[static_cast<size_t>(t) - 1];
      const char* nl = static_cast<const char*>(memchr(guess, '\n', static_cast<size_t>(cut - guess)));
      edge[static_cast<size_t>(t)] = nl ? nl + 1 : cut;
    }
    for (auto& o : outs) o.clear();
    auto piece = [&](int t) {
      IndexOut& o = outs[static_cast<size_t>(t)];
      if (edge[static_cast<size_t>(t)] >= edge[static_cast<size_t>(t) + 1]) return;
      try {
        index_lines(edge[static_cast<size_t>(t)], edge[static_cast<size_t>(t) + 1], p, st->chrom, st->start, st->end, st->anc, o);
      } catch (...) {
        o.failed = true;
      }
    };
    st->pool->run(nt, piece);
    IndexOut& out = st->batch[b];
    out.clear();
    for (auto& o : outs) {
      if (o.failed) return sai_set_error(SAI_ERR_HIP, "%s: indexing failed (out of memory)", path);
      if (!o.error.empty()) return sai_set_error(SAI_ERR_ARG, "%s: %s", path, o.error.c_str());
      st->n_matched += o.matched;
      out.off.insert(out.off.end(), o.off.begin(), o.off.end());
      out.len.insert(out.len.end(), o.len.begin(), o.len.end());
      out.pos.insert(out.pos.end(), o.pos.begin(), o.pos.end());
      out.flip.insert(out.flip.end(), o.flip.begin(), o.flip.end());
      out.gi.insert(out.gi.end(), o.gi.begin(), o.gi.end());
      seen_chrom = seen_chrom || o.saw_chrom;
      if (o.beyond_stop || (seen_chrom && o.last_line_other)) done = true;
    }
    {
      std::lock_guard<std::mutex> lk(st->m);
      st->batch_bytes[b] = usable;
      st->state[b] = 1;
      ++st->produced;
    }
    st->cv.notify_all();
    file_off += usable;
  }
  return SAI_OK;
}

int stream_run(sai_vcf_stream* st) {
  const char* path = st->path.c_str();
  if (!st->anc_path.empty()) {
    if (int rc = load_anc(st->anc_path.c_str(), st->chrom, st->start, st->end, st->anc, &st->n_anc)) return rc;
  }
  std::vector<const char*> names;
  for (auto& n : st->names) names.push_back(n.c_str());
  if (file_is_plain_text(path)) return stream_run_plain(st, names);
  bool done = false, seen_chrom = false;
  auto on_header = [&](const char*& p, const char* endp) -> int {
    while (!st->header_seen && p < endp) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(endp - p)));
      if (!eol) eol = endp;
      if (*p != '#') return sai_set_error(SAI_ERR_ARG, "%s: no #CHROM header line before the records", path);
      if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) {
        if (int hrc = parse_header(p, eol, path, static_cast<int32_t>(names.size()), names.data(), st->ploidy.data(), st->sel))
          return hrc;
        std::lock_guard<std::mutex> lk(st->m);
        st->header_seen = true;
      }
      p = eol + 1;
    }
    return 0;
  };
  auto on_records = [&](const char* p, const char* endp) -> int {
    if (p >= endp) return 0;
    const int rc = stream_emit(st, p, endp, &done, &seen_chrom);
    if (rc < 0) return rc;
    return (rc > 0 || done) ? 1 : 0;
  };
  int rc;
  TbiRef idx;
  if (st->start >= 0 && file_is_bgzf(path) && load_tbi(path, st->chrom, idx)) {
    rc = for_each_block_from(path, 1, 0, size_t(1) << 16, [&](const char* p, const char* endp) -> int {
      if (int hrc = on_header(p, endp)) return hrc;
      return (st->header_seen || p < endp) ? 1 : 0;
    });
    const uint64_t window = static_cast<uint64_t>(st->start > 0 ? st->start - 1 : 0) >> 14;
    if (rc == SAI_OK && st->header_seen && idx.present && window < idx.ioff.size())
      rc = for_each_block_from(path, st->n_threads, idx.ioff[window], 0, on_records);
  } else {
    rc = for_each_block(path, st->n_threads, [&](const char* p, const char* endp) -> int {
      if (int hrc = on_header(p, endp)) return hrc;
      return on_records(p, endp);
    });
  }
  if (rc) return rc;
  if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
  return SAI_OK;
}

void stream_producer(sai_vcf_stream* st) {
  int rc;
  std::string err;
  try {
    rc = stream_run(st);
    if (rc) err = sai_last_error();  // the producer thread's own message
  } catch (const std::bad_alloc&) {
    rc = SAI_ERR_HIP;
    err = "sai_vcf_stream: out of host memory";
  } catch (const std::exception& e) {
    rc = SAI_ERR_HIP;
    err = std::string("sai_vcf_stream: ") + e.what();
  } catch (...) {
    rc = SAI_ERR_HIP;
    err = "sai_vcf_stream: unknown failure";
  }
  {
    std::lock_guard<std::mutex> lk(st->m);
    st->rc = rc;
    st->err = err;
    st->finished = true;
  }
  st->cv.notify_all();
}

}  // namespace

extern "C" {

int sai_vcf_stream_open(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                        const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                        int32_t n_threads, void* pinned0_host, void* pinned1_host, int64_t buffer_bytes,
                        sai_vcf_stream** stream_out) {
  return guarded("sai_vcf_stream_open", [&]() -> int {
    if (!path || !chrom || !stream_out) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    *stream_out = nullptr;
    if (n_samples < 1 || !sample_names || !ploidy) return sai_set_error(SAI_ERR_ARG, "empty sample selection");
    if (!pinned0_host || !pinned1_host || buffer_bytes < (1 << 16)) return sai_set_error(SAI_ERR_ARG, "two staging buffers of at least 64 KiB are needed");
    for (int32_t s = 0; s < n_samples; ++s)
      if (ploidy[s] < 1 || ploidy[s] > 64) return sai_set_error(SAI_ERR_ARG, "ploidy of sample %d out of range", s);
    {
      FILE* f = fopen(path, "rb");
      if (!f) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
      fclose(f);
    }
    std::unique_ptr<sai_vcf_stream> st(new sai_vcf_stream);
    st->path = path;
    st->chrom = chrom;
    st->start = start;
    st->end = end;
    st->n_threads = n_threads < 1 ? 1 : n_threads;
    if (anc_bed_path) st->anc_path = anc_bed_path;
    for (int32_t s = 0; s < n_samples; ++s) {
      st->names.emplace_back(sample_names[s]);
      st->ploidy.push_back(ploidy[s]);
    }
    st->bufs[0] = static_cast<char*>(pinned0_host);
    st->bufs[1] = static_cast<char*>(pinned1_host);
    st->cap = static_cast<size_t>(buffer_bytes);
    sai_vcf_stream* raw = st.get();
    st->producer = std::thread(stream_producer, raw);
    *stream_out = st.release();
    return SAI_OK;
  });
}

int sai_vcf_stream_next(sai_vcf_stream* st, int32_t* buffer_index, int64_t* n_text_bytes, int64_t* n_lines,
                        const int64_t** line_off_host, const int32_t** line_len_host, const int32_t** line_pos_host,
                        const uint8_t** line_flip_host, const uint8_t** line_gi_host, int32_t* done) {
  if (!st || !buffer_index || !n_text_bytes || !n_lines || !line_off_host || !line_len_host || !line_pos_host ||
      !line_flip_host || !line_gi_host || !done)
    return sai_set_error(SAI_ERR_ARG, "NULL argument");
  std::unique_lock<std::mutex> lk(st->m);
  if (st->held >= 0) {  // the caller is done with the batch it got last time
    st->state[st->held] = 0;
    st->held = -1;
    st->cv.notify_all();
  }
  const int b = static_cast<int>(st->consumed % 2);
  st->cv.wait(lk, [&] { return st->state[b] == 1 || st->finished; });
  if (st->state[b] != 1) {  // nothing more will come
    *done = 1;
    *n_lines = *n_text_bytes = 0;
    *buffer_index = -1;
    if (st->rc) return sai_set_error(st->rc, "%s", st->err.c_str());
    return SAI_OK;
  }
  const IndexOut& o = st->batch[b];
  st->state[b] = 2;
  st->held = b;
  ++st->consumed;
  *done = 0;
  *buffer_index = b;
  *n_text_bytes = static_cast<int64_t>(st->batch_bytes[b]);
  *n_lines = static_cast<int64_t>(o.off.size());
  *line_off_host = o.off.data();
  *line_len_host = o.len.data();
  *line_pos_host = o.pos.data();
  *line_flip_host = o.flip.data();
  *line_gi_host = o.gi.data();
  return SAI_OK;
}

int sai_vcf_stream_selection(sai_vcf_stream* st, int32_t* slot_of_col_host, int32_t capacity, int32_t* n_cols,
                             int64_t* n_matched, int64_t* n_anc_entries) {
  if (!st || !n_cols) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(st->m);
  if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "the header has not been read yet");
  *n_cols = st->sel.max_col + 1;
  if (slot_of_col_host) {
    if (capacity < *n_cols) return sai_set_error(SAI_ERR_ARG, "slot_of_col capacity %d < %d", capacity, *n_cols);
    for (int32_t c = 0; c < *n_cols; ++c) slot_of_col_host[c] = st->sel.slot_of_col[static_cast<size_t>(c)];
  }
  if (n_matched) *n_matched = st->n_matched;  // complete once sai_vcf_stream_next has reported done
  if (n_anc_entries) *n_anc_entries = st->n_anc;
  return SAI_OK;
}

int sai_vcf_stream_close(sai_vcf_stream* st) {
  if (!st) return SAI_OK;
  {
    std::lock_guard<std::mutex> lk(st->m);
    st->cancel = true;
  }
  st->cv.notify_all();
  if (st->producer.joinable()) st->producer.join();
  delete st;
  return SAI_OK;
}

}  // extern "C"
