// sai_bgzf_stream_* + sai_vcf_index_text / _heads: whole BGZF members for the GPU inflate, the tabix
// region seek, and the record index from the text or from the line heads the GPU extracts (split out of
// vcf_ingest.cpp in round 3).

#include "ingest_index.hpp"

// ------------------------------------------------------------------------------------------
// bgzip input for the GPU inflate (sai_bgzf_stream_*, sai_vcf_index_text): a reader thread hands
// the file's BGZF members over AS THEY ARE -- whole members, with the table sai_inflate_bgzf needs
// -- in the caller's two pinned buffers alternately; the caller inflates them on the GPU, copies
// the text back once, and sai_vcf_index_text checks the members' CRCs and indexes the record lines
// of that text (header, chromosome / region filter, POS, the ancestral-allele decision, the GT
// index: index_lines above, unchanged).  The text never leaves HBM for the tokenizer.
// ------------------------------------------------------------------------------------------
struct sai_bgzf_stream {
  std::string path, chrom;
  int64_t start = -1, end = -1;
  int n_threads = 1;
  std::vector<std::string> names;
  std::vector<int32_t> ploidy;
  // reader
  unsigned char* bufs[2] = {nullptr, nullptr};
  size_t cap = 0, text_cap = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<sai_bgzf_member> members[2];
  size_t comp_bytes[2] = {0, 0}, text_bytes[2] = {0, 0};
  int state[2] = {0, 0};  // 0 free, 1 full, 2 held by the consumer
  int64_t produced = 0, consumed = 0;
  int held = -1;
  bool finished = false, cancel = false;
  int rc = 0;
  std::string err;
  std::thread reader;
  // region seek through <vcf>.tbi: the reader starts at the member of the region's first record and
  // slows down at the member that holds the first record of a later window (-1: no bound)
  int64_t file_begin = 0, file_stop = -1;
  // the record index has seen a record beyond the region (or the next chromosome): set by the consumer,
  // under `m`; past `file_stop` the reader goes on in small batches until then
  bool consumer_done = false;
  int64_t first_text_skip = 0;  // text of the first member that precedes the region's first record
  bool nothing_to_read = false; // the index says the region holds no record
  // indexer (the consumer's thread)
  Selection sel;
  AncMap anc;
  bool header_seen = false, seen_chrom = false, done = false;
  int64_t n_matched = 0, n_anc = 0;
  IndexOut out;
  std::vector<IndexOut> outs;
  std::unique_ptr<WorkerPool> pool;
};

namespace {

int bgzf_reader_run(sai_bgzf_stream* st) {
  const char* path = st->path.c_str();
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
  struct FdGuard { int fd; ~FdGuard() { close(fd); } } guard{fd};
  struct stat sb;
  if (fstat(fd, &sb) != 0) return sai_set_error(SAI_ERR_ARG, "cannot stat %s", path);
  const size_t total = static_cast<size_t>(sb.st_size);
  size_t file_off = static_cast<size_t>(st->file_begin);
  if (st->nothing_to_read || file_off >= total) return SAI_OK;
  bool past_stop = false;
  constexpr size_t kPastStopBytes = size_t(1) << 18;  // per batch beyond the index's bound
  WorkerPool readers(std::max(1, std::min(st->n_threads, 8)));
  while (file_off < total) {
    int b;
    {
      std::unique_lock<std::mutex> lk(st->m);
      b = static_cast<int>(st->produced % 2);
      st->cv.wait(lk, [&] { return st->state[b] == 0 || st->cancel || st->consumer_done; });
      if (st->cancel || st->consumer_done) return SAI_OK;
    }
    unsigned char* dst = st->bufs[b];
    size_t want = std::min(st->cap - 8, total - file_off);  // 8 bytes of zero padding behind the data
    if (st->file_stop >= 0) {
      // A region.  The index only says where reading may SLOW DOWN, not where it ends: a linear-index entry
      // is the first record that OVERLAPS a 16 kb window (REF length, INFO/END), so an indel or SV that
      // starts before the region's end and reaches into the next window makes `file_stop` ITS member, with
      // records of the region still behind it.  Up to that member in one piece (a member is < 64 KiB +
      // header), beyond it a few members at a time until the record index reports a record past the end
      // (`consumer_done`) -- what the host readers do (vcf_stream.cpp, vcf_ingest.cpp: seek to the start,
      // stop at POS > end).
      const size_t stop = static_cast<size_t>(st->file_stop);
      want = std::min(want, past_stop || file_off >= stop ? kPastStopBytes : stop + (size_t(1) << 17) - file_off);
    }
    if (want == 0) break;
    {
      // one thread copies ~3 GB/s out of the page cache, i.e. ~35 GB/s of text: not enough
      const int rt = static_cast<int>(std::min<size_t>(static_cast<size_t>(readers.size()), want / (size_t(1) << 20) + 1));
      std::vector<char> bad(static_cast<size_t>(rt), 0);
      auto piece = [&](int t) {
        size_t lo = want * static_cast<size_t>(t) / static_cast<size_t>(rt), hi = want * static_cast<size_t>(t + 1) / static_cast<size_t>(rt);
        while (lo < hi) {
          const ssize_t got = pread(fd, dst + lo, hi - lo, static_cast<off_t>(file_off + lo));
          if (got <= 0) { bad[static_cast<size_t>(t)] = 1; return; }
          lo += static_cast<size_t>(got);
        }
      };
      readers.run(rt, piece);
      for (char x : bad)
        if (x) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
    }
    const size_t have = want;
    std::vector<sai_bgzf_member>& mem = st->members[b];
    mem.clear();
    size_t off = 0, out_total = 0;
    while (off < have) {
      size_t hlen = 0;
      const long bsize = bgzf_member_size(dst + off, have - off, &hlen);
      if (bsize < 0) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block", path);
      if (bsize == 0 || off + static_cast<size_t>(bsize) > have) break;  // incomplete member: next batch
      if (static_cast<size_t>(bsize) < hlen + 8) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block", path);
      const unsigned char* tail = dst + off + bsize - 8;
      const uint32_t isize = le32(tail + 4);
      if (isize > 65536u) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block (ISIZE > 64 KiB)", path);
      if (out_total + isize > st->text_cap) {
        if (mem.empty()) return sai_set_error(SAI_ERR_ARG, "text batch smaller than one BGZF block");
        break;
      }
      sai_bgzf_member r;
      r.data_off = static_cast<int64_t>(off + hlen);
      r.out_off = static_cast<int64_t>(out_total);
      r.data_len = static_cast<uint32_t>(static_cast<size_t>(bsize) - hlen - 8);
      r.isize = isize;
      r.crc = le32(tail);
      r.reserved = 0;
      mem.push_back(r);
      out_total += isize;
      const bool at_stop = st->file_stop >= 0 && file_off + off >= static_cast<size_t>(st->file_stop);
      off += static_cast<size_t>(bsize);
      if (at_stop && !past_stop) {  // hand this batch over now: usually the index then says "done"
        past_stop = true;
        break;
      }
    }
    if (mem.empty()) {
      if (file_off + have >= total) return sai_set_error(SAI_ERR_ARG, "%s: truncated BGZF file", path);
      return sai_set_error(SAI_ERR_UNSUPPORTED, "%s: a BGZF block is larger than the staging buffer", path);
    }
    const size_t padded = (off + 3) & ~size_t(3);
    memset(dst + off, 0, padded + 4 - off);
    {
      std::lock_guard<std::mutex> lk(st->m);
      st->comp_bytes[b] = padded + 4;
      st->text_bytes[b] = out_total;
      st->state[b] = 1;
      ++st->produced;
    }
    st->cv.notify_all();
    file_off += off;
  }
  return SAI_OK;
}

void bgzf_reader_thread(sai_bgzf_stream* st) {
  int rc;
  std::string err;
  try {
    rc = bgzf_reader_run(st);
    if (rc) err = sai_last_error();
  } catch (const std::exception& e) {
    rc = SAI_ERR_HIP;
    err = std::string("sai_bgzf_stream: ") + e.what();
  } catch (...) {
    rc = SAI_ERR_HIP;
    err = "sai_bgzf_stream: unknown failure";
  }
  {
    std::lock_guard<std::mutex> lk(st->m);
    st->rc = rc;
    st->err = err;
    st->finished = true;
  }
  st->cv.notify_all();
}

// index_lines on the line heads the GPU gathered (sai_text_line_starts / _heads): line i of the
// batch starts at start[i], its first min(H, length) bytes are heads[i * H ...], info[i] says how long
// its fixed columns are and whether it ends with "\r\n".  Same decisions, same outputs.
void index_head_lines(const char* heads, int32_t H, const int64_t* start, const int32_t* info, int64_t i0, int64_t i1,
                      const std::string& chrom, int64_t region_start, int64_t stop, const AncMap& anc, IndexOut& out) {
  for (int64_t i = i0; i < i1; ++i) {
    const char* line = heads + i * static_cast<int64_t>(H);
    const int cr = static_cast<int>(static_cast<uint32_t>(info[i]) >> 31);
    const int64_t line_len = start[i + 1] - 1 - cr - start[i];
    if (line_len <= 0 || *line == '#') continue;
    const int32_t fixed = info[i] & 0x7FFFFFFF;
    const char* le = line + std::min<int64_t>(H, line_len);  // end of what is known of the line
    const char* t1 = find_tab(line, le);
    if (static_cast<size_t>(t1 - line) != chrom.size() || memcmp(line, chrom.data(), chrom.size()) != 0) {
      out.last_line_other = true;
      continue;
    }
    if (t1 >= le) continue;
    out.last_line_other = false;
    out.saw_chrom = true;
    const char* f = t1 + 1;
    int64_t pos = 0;
    while (f < le && *f >= '0' && *f <= '9') pos = pos * 10 + (*f++ - '0');
    if (stop >= 0 && pos > stop) out.beyond_stop = true;
    if ((region_start >= 0 && pos < region_start) || (stop >= 0 && pos > stop)) continue;
    ++out.matched;
    if (fixed > H) { out.error = "line head shorter than the fixed columns at " + chrom + ":" + std::to_string(pos); return; }
    const char* col[10];
    col[0] = line;
    col[1] = t1 + 1;
    const char* q = find_tab(f, le);
    bool ok = true;
    for (int c = 2; c <= 9; ++c) {
      if (q >= le) { ok = false; break; }
      col[c] = q + 1;
      q = find_tab(col[c], le);
    }
    if (!ok) { out.error = "record with fewer than 10 columns at " + chrom + ":" + std::to_string(pos); return; }
    bool flip = false;
    if (anc.active) {
      auto it = anc.allele.find(pos);
      if (it == anc.allele.end()) continue;
      const char* ref = col[3];
      const size_t ref_len = static_cast<size_t>(col[4] - 1 - col[3]);
      const char* alt = col[4];
      const char* alt_end = col[5] - 1;
      const void* comma = memchr(alt, ',', static_cast<size_t>(alt_end - alt));
      const size_t alt_len = static_cast<size_t>((comma ? static_cast<const char*>(comma) : alt_end) - alt);
      const AncAllele& a = it->second;
      if (a.size() == alt_len && memcmp(a.data(), alt, alt_len) == 0) flip = true;
      else if (!(a.size() == ref_len && memcmp(a.data(), ref, ref_len) == 0)) continue;
    }
    int gi = -1;
    {
      const char* fs = col[8];
      const char* fe = col[9] - 1;
      int k = 0;
      while (fs <= fe) {
        const void* c = memchr(fs, ':', static_cast<size_t>(fe - fs));
        const char* ce = c ? static_cast<const char*>(c) : fe;
        if (ce - fs == 2 && fs[0] == 'G' && fs[1] == 'T') { gi = k; break; }
        if (!c) break;
        fs = ce + 1;
        ++k;
      }
    }
    if (gi < 0) { out.error = "record " + chrom + ":" + std::to_string(pos) + " has no GT field"; return; }
    const int64_t samples_at = col[9] - line;
    if (gi > 255 || line_len - samples_at > 0x7FFFFFFF) { out.error = "record " + chrom + ":" + std::to_string(pos) + " is outside the streaming limits"; return; }
    out.off.push_back(start[i] + samples_at);
    out.len.push_back(static_cast<int32_t>(line_len - samples_at));
    out.pos.push_back(static_cast<int32_t>(pos));
    out.flip.push_back(flip ? 1 : 0);
    out.gi.push_back(static_cast<uint8_t>(gi));
  }
}

uint32_t crc32_of(const void* p, size_t n) {
  if (const LibDeflate* l = libdeflate()) return l->crc32(0, p, n);
  return static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), static_cast<const unsigned char*>(p), static_cast<uInt>(n)));
}

}  // namespace

extern "C" {

int sai_bgzf_stream_open(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                         const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                         int32_t n_threads, void* comp0_host, void* comp1_host, int64_t comp_buffer_bytes,
                         int64_t text_batch_bytes, sai_bgzf_stream** stream_out) {
  return guarded("sai_bgzf_stream_open", [&]() -> int {
    if (!path || !chrom || !stream_out) return sai_set_error(SAI_ERR_ARG, "NULL argument");
    *stream_out = nullptr;
    // n_samples == 0: only the record index is wanted (positions of a chromosome: ChunkGenerator's scan)
    if (n_samples < 0 || (n_samples > 0 && (!sample_names || !ploidy))) return sai_set_error(SAI_ERR_ARG, "bad sample selection");
    if (!comp0_host || !comp1_host || comp_buffer_bytes < (1 << 17) || text_batch_bytes < (1 << 16))
      return sai_set_error(SAI_ERR_ARG, "two staging buffers of at least 128 KiB and a text batch of at least 64 KiB are needed");
    for (int32_t s = 0; s < n_samples; ++s)
      if (ploidy[s] < 1 || ploidy[s] > 64) return sai_set_error(SAI_ERR_ARG, "ploidy of sample %d out of range", s);
    {
      FILE* f = fopen(path, "rb");
      if (!f) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
      fclose(f);
    }
    if (!file_is_bgzf(path)) return sai_set_error(SAI_ERR_UNSUPPORTED, "%s is not a bgzip file", path);
    std::unique_ptr<sai_bgzf_stream> st(new sai_bgzf_stream);
    if (start >= 0) {
      // A region of a file with a usable index: only the members from the region's first record (the
      // linear index, one entry per 16 kb window; utils.py:117-138 gets the same from scikit-allel) to the
      // member of the first record of a LATER window take the trip -- each worker of a sharded run reads
      // its own region (chunk_generator.py:130-142), not the file.  The record index filters by POS as
      // always and decides when the region is over (bgzf_reader_run reads on past the bound, a few members
      // at a time, until it says so), so a coarse or stale bound costs bytes, never records: an entry equal
      // to the window's own (an index that fills empty windows from the previous one) is not taken.
      TbiRef idx;
      if (load_tbi(path, chrom, idx)) {
        const uint64_t w0 = static_cast<uint64_t>(start > 0 ? start - 1 : 0) >> 14;
        if (!idx.present || w0 >= idx.ioff.size()) {
          st->nothing_to_read = true;
        } else {
          st->file_begin = static_cast<int64_t>(idx.ioff[w0] >> 16);
          st->first_text_skip = static_cast<int64_t>(idx.ioff[w0] & 0xFFFFu);
          if (end >= 0) {
            const uint64_t w1 = std::min<uint64_t>(static_cast<uint64_t>(end > 0 ? end - 1 : 0) >> 14, idx.ioff.size() - 1);
            for (uint64_t k = w1 + 1; k < idx.ioff.size(); ++k)
              if (idx.ioff[k] > idx.ioff[w1]) {
                st->file_stop = static_cast<int64_t>(idx.ioff[k] >> 16);
                break;
              }
            if (st->file_stop >= 0 && st->file_stop < st->file_begin) st->file_stop = -1;  // not a sorted index: no end bound
          }
        }
      }
    }
    st->path = path;
    st->chrom = chrom;
    st->start = start;
    st->end = end;
    st->n_threads = n_threads < 1 ? 1 : n_threads;
    for (int32_t s = 0; s < n_samples; ++s) {
      st->names.emplace_back(sample_names[s]);
      st->ploidy.push_back(ploidy[s]);
    }
    if (anc_bed_path)
      if (int rc = load_anc(anc_bed_path, st->chrom, st->start, st->end, st->anc, &st->n_anc)) return rc;
    {  // the header, from the top of the file (a few blocks, inflated here): the record index --
       // from the text or from the line heads the GPU extracts -- then only ever skips '#' lines
      std::vector<const char*> names;
      for (auto& n : st->names) names.push_back(n.c_str());
      sai_bgzf_stream* raw = st.get();
      const int hrc = for_each_block_from(path, 1, 0, size_t(1) << 16, [&](const char* p, const char* endp) -> int {
        while (!raw->header_seen && p < endp) {
          const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(endp - p)));
          if (!eol) eol = endp;
          if (*p != '#') return sai_set_error(SAI_ERR_ARG, "%s: no #CHROM header line before the records", path);
          if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) {
            if (int rc = parse_header(p, eol, path, static_cast<int32_t>(names.size()), names.data(), raw->ploidy.data(), raw->sel))
              return rc;
            raw->header_seen = true;
          }
          p = eol + 1;
        }
        return (raw->header_seen || p < endp) ? 1 : 0;
      });
      if (hrc) return hrc;
      if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
    }
    st->bufs[0] = static_cast<unsigned char*>(comp0_host);
    st->bufs[1] = static_cast<unsigned char*>(comp1_host);
    st->cap = static_cast<size_t>(comp_buffer_bytes);
    st->text_cap = static_cast<size_t>(text_batch_bytes);
    st->pool.reset(new WorkerPool(st->n_threads));
    st->outs.resize(static_cast<size_t>(st->n_threads));
    sai_bgzf_stream* raw = st.get();
    st->reader = std::thread(bgzf_reader_thread, raw);
    *stream_out = st.release();
    return SAI_OK;
  });
}

int sai_bgzf_stream_next(sai_bgzf_stream* st, int32_t* buffer_index, int64_t* n_comp_bytes, int32_t* n_members,
                         const sai_bgzf_member** members_host, int64_t* n_text_bytes, int32_t* done) {
  if (!st || !buffer_index || !n_comp_bytes || !n_members || !members_host || !n_text_bytes || !done)
    return sai_set_error(SAI_ERR_ARG, "NULL argument");
  std::unique_lock<std::mutex> lk(st->m);
  if (st->held >= 0) {  // the caller is done with the compressed bytes it got last time
    st->state[st->held] = 0;
    st->held = -1;
    st->cv.notify_all();
  }
  const int b = static_cast<int>(st->consumed % 2);
  st->cv.wait(lk, [&] { return st->state[b] == 1 || st->finished; });
  if (st->state[b] != 1) {
    *done = 1;
    *n_members = 0;
    *n_comp_bytes = *n_text_bytes = 0;
    *buffer_index = -1;
    *members_host = nullptr;
    if (st->rc) return sai_set_error(st->rc, "%s", st->err.c_str());
    return SAI_OK;
  }
  st->state[b] = 2;
  st->held = b;
  ++st->consumed;
  *done = 0;
  *buffer_index = b;
  *n_comp_bytes = static_cast<int64_t>(st->comp_bytes[b]);
  *n_members = static_cast<int32_t>(st->members[b].size());
  *members_host = st->members[b].data();
  *n_text_bytes = static_cast<int64_t>(st->text_bytes[b]);
  return SAI_OK;
}

int sai_bgzf_stream_region(sai_bgzf_stream* st, int64_t* file_begin, int64_t* file_stop, int64_t* first_text_skip) {
  if (!st || !file_begin || !file_stop || !first_text_skip) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  *file_begin = st->nothing_to_read ? -1 : st->file_begin;
  *file_stop = st->file_stop;
  *first_text_skip = st->first_text_skip;
  return SAI_OK;
}

int sai_bgzf_stream_release(sai_bgzf_stream* st) {
  if (!st) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(st->m);
  if (st->held >= 0) {  // the compressed bytes have left the pinned buffer: the reader may refill it now
    st->state[st->held] = 0;
    st->held = -1;
    st->cv.notify_all();
  }
  return SAI_OK;
}

int sai_vcf_index_text(sai_bgzf_stream* st, const char* text_host, int64_t n_bytes, int64_t n_carry,
                       const sai_bgzf_member* members_host, int32_t n_members, int32_t is_last, int64_t* n_usable,
                       int64_t* n_lines, const int64_t** line_off_host, const int32_t** line_len_host,
                       const int32_t** line_pos_host, const uint8_t** line_flip_host, const uint8_t** line_gi_host,
                       int32_t* done) {
  return guarded("sai_vcf_index_text", [&]() -> int {
    if (!st || !n_usable || !n_lines || !line_off_host || !line_len_host || !line_pos_host || !line_flip_host ||
        !line_gi_host || !done)
      return sai_set_error(SAI_ERR_ARG, "NULL argument");
    if (n_bytes < 0 || n_carry < 0 || n_carry > n_bytes || n_members < 0 || (n_bytes > 0 && !text_host) ||
        (n_members > 0 && !members_host))
      return sai_set_error(SAI_ERR_ARG, "bad text range");
    const char* path = st->path.c_str();
    const int nt = std::max(1, st->n_threads);
    // 1. the text of every member against the CRC-32 of its trailer
    if (n_members > 0) {
      std::vector<char> bad(static_cast<size_t>(nt), 0);
      const char* base = text_host + n_carry;
      const int64_t room = n_bytes - n_carry;
      auto check = [&](int t) {
        const size_t lo = static_cast<size_t>(n_members) * static_cast<size_t>(t) / static_cast<size_t>(nt);
        const size_t hi = static_cast<size_t>(n_members) * static_cast<size_t>(t + 1) / static_cast<size_t>(nt);
        for (size_t i = lo; i < hi; ++i) {
          const sai_bgzf_member& r = members_host[i];
          if (r.out_off < 0 || r.out_off + static_cast<int64_t>(r.isize) > room) { bad[static_cast<size_t>(t)] = 2; return; }
          if (r.isize && crc32_of(base + r.out_off, r.isize) != r.crc) { bad[static_cast<size_t>(t)] = 1; return; }
        }
      };
      st->pool->run(nt, check);
      for (char b : bad) {
        if (b == 2) return sai_set_error(SAI_ERR_ARG, "member table does not fit the text");
        if (b) return sai_set_error(SAI_ERR_ARG, "%s: BGZF block fails to inflate or its CRC", path);
      }
    }
    const char* p = text_host;
    const char* endp = text_host + n_bytes;
    // 2. whole lines only; the rest is the caller's carry (the last batch may end without a newline)
    const char* cut = endp;
    if (!is_last) {
      while (cut > p && cut[-1] != '\n') --cut;
    }
    // 3. the header
    while (!st->header_seen && p < cut) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(cut - p)));
      if (!eol) eol = cut;
      if (*p != '#') return sai_set_error(SAI_ERR_ARG, "%s: no #CHROM header line before the records", path);
      if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) {
        std::vector<const char*> names;
        for (auto& n : st->names) names.push_back(n.c_str());
        if (int hrc = parse_header(p, eol, path, static_cast<int32_t>(names.size()), names.data(), st->ploidy.data(), st->sel))
          return hrc;
        st->header_seen = true;
      }
      p = eol < cut ? eol + 1 : cut;
    }
    if (is_last && !st->header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
    // 4. the record lines, in parallel pieces cut at line ends
    st->out.clear();
    if (p < cut && st->header_seen && !st->done) {
      const size_t total = static_cast<size_t>(cut - p);
      std::vector<const char*> edge(static_cast<size_t>(nt) + 1, cut);
      edge[0] = p;
      for (int t = 1; t < nt; ++t) {
        const char* guess = p + total * static_cast<size_t>(t) / static_cast<size_t>(nt);
        if (guess < edge[static_cast<size_t>(t) - 1]) guess = edge[static_cast<size_t>(t) - 1];
        const char* nl = static_cast<const char*>(memchr(guess, '\n', static_cast<size_t>(cut - guess)));
        edge[static_cast<size_t>(t)] = nl ? nl + 1 : cut;
      }
      for (auto& o : st->outs) o.clear();
      auto piece = [&](int t) {
        IndexOut& o = st->outs[static_cast<size_t>(t)];
        if (edge[static_cast<size_t>(t)] >= edge[static_cast<size_t>(t) + 1]) return;
        try {
          index_lines(edge[static_cast<size_t>(t)], edge[static_cast<size_t>(t) + 1], text_host, st->chrom, st->start, st->end, st->anc, o);
        } catch (...) {
          o.failed = true;
        }
      };
      st->pool->run(nt, piece);
      for (auto& o : st->outs) {
        if (o.failed) return sai_set_error(SAI_ERR_HIP, "%s: indexing failed (out of memory)", path);
        if (!o.error.empty()) return sai_set_error(SAI_ERR_ARG, "%s: %s", path, o.error.c_str());
        st->n_matched += o.matched;
        st->out.off.insert(st->out.off.end(), o.off.begin(), o.off.end());
        st->out.len.insert(st->out.len.end(), o.len.begin(), o.len.end());
        st->out.pos.insert(st->out.pos.end(), o.pos.begin(), o.pos.end());
        st->out.flip.insert(st->out.flip.end(), o.flip.begin(), o.flip.end());
        st->out.gi.insert(st->out.gi.end(), o.gi.begin(), o.gi.end());
        st->seen_chrom = st->seen_chrom || o.saw_chrom;
        if (o.beyond_stop || (st->seen_chrom && o.last_line_other)) st->done = true;
      }
      if (st->done) {
        {
          std::lock_guard<std::mutex> lk(st->m);
          st->consumer_done = true;
        }
        st->cv.notify_all();
      }
    }
    *n_usable = static_cast<int64_t>(cut - text_host);
    *n_lines = static_cast<int64_t>(st->out.off.size());
    *line_off_host = st->out.off.data();
    *line_len_host = st->out.len.data();
    *line_pos_host = st->out.pos.data();
    *line_flip_host = st->out.flip.data();
    *line_gi_host = st->out.gi.data();
    *done = st->done ? 1 : 0;
    return SAI_OK;
  });
}

int sai_vcf_index_heads(sai_bgzf_stream* st, const char* heads_host, int32_t head_bytes, const int64_t* line_start_host,
                        const int32_t* line_info_host, int64_t n_lines, int64_t* n_lines_out,
                        const int64_t** line_off_host, const int32_t** line_len_host, const int32_t** line_pos_host,
                        const uint8_t** line_flip_host, const uint8_t** line_gi_host, int32_t* done) {
  return guarded("sai_vcf_index_heads", [&]() -> int {
    if (!st || !n_lines_out || !line_off_host || !line_len_host || !line_pos_host || !line_flip_host || !line_gi_host || !done)
      return sai_set_error(SAI_ERR_ARG, "NULL argument");
    if (n_lines < 0 || head_bytes < 4 || (n_lines > 0 && (!heads_host || !line_start_host || !line_info_host)))
      return sai_set_error(SAI_ERR_ARG, "bad line table");
    const char* path = st->path.c_str();
    if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
    const int nt = std::max(1, st->n_threads);
    st->out.clear();
    if (n_lines > 0 && !st->done) {
      for (auto& o : st->outs) o.clear();
      auto piece = [&](int t) {
        IndexOut& o = st->outs[static_cast<size_t>(t)];
        const int64_t i0 = n_lines * t / nt, i1 = n_lines * (t + 1) / nt;
        if (i0 >= i1) return;
        try {
          index_head_lines(heads_host, head_bytes, line_start_host, line_info_host, i0, i1, st->chrom, st->start, st->end, st->anc, o);
        } catch (...) {
          o.failed = true;
        }
      };
      st->pool->run(nt, piece);
      for (auto& o : st->outs) {
        if (o.failed) return sai_set_error(SAI_ERR_HIP, "%s: indexing failed (out of memory)", path);
        if (!o.error.empty()) return sai_set_error(SAI_ERR_ARG, "%s: %s", path, o.error.c_str());
        st->n_matched += o.matched;
        st->out.off.insert(st->out.off.end(), o.off.begin(), o.off.end());
        st->out.len.insert(st->out.len.end(), o.len.begin(), o.len.end());
        st->out.pos.insert(st->out.pos.end(), o.pos.begin(), o.pos.end());
        st->out.flip.insert(st->out.flip.end(), o.flip.begin(), o.flip.end());
        st->out.gi.insert(st->out.gi.end(), o.gi.begin(), o.gi.end());
        st->seen_chrom = st->seen_chrom || o.saw_chrom;
        if (o.beyond_stop || (st->seen_chrom && o.last_line_other)) st->done = true;
      }
      if (st->done) {
        {
          std::lock_guard<std::mutex> lk(st->m);
          st->consumer_done = true;
        }
        st->cv.notify_all();
      }
    }
    *n_lines_out = static_cast<int64_t>(st->out.off.size());
    *line_off_host = st->out.off.data();
    *line_len_host = st->out.len.data();
    *line_pos_host = st->out.pos.data();
    *line_flip_host = st->out.flip.data();
    *line_gi_host = st->out.gi.data();
    *done = st->done ? 1 : 0;
    return SAI_OK;
  });
}

int sai_bgzf_stream_selection(sai_bgzf_stream* st, int32_t* slot_of_col_host, int32_t capacity, int32_t* n_cols,
                              int64_t* n_matched, int64_t* n_anc_entries) {
  if (!st || !n_cols) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  if (!st->header_seen) return sai_set_error(SAI_ERR_ARG, "the header has not been read yet");
  *n_cols = st->sel.max_col + 1;
  if (slot_of_col_host) {
    if (capacity < *n_cols) return sai_set_error(SAI_ERR_ARG, "slot_of_col capacity %d < %d", capacity, *n_cols);
    for (int32_t c = 0; c < *n_cols; ++c) slot_of_col_host[c] = st->sel.slot_of_col[static_cast<size_t>(c)];
  }
  if (n_matched) *n_matched = st->n_matched;
  if (n_anc_entries) *n_anc_entries = st->n_anc;
  return SAI_OK;
}

int sai_bgzf_stream_close(sai_bgzf_stream* st) {
  if (!st) return SAI_OK;
  {
    std::lock_guard<std::mutex> lk(st->m);
    st->cancel = true;
  }
  st->cv.notify_all();
  if (st->reader.joinable()) st->reader.join();
  delete st;
  return SAI_OK;
}

}  // extern "C"
