// Host-side VCF ingest for libsaihip (first "next" row): one pass over a
// plain or gzip/bgzip VCF, multithreaded tokenising of the record lines, output = unphased ALT
// dosage as int8 [record][sample] in the reference's matrix order plus int32 positions -- what
// sai/utils/utils.py:78-186 + 389-410 obtain from scikit-allel (GT of the selected samples,
// first ALT, region filter, '.' = -1, calls padded / cut to the population's ploidy, ploidy axis
// summed) -- and, when an ancestral-allele BED is given, the polarisation of utils.py:435-555
// (keep only listed sites whose ancestral allele is REF or ALT; where ALT is ancestral every
// allele call a becomes |a - 1|, so a missing allele becomes 2 exactly as in the reference).
// The Python reader sai_amd/utils/vcf.py is the readable statement of the same rules; the two are
// tested against each other and against the reference tests' expectations.

#include "ingest_base.hpp"

struct sai_vcf_block {
  int32_t n_samples = 0;
  int64_t n_matched = 0;      // records of the chromosome inside the region
  int64_t n_anc_entries = 0;  // ancestral-allele entries loaded for the chromosome/region
  std::vector<int32_t> pos;
  std::vector<int8_t> dosage;  // [record][sample]
};

extern "C" {


static int vcf_scan_impl(const char* path, const char* chrom, int64_t* first_pos, int64_t* last_pos) {
  if (!path || !chrom || !first_pos || !last_pos) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  const std::string c(chrom);
  int64_t first = -1, last = -1;
  bool header_seen = false;
  auto scan = [&](const char* p, const char* end) -> int {
    while (p < end) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(end - p)));
      if (!eol) eol = end;
      if (*p == '#') {
        if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) header_seen = true;
      } else if (eol > p) {
        const char* t1 = find_tab(p, eol);
        if (static_cast<size_t>(t1 - p) == c.size() && memcmp(p, c.data(), c.size()) == 0 && t1 < eol) {
          int64_t v = 0;
          for (const char* f = t1 + 1; f < eol && *f >= '0' && *f <= '9'; ++f) v = v * 10 + (*f - '0');
          if (first < 0) first = v;
          last = v;
        } else if (first >= 0) {
          return 1;  // the first contiguous run of the chromosome is over (chunk_generator.py:66-73)
        }
      }
      p = eol + 1;
    }
    return 0;
  };
  // the full pass, a batch at a time with kScanThreads pieces side by side: a piece reports the first
  // run of the chromosome inside it, the pieces are merged in file order
  struct ScanPiece {
    int64_t first = -1, last = -1;
    bool other_before = false, ended = false, header = false;
  };
  auto scan_piece = [&c](const char* p, const char* end, ScanPiece& r) {
    while (p < end) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(end - p)));
      if (!eol) eol = end;
      if (*p == '#') {
        if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) r.header = true;
      } else if (eol > p) {
        const char* t1 = find_tab(p, eol);
        if (static_cast<size_t>(t1 - p) == c.size() && memcmp(p, c.data(), c.size()) == 0 && t1 < eol) {
          int64_t v = 0;
          for (const char* f = t1 + 1; f < eol && *f >= '0' && *f <= '9'; ++f) v = v * 10 + (*f - '0');
          if (r.first < 0) r.first = v;
          r.last = v;
        } else if (r.first >= 0) {
          r.ended = true;
          return;
        } else {
          r.other_before = true;
        }
      }
      p = eol + 1;
    }
  };
  auto scan_parallel = [&](const char* p, const char* end) -> int {
    const size_t total = static_cast<size_t>(end - p);
    const int nt = total < (size_t(1) << 20) ? 1 : kScanThreads;
    std::vector<const char*> edge(static_cast<size_t>(nt) + 1, end);
    edge[0] = p;
    for (int t = 1; t < nt; ++t) {
      const char* guess = p + total * static_cast<size_t>(t) / static_cast<size_t>(nt);
      if (guess < edge[static_cast<size_t>(t) - 1]) guess = edge[static_cast<size_t>(t) - 1];
      const char* nl = static_cast<const char*>(memchr(guess, '\n', static_cast<size_t>(end - guess)));
      edge[static_cast<size_t>(t)] = nl ? nl + 1 : end;
    }
    std::vector<ScanPiece> res(static_cast<size_t>(nt));
    {
      ThreadGroup th;  // scan_piece allocates nothing: no throw inside the workers
      for (int t = 1; t < nt; ++t)
        th.spawn([&, t] { scan_piece(edge[static_cast<size_t>(t)], edge[static_cast<size_t>(t) + 1], res[static_cast<size_t>(t)]); });
      scan_piece(edge[0], edge[1], res[0]);
      th.join();
    }
    for (const ScanPiece& r : res) {
      header_seen = header_seen || r.header;
      if (first < 0) {
        if (r.first >= 0) {
          first = r.first;
          last = r.last;
          if (r.ended) return 1;
        }
      } else {
        if (r.other_before) return 1;  // the run ended where the previous piece ended
        if (r.first >= 0) {
          last = r.last;
          if (r.ended) return 1;
        }
      }
    }
    return 0;
  };
  TbiRef idx;
  if (file_is_bgzf(path) && load_tbi(path, c, idx)) {
    // indexed: the first record sits at the smallest chunk start, the last one inside the chunk that
    // starts last / the last 16 kb window -- two short reads instead of the whole file
    if (idx.present && idx.first_voff != ~0ull) {
      int64_t lo = -1;
      int rc = for_each_block_from(path, 1, idx.first_voff, size_t(1) << 16, [&](const char* p, const char* end) -> int {
        const int r = scan(p, end);
        return (r != 0 || first >= 0) ? 1 : 0;
      });
      if (rc) return rc;
      lo = first;
      first = -1;
      // the last record lies behind the start of the chunk that starts last AND behind the first record of
      // the last 16 kb window that holds one (the linear index's last entry) -- an index with one long
      // chunk per chromosome would otherwise send this read back to the chromosome's first record
      const uint64_t tail_voff = std::max(idx.last_chunk_voff, idx.ioff.empty() ? uint64_t{0} : idx.ioff.back());
      rc = for_each_block_from(path, kScanThreads, tail_voff, size_t(4) << 20, scan);
      if (rc) return rc;
      if (lo >= 0 && last >= 0) {
        *first_pos = lo;
        *last_pos = last;
        return SAI_OK;
      }
      first = last = -1;  // index and file disagree: fall through to the full pass
    } else {
      *first_pos = *last_pos = -1;  // chromosome not in the index
      return SAI_OK;
    }
  }
  if (file_is_plain_text(path)) {
    // Uncompressed text: every thread walks its own part of the file through a small buffer of its
    // own (1 MiB: faulted in once, then cache-resident), so the scan costs one pass over the page
    // cache -- no 48 MiB batch buffer to fault in, no mapping to build and tear down (measured on
    // 241 MB: mmap 34 ms, mmap + MAP_POPULATE 18 ms, pread into fresh memory 165 ms, this 5 ms).
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
    struct FdGuard { int fd; ~FdGuard() { close(fd); } } guard{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) return sai_set_error(SAI_ERR_ARG, "cannot stat %s", path);
    const size_t total = static_cast<size_t>(sb.st_size);
    const int nt = total < (size_t(4) << 20) ? 1 : kScanThreads;
    std::vector<ScanPiece> res(static_cast<size_t>(nt));
    std::vector<char> io_bad(static_cast<size_t>(nt), 0);
    auto walk = [&](int t) {
      const size_t a = total * static_cast<size_t>(t) / static_cast<size_t>(nt);
      const size_t z = total * static_cast<size_t>(t + 1) / static_cast<size_t>(nt);
      ScanPiece& r = res[static_cast<size_t>(t)];
      try {
        std::vector<char> buf(size_t(1) << 20);
        bool skip = false;  // the line that straddles `a` belongs to the previous part
        if (a > 0) {
          char c = 0;
          if (pread(fd, &c, 1, static_cast<off_t>(a - 1)) != 1) { io_bad[static_cast<size_t>(t)] = 1; return; }
          skip = c != '\n';
        }
        size_t off = a, have = 0;     // file offset of the next read; bytes carried at the front of buf
        size_t line0 = a;             // file offset of buf[0]
        while (line0 < z && off < total && !r.ended) {
          if (have == buf.size()) buf.resize(buf.size() * 2);  // one line longer than the buffer
          const ssize_t got = pread(fd, buf.data() + have, std::min(buf.size() - have, total - off), static_cast<off_t>(off));
          if (got <= 0) { io_bad[static_cast<size_t>(t)] = 1; return; }
          off += static_cast<size_t>(got);
          have += static_cast<size_t>(got);
          const bool eof = off >= total;
          // whole lines in buf[0, usable); only those that START before z are this part's
          size_t usable = have;
          if (!eof) {
            usable = 0;
            for (size_t i = have; i > 0; --i)
              if (buf[i - 1] == '\n') { usable = i; break; }
            if (usable == 0) continue;
          }
          size_t begin = 0;
          if (skip) {
            const void* nl = memchr(buf.data(), '\n', usable);
            if (!nl) { begin = usable; } else { begin = static_cast<size_t>(static_cast<const char*>(nl) - buf.data()) + 1; skip = false; }
          }
          size_t stop = usable;
          if (line0 + usable > z) {  // cut after the line that holds byte z - 1
            const size_t rel = z - line0;  // first byte that may start a foreign line
            if (rel <= begin) stop = begin;
            else {
              stop = rel;
              if (buf[rel - 1] != '\n') {
                const void* nl = memchr(buf.data() + rel, '\n', usable - rel);
                stop = nl ? static_cast<size_t>(static_cast<const char*>(nl) - buf.data()) + 1 : usable;
              }
            }
          }
          if (stop > begin) scan_piece(buf.data() + begin, buf.data() + stop, r);
          if (stop < usable) break;  // reached the end of this part
          const size_t rest = have - usable;
          if (rest) memmove(buf.data(), buf.data() + usable, rest);
          line0 += usable;
          have = rest;
          if (eof) break;
        }
      } catch (...) {
        io_bad[static_cast<size_t>(t)] = 2;
      }
    };
    {
      ThreadGroup th;
      for (int t = 1; t < nt; ++t) th.spawn([&walk, t] { walk(t); });
      walk(0);
      th.join();
    }
    for (char x : io_bad)
      if (x) return sai_set_error(x == 1 ? SAI_ERR_ARG : SAI_ERR_HIP, x == 1 ? "read error in %s" : "%s: out of host memory", path);
    for (const ScanPiece& r : res) {  // merge in file order, as scan_parallel does
      header_seen = header_seen || r.header;
      if (first < 0) {
        if (r.first >= 0) {
          first = r.first;
          last = r.last;
          if (r.ended) break;
        }
      } else {
        if (r.other_before) break;
        if (r.first >= 0) {
          last = r.last;
          if (r.ended) break;
        }
      }
    }
    if (!header_seen && first < 0) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
    *first_pos = first;
    *last_pos = last;
    return SAI_OK;
  }
  const int rc = for_each_block(path, kScanThreads, scan_parallel);
  if (rc) return rc;
  if (!header_seen && first < 0) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
  *first_pos = first;
  *last_pos = last;
  return SAI_OK;
}

static int vcf_load_impl(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                         const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path,
                         int32_t n_threads, sai_vcf_block** block_out) {
  if (!path || !chrom || !block_out) return sai_set_error(SAI_ERR_ARG, "NULL argument");
  *block_out = nullptr;
  if (n_samples < 1 || !sample_names || !ploidy) return sai_set_error(SAI_ERR_ARG, "empty sample selection");
  for (int32_t s = 0; s < n_samples; ++s)
    if (ploidy[s] < 1 || ploidy[s] > 64) return sai_set_error(SAI_ERR_ARG, "ploidy of sample %d out of range", s);
  if (n_threads < 1) n_threads = 1;
  const std::string c(chrom);
  std::unique_ptr<sai_vcf_block> holder(new sai_vcf_block);  // released to the caller only on success
  sai_vcf_block* blk = holder.get();
  blk->n_samples = n_samples;
  AncMap anc;
  if (anc_bed_path) {
    if (int rc = load_anc(anc_bed_path, c, start, end, anc, &blk->n_anc_entries)) return rc;
  }
  Selection sel;
  bool header_seen = false;
  // per-thread scratch lives across blocks: clear() keeps the capacity, so the allocator (and the
  // page-fault cost of fresh memory) is paid once, not per block
  std::vector<ThreadOut> outs(static_cast<size_t>(n_threads));
  WorkerPool tok_pool(n_threads);
  bool done = false, seen_chrom = false;  // early stop of an indexed (hence sorted) region read
  auto on_header = [&](const char*& p, const char* endp) -> int {
    while (!header_seen && p < endp) {  // header lines (serial)
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(endp - p)));
      if (!eol) eol = endp;
      if (*p != '#') return sai_set_error(SAI_ERR_ARG, "%s: no #CHROM header line before the records", path);
      if (eol - p > 6 && memcmp(p, "#CHROM", 6) == 0) {
        if (int hrc = parse_header(p, eol, path, n_samples, sample_names, ploidy, sel)) return hrc;
        header_seen = true;
      }
      p = eol + 1;
    }
    return 0;
  };
  auto on_records = [&](const char* p, const char* endp) -> int {
    if (p >= endp) return 0;
    // split [p, endp) into n_threads pieces at line boundaries
    std::vector<const char*> cut(static_cast<size_t>(n_threads) + 1, endp);
    cut[0] = p;
    const size_t total = static_cast<size_t>(endp - p);
    for (int t = 1; t < n_threads; ++t) {
      const char* guess = p + total * static_cast<size_t>(t) / static_cast<size_t>(n_threads);
      if (guess < cut[static_cast<size_t>(t) - 1]) guess = cut[static_cast<size_t>(t) - 1];
      const char* nl = static_cast<const char*>(memchr(guess, '\n', static_cast<size_t>(endp - guess)));
      cut[static_cast<size_t>(t)] = nl ? nl + 1 : endp;
    }
    for (auto& o : outs) {
      o.pos.clear();
      o.dosage.clear();
      o.matched = 0;
      o.saw_chrom = o.beyond_stop = o.last_line_other = o.failed = false;
    }
    auto piece = [&](int t) {  // an exception must not leave a worker thread (that is std::terminate)
      ThreadOut& o = outs[static_cast<size_t>(t)];
      try {
        parse_lines(cut[static_cast<size_t>(t)], cut[static_cast<size_t>(t) + 1], c, start, end, sel, anc, true, o);
      } catch (const std::exception& e) {
        try { o.error = std::string("tokenizer failed: ") + e.what(); } catch (...) { o.failed = true; }
      } catch (...) {
        o.failed = true;
      }
    };
    tok_pool.run(n_threads, [&](int t) {
      if (cut[static_cast<size_t>(t)] < cut[static_cast<size_t>(t) + 1]) piece(t);
    });
    for (auto& o : outs) {  // pieces in file order
      if (o.failed) return sai_set_error(SAI_ERR_HIP, "%s: tokenizer failed (out of memory)", path);
      if (!o.error.empty()) return sai_set_error(SAI_ERR_ARG, "%s: %s", path, o.error.c_str());
      blk->n_matched += o.matched;
      blk->pos.insert(blk->pos.end(), o.pos.begin(), o.pos.end());
      blk->dosage.insert(blk->dosage.end(), o.dosage.begin(), o.dosage.end());
      seen_chrom = seen_chrom || o.saw_chrom;
      if (o.beyond_stop || (seen_chrom && o.last_line_other)) done = true;
    }
    return 0;
  };
  int rc;
  TbiRef idx;
  if (start >= 0 && file_is_bgzf(path) && load_tbi(path, c, idx)) {
    // indexed region: the header from the top of the file, then the records from the region's first
    // 16 kb window until the first record past it
    rc = for_each_block_from(path, 1, 0, size_t(1) << 16, [&](const char* p, const char* endp) -> int {
      if (int hrc = on_header(p, endp)) return hrc;
      return (header_seen || p < endp) ? 1 : 0;
    });
    const uint64_t window = static_cast<uint64_t>(start > 0 ? start - 1 : 0) >> 14;
    if (rc == SAI_OK && header_seen && idx.present && window < idx.ioff.size()) {
      rc = for_each_block_from(path, n_threads, idx.ioff[window], 0, [&](const char* p, const char* endp) -> int {
        if (int rrc = on_records(p, endp)) return rrc;
        return done ? 1 : 0;
      });
    }
  } else {
    rc = for_each_block(path, n_threads, [&](const char* p, const char* endp) -> int {
      if (int hrc = on_header(p, endp)) return hrc;
      return on_records(p, endp);
    });
  }
  if (rc) return rc;
  if (!header_seen) return sai_set_error(SAI_ERR_ARG, "%s: not a VCF (no #CHROM header)", path);
  *block_out = holder.release();
  return SAI_OK;
}

int sai_vcf_scan(const char* path, const char* chrom, int64_t* first_pos, int64_t* last_pos) {
  return guarded("sai_vcf_scan", [&] { return vcf_scan_impl(path, chrom, first_pos, last_pos); });
}

int sai_vcf_load(const char* path, const char* chrom, int64_t start, int64_t end, int32_t n_samples,
                 const char* const* sample_names, const int32_t* ploidy, const char* anc_bed_path, int32_t n_threads,
                 sai_vcf_block** block_out) {
  return guarded("sai_vcf_load", [&] {
    return vcf_load_impl(path, chrom, start, end, n_samples, sample_names, ploidy, anc_bed_path, n_threads, block_out);
  });
}

int sai_vcf_block_info(const sai_vcf_block* block, int64_t* n_records, int64_t* n_matched, int64_t* n_anc_entries) {
  if (!block) return sai_set_error(SAI_ERR_ARG, "block is NULL");
  if (n_records) *n_records = static_cast<int64_t>(block->pos.size());
  if (n_matched) *n_matched = block->n_matched;
  if (n_anc_entries) *n_anc_entries = block->n_anc_entries;
  return SAI_OK;
}

int sai_vcf_block_copy(const sai_vcf_block* block, int32_t* pos_host, int8_t* dosage_host) {
  if (!block) return sai_set_error(SAI_ERR_ARG, "block is NULL");
  if (!block->pos.empty() && (!pos_host || !dosage_host)) return sai_set_error(SAI_ERR_ARG, "NULL output buffer");
  if (!block->pos.empty()) {
    memcpy(pos_host, block->pos.data(), block->pos.size() * sizeof(int32_t));
    memcpy(dosage_host, block->dosage.data(), block->dosage.size());
  }
  return SAI_OK;
}

int sai_vcf_block_free(sai_vcf_block* block) {
  delete block;
  return SAI_OK;
}

}  // extern "C"
