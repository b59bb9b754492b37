// The record index of a text batch (index_lines) and its output, shared by the two streaming readers.
#pragma once

#include "ingest_base.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// Streaming ingest for the GPU tokenizer (sai_vcf_stream_*): the host reads / inflates the file and
// INDEXES its record lines -- chromosome and region filter, POS, the ancestral-allele decision
// (keep / flip / drop: it needs only the fixed columns), the GT sub-field index, where the sample
// columns start -- while the genotype text itself crosses PCIe untouched and is tokenised by
// sai_tokenize_gt on the GPU.  A producer thread runs the same file walk as sai_vcf_load (plain,
// gzip, bgzip with parallel inflate, tabix seek, early stop) and fills the caller's two pinned
// buffers alternately; the consumer takes batch k while batch k+1 is being read.
// ------------------------------------------------------------------------------------------

struct IndexOut {
  std::vector<int64_t> off;   // first byte of the first sample column, relative to the batch text
  std::vector<int32_t> len;   // bytes from there to the end of the line (without "\r")
  std::vector<int32_t> pos;
  std::vector<uint8_t> flip, gi;
  int64_t matched = 0;
  bool saw_chrom = false, beyond_stop = false, last_line_other = false, failed = false;
  std::string error;
  void clear() {
    off.clear(); len.clear(); pos.clear(); flip.clear(); gi.clear();
    matched = 0;
    saw_chrom = beyond_stop = last_line_other = failed = false;
    error.clear();
  }
};

// The fixed columns of the record lines of [begin, end): parse_lines without the sample loop.
void index_lines(const char* begin, const char* end, const char* text0, const std::string& chrom, int64_t start,
                 int64_t stop, const AncMap& anc, IndexOut& out) {
  const char* p = begin;
  while (p < end) {
    const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(end - p)));
    if (!eol) eol = end;
    const char* line = p;
    p = eol + 1;
    const char* le = eol;
    if (le > line && le[-1] == '\r') --le;
    if (le == line || *line == '#') continue;
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
    if ((start >= 0 && pos < start) || (stop >= 0 && pos > stop)) continue;
    ++out.matched;
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
    if (gi > 255 || le - col[9] > 0x7FFFFFFF) { out.error = "record " + chrom + ":" + std::to_string(pos) + " is outside the streaming limits"; return; }
    out.off.push_back(static_cast<int64_t>(col[9] - text0));
    out.len.push_back(static_cast<int32_t>(le - col[9]));
    out.pos.push_back(static_cast<int32_t>(pos));
    out.flip.push_back(flip ? 1 : 0);
    out.gi.push_back(static_cast<uint8_t>(gi));
  }
}

}  // namespace

