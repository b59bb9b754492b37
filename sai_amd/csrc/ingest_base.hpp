// Shared by the host-side ingest units (vcf_ingest.cpp: scan + block loader, vcf_stream.cpp: the
// host-inflating stream, bgzf_stream.cpp: whole BGZF members for the GPU inflate): sample selection,
// the ancestral-allele table, the tabix index, BGZF members and the block walkers over plain / gzip /
// bgzip files.  Header-only on purpose -- every unit compiles its own copy (internal linkage), so the
// three units stay independent objects of one library and of the sanitizer build.
#pragma once

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "host_threads.hpp"
#include "saihip.h"

extern "C" int sai_set_error(int code, const char* fmt, ...);  // defined in host_core.cpp


namespace {

constexpr int kScanThreads = 16;  // inflate threads of sai_vcf_scan (it has no thread argument)

struct GzReader {
  gzFile f = nullptr;
  explicit GzReader(const char* path) { f = gzopen(path, "rb"); if (f) gzbuffer(f, 1 << 20); }
  ~GzReader() { if (f) gzclose(f); }
};

inline uint32_t le32(const unsigned char* p) {
  return static_cast<uint32_t>(p[0]) | static_cast<uint32_t>(p[1]) << 8 | static_cast<uint32_t>(p[2]) << 16 |
         static_cast<uint32_t>(p[3]) << 24;
}

inline const char* find_tab(const char* p, const char* end) {
  const void* t = memchr(p, '\t', static_cast<size_t>(end - p));
  return t ? static_cast<const char*>(t) : end;
}

struct Selection {
  std::vector<int32_t> slot_of_col;  // sample column (0-based after FORMAT) -> output slot or -1
  std::vector<int32_t> ploidy;       // per output slot
  int32_t n_out = 0;
  int32_t max_col = -1;
};

// Ancestral alleles of one chromosome / region: position -> allele, as sorted arrays (a BED of a whole
// chromosome has millions of lines; a hash map of std::string took seconds to fill).  `allele` keeps
// the few map operations the readers use: find(pos), end(), size().
struct AncAllele {
  const char* p = nullptr;
  size_t n = 0;
  size_t size() const { return n; }
  const char* data() const { return p; }
};
struct AncTable {
  std::vector<int64_t> pos;   // ascending, unique
  std::vector<uint32_t> off;  // allele of entry i = text[off[i] .. off[i] + len[i])
  std::vector<uint32_t> len;
  std::string text;
  struct Hit {
    bool ok = false;
    AncAllele second;
    const Hit* operator->() const { return this; }
    bool operator==(const Hit& o) const { return ok == o.ok; }
    bool operator!=(const Hit& o) const { return ok != o.ok; }
  };
  Hit find(int64_t p) const {
    const auto it = std::lower_bound(pos.begin(), pos.end(), p);
    Hit h;
    if (it != pos.end() && *it == p) {
      const size_t i = static_cast<size_t>(it - pos.begin());
      h.ok = true;
      h.second.p = text.data() + off[i];
      h.second.n = len[i];
    }
    return h;
  }
  Hit end() const { return Hit(); }
  size_t size() const { return pos.size(); }
};
struct AncMap {
  bool active = false;
  AncTable allele;
};

struct ThreadOut {
  std::vector<int32_t> pos;
  std::vector<int8_t> dosage;
  int64_t matched = 0;  // records of the chromosome inside the region, before polarisation
  int64_t first = -1, last = -1;
  bool saw_chrom = false;        // a record line of the requested chromosome
  bool beyond_stop = false;      // ... with POS past the region
  bool last_line_other = false;  // the last record line of the piece is another chromosome
  bool failed = false;           // an exception ended the piece and not even its text could be kept
  std::string error;
};

// one-character alleles: value ('.' = -1, digits) and value after flipping |a - 1|
constexpr int kBadAllele = 64;
struct AlleleLut {
  int8_t v[256];
  int8_t f[256];
  constexpr AlleleLut() : v(), f() {
    for (int i = 0; i < 256; ++i) { v[i] = kBadAllele; f[i] = 0; }
    v[static_cast<unsigned char>('.')] = -1;
    f[static_cast<unsigned char>('.')] = 2;
    for (int dgt = 0; dgt < 10; ++dgt) {
      v['0' + dgt] = static_cast<int8_t>(dgt);
      f['0' + dgt] = static_cast<int8_t>(dgt >= 1 ? dgt - 1 : 1);
    }
  }
};
constexpr AlleleLut kAllele;

// Parse the record lines of [begin, end) (whole lines).
void parse_lines(const char* begin, const char* end, const std::string& chrom, int64_t start, int64_t stop,
                 const Selection& sel, const AncMap& anc, bool want_rows, ThreadOut& out) {
  const char* p = begin;
  std::vector<int8_t> row(static_cast<size_t>(sel.n_out)), frow(static_cast<size_t>(sel.n_out));
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
    if (out.first < 0) out.first = pos;
    out.last = pos;
    if (stop >= 0 && pos > stop) out.beyond_stop = true;
    if ((start >= 0 && pos < start) || (stop >= 0 && pos > stop)) continue;
    ++out.matched;
    if (!want_rows) continue;
    // columns: 0 CHROM 1 POS 2 ID 3 REF 4 ALT 5 QUAL 6 FILTER 7 INFO 8 FORMAT 9.. samples
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
    // FORMAT: index of GT
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
    // one forward scan over the sample columns (byte loops: the fields are 3-4 bytes long, so
    // memchr calls would cost more than they save)
    const char* s = col[9];
    for (int c = 0; c <= sel.max_col; ++c) {
      if (s > le) { out.error = "record " + chrom + ":" + std::to_string(pos) + " has too few sample columns"; return; }
      const int slot = sel.slot_of_col[static_cast<size_t>(c)];
      const char* g = s;
      if (slot >= 0) {
        for (int k = 0; k < gi; ++k) {  // skip to the GT sub-field
          while (g < le && *g != ':' && *g != '\t') ++g;
          if (g < le && *g == ':') ++g;
        }
        const int pl = sel.ploidy[static_cast<size_t>(slot)];
        int n = 0, d = 0, fd = 0;
        // fast path, branch-free in the data: a diploid call of two one-character alleles
        if (pl == 2 && g + 3 <= le) {
          const int a0 = kAllele.v[static_cast<unsigned char>(g[0])];
          const int a1 = kAllele.v[static_cast<unsigned char>(g[2])];
          const char sep = g[1];
          const char term = g + 3 < le ? g[3] : '\t';
          if (a0 != kBadAllele && a1 != kBadAllele && (sep == '|' || sep == '/') && (term == '\t' || term == ':')) {
            row[static_cast<size_t>(slot)] = static_cast<int8_t>(a0 + a1);
            frow[static_cast<size_t>(slot)] = static_cast<int8_t>(kAllele.f[static_cast<unsigned char>(g[0])] +
                                                                   kAllele.f[static_cast<unsigned char>(g[2])]);
            g += 3;
            while (g < le && *g != '\t') ++g;
            s = g + 1;
            continue;
          }
        }
        for (;;) {
          const char ch = g < le ? *g : '\t';
          int a;
          if (ch == '.') {
            a = -1;
            ++g;
          } else if (ch >= '0' && ch <= '9') {
            a = 0;
            do { a = a * 10 + (*g++ - '0'); } while (g < le && *g >= '0' && *g <= '9');
          } else if (ch == '|' || ch == '/' || ch == ':' || ch == '\t') {
            a = -1;  // empty allele
          } else {
            out.error = "unparsable genotype at " + chrom + ":" + std::to_string(pos);
            return;
          }
          if (n < pl) {  // alleles beyond the ploidy asked for are ignored
            d += a;
            fd += a >= 1 ? a - 1 : 1 - a;
            ++n;
          }
          if (g < le && (*g == '|' || *g == '/')) { ++g; continue; }
          break;
        }
        for (; n < pl; ++n) {  // fewer alleles than the ploidy asked for: padded with missing
          d -= 1;
          fd += 2;
        }
        if (d > 127 || fd > 127 || d < -128) { out.error = "dosage outside the int8 range at " + chrom + ":" + std::to_string(pos); return; }
        row[static_cast<size_t>(slot)] = static_cast<int8_t>(d);
        frow[static_cast<size_t>(slot)] = static_cast<int8_t>(fd);
      }
      while (g < le && *g != '\t') ++g;
      s = g + 1;
    }
    out.pos.push_back(static_cast<int32_t>(pos));
    const std::vector<int8_t>& src = flip ? frow : row;
    out.dosage.insert(out.dosage.end(), src.begin(), src.end());
  }
}

bool read_all_gz(const std::string& path, std::vector<unsigned char>& out);

// The BED of ancestral alleles: whitespace-separated chrom, start, pos, allele (+ anything); lines of
// other chromosomes or outside the region are skipped, a later line of a position replaces an earlier
// one, a line with one to three columns is an error (as before).  The file is read whole and parsed
// by up to 8 threads over line-aligned pieces; the entries are sorted only when the file is not.
int load_anc(const char* path, const std::string& chrom, int64_t start, int64_t stop, AncMap& anc, int64_t* n_entries) {
  std::vector<unsigned char> raw;
  {
    FILE* f = fopen(path, "rb");
    if (!f) return sai_set_error(SAI_ERR_ARG, "cannot open ancestral-allele file %s", path);
    fclose(f);
  }
  if (!read_all_gz(path, raw)) return sai_set_error(SAI_ERR_ARG, "cannot read ancestral-allele file %s", path);  // gzread passes plain text through
  const char* base = reinterpret_cast<const char*>(raw.data());
  const size_t total = raw.size();
  struct Piece {
    std::vector<int64_t> pos;
    std::vector<uint32_t> off, len;
    std::string text;
    bool short_line = false;
  };
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = static_cast<int>(std::max<size_t>(1, std::min<size_t>({size_t(8), hw ? hw : 1, total / (size_t(1) << 20) + 1})));
  std::vector<Piece> pieces(static_cast<size_t>(nt));
  std::vector<size_t> edge(static_cast<size_t>(nt) + 1, total);
  edge[0] = 0;
  for (int t = 1; t < nt; ++t) {
    size_t guess = std::max(edge[static_cast<size_t>(t) - 1], total * static_cast<size_t>(t) / static_cast<size_t>(nt));
    const void* nl = guess < total ? memchr(base + guess, '\n', total - guess) : nullptr;
    edge[static_cast<size_t>(t)] = nl ? static_cast<size_t>(static_cast<const char*>(nl) - base) + 1 : total;
  }
  auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; };
  auto work = [&](int t) {
    Piece& pc = pieces[static_cast<size_t>(t)];
    const char* p = base + edge[static_cast<size_t>(t)];
    const char* endp = base + edge[static_cast<size_t>(t) + 1];
    while (p < endp) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(endp - p)));
      if (!eol) eol = endp;
      const char* tok[4];
      size_t tlen[4];
      int found = 0;
      const char* q = p;
      while (found < 4) {
        while (q < eol && is_ws(*q)) ++q;
        if (q >= eol) break;
        const char* s0 = q;
        while (q < eol && !is_ws(*q)) ++q;
        tok[found] = s0;
        tlen[found] = static_cast<size_t>(q - s0);
        ++found;
      }
      p = eol + 1;
      if (found == 0) continue;
      if (found < 4) { pc.short_line = true; return; }
      if (tlen[0] != chrom.size() || memcmp(tok[0], chrom.data(), chrom.size()) != 0) continue;
      // strtoll's reading of the third column: optional sign, leading digits
      const char* d = tok[2];
      const char* de = tok[2] + tlen[2];
      bool neg = false;
      if (d < de && (*d == '+' || *d == '-')) neg = *d++ == '-';
      int64_t v = 0;
      while (d < de && *d >= '0' && *d <= '9') v = v * 10 + (*d++ - '0');
      if (neg) v = -v;
      if ((start >= 0 && v < start) || (stop >= 0 && v > stop)) continue;
      pc.pos.push_back(v);
      pc.off.push_back(static_cast<uint32_t>(pc.text.size()));
      pc.len.push_back(static_cast<uint32_t>(tlen[3]));
      pc.text.append(tok[3], tlen[3]);
    }
  };
  {
    ThreadGroup tg;
    for (int t = 1; t < nt; ++t) tg.spawn([&work, t] { work(t); });
    work(0);
    tg.join();
  }
  AncTable& tab = anc.allele;
  tab = AncTable();
  size_t n = 0, bytes = 0;
  for (const Piece& pc : pieces) {
    if (pc.short_line) return sai_set_error(SAI_ERR_ARG, "%s: line with fewer than 4 columns", path);
    n += pc.pos.size();
    bytes += pc.text.size();
  }
  if (bytes >= (size_t(1) << 32)) return sai_set_error(SAI_ERR_UNSUPPORTED, "%s: more than 4 GiB of alleles", path);
  tab.pos.reserve(n);
  tab.off.reserve(n);
  tab.len.reserve(n);
  tab.text.reserve(bytes);
  bool sorted = true;
  for (const Piece& pc : pieces) {
    const uint32_t shift = static_cast<uint32_t>(tab.text.size());
    for (size_t i = 0; i < pc.pos.size(); ++i) {
      if (!tab.pos.empty() && pc.pos[i] <= tab.pos.back()) {
        if (sorted && pc.pos[i] == tab.pos.back()) {  // the later line of a position wins
          tab.off.back() = pc.off[i] + shift;
          tab.len.back() = pc.len[i];
          continue;
        }
        sorted = false;
      }
      tab.pos.push_back(pc.pos[i]);
      tab.off.push_back(pc.off[i] + shift);
      tab.len.push_back(pc.len[i]);
    }
    tab.text += pc.text;
  }
  if (!sorted) {  // order by position, the LAST line of a position kept (what a map filled in file order holds)
    std::vector<uint32_t> order(tab.pos.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<uint32_t>(i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return tab.pos[x] < tab.pos[y]; });
    AncTable out;
    out.text = std::move(tab.text);
    for (size_t k = 0; k < order.size(); ++k) {
      const uint32_t i = order[k];
      if (!out.pos.empty() && out.pos.back() == tab.pos[i]) {
        out.off.back() = tab.off[i];
        out.len.back() = tab.len[i];
        continue;
      }
      out.pos.push_back(tab.pos[i]);
      out.off.push_back(tab.off[i]);
      out.len.push_back(tab.len[i]);
    }
    tab = std::move(out);
  }
  anc.active = true;
  if (n_entries) *n_entries = static_cast<int64_t>(tab.size());
  return SAI_OK;
}

// ---- tabix index (.tbi) ----------------------------------------------------------------------
// When `<vcf>.tbi` lies next to a bgzip VCF, a region load seeks to the first 16 kb window of the
// region through the index's linear part and stops at the first record past the region (an
// indexed file is sorted), and the chromosome scan reads two records instead of the file.  The
// reference gets the same effect from scikit-allel / pysam using the same index
// (utils.py:123-138, chunk_generator.py:64-73).  Virtual offset = compressed offset of a member
// << 16 | offset inside its uncompressed data.

struct TbiRef {
  bool present = false;           // the chromosome is in the index
  std::vector<uint64_t> ioff;     // linear index: first record overlapping each 16 kb window
  uint64_t first_voff = ~0ull;    // smallest / largest start of a chunk of the chromosome's bins
  uint64_t last_chunk_voff = 0;
};

bool read_all_gz(const std::string& path, std::vector<unsigned char>& out) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) return false;
  out.clear();
  unsigned char buf[1 << 16];
  int got;
  while ((got = gzread(f, buf, sizeof(buf))) > 0) out.insert(out.end(), buf, buf + got);
  gzclose(f);
  return got == 0;
}

// false: no usable index (absent, unreadable, malformed) -- the caller falls back to a full pass
bool load_tbi(const char* vcf_path, const std::string& chrom, TbiRef& ref) {
  std::vector<unsigned char> d;
  const std::string tbi_path = std::string(vcf_path) + ".tbi";
  {  // an index older than its file describes other bytes: a region would be silently cut short.
     // Whole seconds, as htslib compares them: a checkout or copy writes both files within moments
     // of each other in either order.
    struct stat sv, si;
    if (stat(vcf_path, &sv) != 0 || stat(tbi_path.c_str(), &si) != 0) return false;
    if (si.st_mtim.tv_sec < sv.st_mtim.tv_sec) return false;
  }
  if (!read_all_gz(tbi_path, d)) return false;
  size_t o = 0;
  auto need = [&](size_t n) { return o + n <= d.size(); };
  auto i32 = [&]() { const int32_t v = static_cast<int32_t>(le32(d.data() + o)); o += 4; return v; };
  auto u64 = [&]() { const uint64_t v = static_cast<uint64_t>(le32(d.data() + o)) | static_cast<uint64_t>(le32(d.data() + o + 4)) << 32; o += 8; return v; };
  if (!need(36) || memcmp(d.data(), "TBI\1", 4) != 0) return false;
  o = 4;
  const int32_t n_ref = i32();
  o += 6 * 4;  // format, col_seq, col_beg, col_end, meta, skip
  const int32_t l_nm = i32();
  if (n_ref < 0 || l_nm < 0 || !need(static_cast<size_t>(l_nm))) return false;
  std::vector<std::string> names;
  for (size_t b = o, e = o + static_cast<size_t>(l_nm); b < e;) {
    const void* z = memchr(d.data() + b, 0, e - b);
    if (!z) return false;
    names.emplace_back(reinterpret_cast<const char*>(d.data() + b));
    b = static_cast<size_t>(static_cast<const unsigned char*>(z) - d.data()) + 1;
  }
  o += static_cast<size_t>(l_nm);
  if (static_cast<int32_t>(names.size()) != n_ref) return false;
  for (int32_t r = 0; r < n_ref; ++r) {
    const bool mine = names[static_cast<size_t>(r)] == chrom;
    if (!need(4)) return false;
    const int32_t n_bin = i32();
    for (int32_t b = 0; b < n_bin; ++b) {
      if (!need(8)) return false;
      const uint32_t bin = static_cast<uint32_t>(i32());
      const int32_t n_chunk = i32();
      if (n_chunk < 0 || !need(static_cast<size_t>(n_chunk) * 16)) return false;
      for (int32_t c = 0; c < n_chunk; ++c) {
        const uint64_t beg = u64();
        u64();  // end
        if (mine && bin != 37450u) {  // 37450 is the metadata pseudo-bin
          ref.first_voff = std::min(ref.first_voff, beg);
          ref.last_chunk_voff = std::max(ref.last_chunk_voff, beg);
        }
      }
    }
    if (!need(4)) return false;
    const int32_t n_intv = i32();
    if (n_intv < 0 || !need(static_cast<size_t>(n_intv) * 8)) return false;
    if (mine) {
      ref.present = true;
      ref.ioff.resize(static_cast<size_t>(n_intv));
      for (auto& v : ref.ioff) v = u64();
    } else {
      o += static_cast<size_t>(n_intv) * 8;
    }
  }
  return true;
}

// ---- BGZF (bgzip) input ---------------------------------------------------------------------
// A bgzip file is a sequence of independent gzip members of at most 64 KiB, each carrying its
// own compressed size in a 'BC' extra subfield and its uncompressed size in the trailer: the
// members of a batch are inflated in parallel straight into their places in the output buffer.

struct BgzfMember {
  size_t data_off;   // first byte of the raw deflate stream inside the compressed buffer
  uint32_t data_len;
  uint32_t isize;    // uncompressed bytes
  uint32_t crc;
  size_t out_off;
};

// Size of the gzip member starting at p (n bytes available) when it is a BGZF member; 0 when more
// bytes are needed, -1 when it is not BGZF.
inline long bgzf_member_size(const unsigned char* p, size_t n, size_t* header_len) {
  if (n < 12) return 0;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
  const size_t xlen = static_cast<size_t>(p[10]) | static_cast<size_t>(p[11]) << 8;
  if (n < 12 + xlen) return 0;
  for (size_t o = 12; o + 4 <= 12 + xlen;) {
    const size_t slen = static_cast<size_t>(p[o + 2]) | static_cast<size_t>(p[o + 3]) << 8;
    if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
      *header_len = 12 + xlen;
      return static_cast<long>(static_cast<size_t>(p[o + 4]) | static_cast<size_t>(p[o + 5]) << 8) + 1;
    }
    o += 4 + slen;
  }
  return -1;
}

// libdeflate (same DEFLATE, 2-3x zlib 1.2.11's inflate rate) when the runtime library is on the
// machine: it ships without a header in this image, so the three entry points are bound by hand
// (their C ABI has been stable since 1.0).  SAI_NO_LIBDEFLATE=1 keeps zlib (the tests run both).
struct LibDeflate {
  void* (*alloc)() = nullptr;
  int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
  void (*release)(void*) = nullptr;
  uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
};

const LibDeflate* libdeflate() {
  static const LibDeflate lib = [] {
    LibDeflate l;
    const char* off = getenv("SAI_NO_LIBDEFLATE");
    if (off && *off && *off != '0') return l;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return l;
    l.alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
    l.decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
    l.release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
    l.crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
    if (!l.alloc || !l.decompress || !l.release || !l.crc32) l = LibDeflate();
    return l;
  }();
  return lib.decompress ? &lib : nullptr;
}

// one decompressor per worker invocation (the objects are not thread-safe)
struct Inflater {
  const LibDeflate* lib;
  void* dec = nullptr;
  Inflater() : lib(libdeflate()) { if (lib) dec = lib->alloc(); }
  ~Inflater() { if (dec) lib->release(dec); }
};

bool inflate_member(const unsigned char* src, const BgzfMember& m, char* dst, Inflater& inf) {
  if (m.isize == 0) return true;
  if (inf.dec) {
    size_t got = 0;
    const int rc = inf.lib->decompress(inf.dec, src + m.data_off, m.data_len, dst + m.out_off, m.isize, &got);
    return rc == 0 && got == m.isize && inf.lib->crc32(0, dst + m.out_off, m.isize) == m.crc;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<unsigned char*>(src + m.data_off);
  zs.avail_in = m.data_len;
  zs.next_out = reinterpret_cast<unsigned char*>(dst + m.out_off);
  zs.avail_out = m.isize;
  const int rc = inflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && zs.total_out == m.isize;
  inflateEnd(&zs);
  return ok && crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char*>(dst + m.out_off), m.isize) == m.crc;
}

// uncompressed bytes handed to the consumer at a time (SAI_VCF_BATCH_BYTES overrides it: the tests
// use a few KiB so that every carry-over path between batches runs)
inline size_t batch_out_bytes() {
  if (const char* e = getenv("SAI_VCF_BATCH_BYTES")) {
    const long long v = atoll(e);
    if (v > 0) return static_cast<size_t>(v);
  }
  return size_t(48) << 20;
}

template <typename F>
int for_each_block_bgzf(FILE* f, const char* path, int n_threads, uint64_t voff_start, size_t batch_out, F&& consume) {
  WorkerPool pool(n_threads);
  std::vector<unsigned char> cbuf(size_t(2) << 20);  // compressed bytes of a batch (grown when a batch needs more)
  std::vector<char> ubuf;
  std::vector<BgzfMember> members;
  size_t chave = 0, carry = 0;
  size_t skip = static_cast<size_t>(voff_start & 0xFFFFu);  // bytes of the first member that precede the record
  bool ceof = false;
  // batches grow from `batch_out` (default 1 MiB) to the maximum: a small indexed region or the
  // header is not charged for 48 MiB of inflating
  // 8 MiB of text per batch at most: the buffers are fresh memory and faulting them in costs more
  // than inflating into them (see for_each_block_plain); small buffers are reused batch after batch
  const size_t batch_max = std::min(batch_out_bytes(), size_t(8) << 20);
  size_t batch_now = std::min(batch_max, batch_out ? batch_out : size_t(1) << 20);
  if (fseeko(f, static_cast<off_t>(voff_start >> 16), SEEK_SET) != 0) return sai_set_error(SAI_ERR_ARG, "seek failed in %s", path);
  for (;;) {
    if (!ceof && chave < cbuf.size()) {
      const size_t got = fread(cbuf.data() + chave, 1, cbuf.size() - chave, f);
      if (got == 0) {
        if (ferror(f)) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
        ceof = true;
      }
      chave += got;
    }
    members.clear();
    size_t off = 0, out_total = 0;
    while (off < chave && out_total < batch_now) {
      size_t hlen = 0;
      const long bsize = bgzf_member_size(cbuf.data() + off, chave - off, &hlen);
      if (bsize < 0) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block", path);
      if (bsize == 0 || off + static_cast<size_t>(bsize) > chave) break;  // incomplete member
      if (static_cast<size_t>(bsize) < hlen + 8) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block", path);
      const unsigned char* tail = cbuf.data() + off + bsize - 8;
      // a BGZF member holds at most 64 KiB of data; the trailer is file content, not a promise
      if (le32(tail + 4) > 65536u) return sai_set_error(SAI_ERR_ARG, "%s: corrupt BGZF block (ISIZE > 64 KiB)", path);
      members.push_back({off + hlen, static_cast<uint32_t>(static_cast<size_t>(bsize) - hlen - 8), le32(tail + 4),
                         le32(tail), out_total});
      out_total += le32(tail + 4);
      off += static_cast<size_t>(bsize);
    }
    if (members.empty()) {
      if (ceof) {
        if (chave) return sai_set_error(SAI_ERR_ARG, "%s: truncated BGZF file", path);
        if (carry) {  // last line without a newline
          const int rc = consume(ubuf.data(), ubuf.data() + carry);
          if (rc < 0) return rc;
        }
        return SAI_OK;
      }
      if (chave == cbuf.size()) cbuf.resize(cbuf.size() * 2);
      continue;
    }
    if (ubuf.size() < carry + out_total) ubuf.resize(carry + out_total);
    {
      const int nt = std::max(1, std::min<int>(n_threads, static_cast<int>(members.size())));
      std::vector<char> bad(static_cast<size_t>(nt), 0);
      auto work = [&](int t) {  // inflate_member allocates nothing but the decompressor's own state: no throw
        const size_t lo = members.size() * static_cast<size_t>(t) / static_cast<size_t>(nt);
        const size_t hi = members.size() * static_cast<size_t>(t + 1) / static_cast<size_t>(nt);
        Inflater inf;
        for (size_t i = lo; i < hi; ++i)
          if (!inflate_member(cbuf.data(), members[i], ubuf.data() + carry, inf)) bad[static_cast<size_t>(t)] = 1;
      };
      pool.run(nt, work);
      for (char b : bad)
        if (b) return sai_set_error(SAI_ERR_ARG, "%s: BGZF block fails to inflate or its CRC", path);
    }
    batch_now = std::min(batch_max, batch_now * 2);
    memmove(cbuf.data(), cbuf.data() + off, chave - off);
    chave -= off;
    size_t have = carry + out_total;
    if (skip) {  // only ever on the first batch (carry == 0): drop what precedes the indexed record
      if (skip > have) return sai_set_error(SAI_ERR_ARG, "%s: index offset beyond its block", path);
      memmove(ubuf.data(), ubuf.data() + skip, have - skip);
      have -= skip;
      skip = 0;
    }
    size_t usable = 0;
    for (size_t i = have; i > 0; --i)
      if (ubuf[i - 1] == '\n') { usable = i; break; }
    if (usable) {
      const int rc = consume(ubuf.data(), ubuf.data() + usable);
      if (rc < 0) return rc;
      if (rc > 0) return SAI_OK;  // consumer has seen enough
    }
    carry = have - usable;
    if (carry && usable) memmove(ubuf.data(), ubuf.data() + usable, carry);
  }
}

bool file_is_bgzf(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  unsigned char head[64];
  const size_t n = fread(head, 1, sizeof(head), f);
  fclose(f);
  size_t hlen = 0;
  return bgzf_member_size(head, n, &hlen) > 0;
}

// bgzip file from virtual offset `voff` on, in batches of about `batch_out` uncompressed bytes
template <typename F>
int for_each_block_from(const char* path, int n_threads, uint64_t voff, size_t batch_out, F&& consume) {
  FILE* f = fopen(path, "rb");
  if (!f) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
  const int rc = for_each_block_bgzf(f, path, n_threads, voff, batch_out, consume);
  fclose(f);
  return rc;
}

// Uncompressed text: the batch is read by `n_threads` preads side by side (zlib's transparent
// gzread is a serial copy through its own buffer, ~1.5 GB/s; the page cache delivers far more).
bool file_is_plain_text(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  unsigned char head[2] = {0, 0};
  const size_t n = fread(head, 1, 2, f);
  fclose(f);
  return !(n == 2 && head[0] == 0x1f && head[1] == 0x8b);
}

template <typename F>
int for_each_block_plain(const char* path, int n_threads, F&& consume) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
  struct FdGuard { int fd; ~FdGuard() { close(fd); } } guard{fd};
  struct stat st;
  if (fstat(fd, &st) != 0) return sai_set_error(SAI_ERR_ARG, "cannot stat %s", path);
  const size_t total = static_cast<size_t>(st.st_size);
  // 8 MiB batches: the buffer is fresh memory, and faulting it in costs more than reading into it
  // (pread of 241 MB into new pages: 50 ms; into pages already touched: 3 ms) -- keep it small and
  // reuse it for every batch
  const size_t batch = std::min(batch_out_bytes(), size_t(8) << 20);
  std::vector<char> buf;
  size_t have = 0, file_off = 0;
  for (;;) {
    const size_t want = std::min(batch, total - file_off);
    if (buf.size() < have + want) buf.resize(have + want);
    const int nt = std::max(1, std::min<int>(n_threads, static_cast<int>(want / (size_t(4) << 20)) + 1));
    std::vector<char> bad(static_cast<size_t>(nt), 0);
    auto work = [&](int t) {  // pread loops touch only the caller's buffer: no throw
      size_t lo = want * static_cast<size_t>(t) / static_cast<size_t>(nt), hi = want * static_cast<size_t>(t + 1) / static_cast<size_t>(nt);
      while (lo < hi) {
        const ssize_t got = pread(fd, buf.data() + have + lo, hi - lo, static_cast<off_t>(file_off + lo));
        if (got <= 0) { bad[static_cast<size_t>(t)] = 1; return; }
        lo += static_cast<size_t>(got);
      }
    };
    {
      ThreadGroup th;
      for (int t = 1; t < nt; ++t) th.spawn([&work, t] { work(t); });
      work(0);
      th.join();
    }
    for (char b : bad)
      if (b) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
    have += want;
    file_off += want;
    const bool eof = file_off >= total;
    size_t usable = have;
    if (!eof) {
      usable = 0;
      for (size_t i = have; i > 0; --i)
        if (buf[i - 1] == '\n') { usable = i; break; }
      if (usable == 0) continue;  // no complete line yet: read on
    }
    if (usable) {
      const int rc = consume(buf.data(), buf.data() + usable);
      if (rc < 0) return rc;
      if (rc > 0) return SAI_OK;  // consumer has seen enough
    }
    const size_t rest = have - usable;
    if (rest) memmove(buf.data(), buf.data() + usable, rest);
    have = rest;
    if (eof) break;
  }
  return SAI_OK;
}

// Streams the file in blocks of whole lines and hands each block to `consume(begin, end)`; bgzip
// files are inflated by `n_threads` threads, plain gzip and uncompressed text go through zlib's
// gzread.  Returns 0, or a negative status after sai_set_error.
template <typename F>
int for_each_block(const char* path, int n_threads, F&& consume) {
  if (FILE* f = fopen(path, "rb")) {
    unsigned char head[64];
    const size_t n = fread(head, 1, sizeof(head), f);
    size_t hlen = 0;
    if (bgzf_member_size(head, n, &hlen) > 0) {
      const int rc = for_each_block_bgzf(f, path, n_threads, 0, 0, consume);
      fclose(f);
      return rc;
    }
    fclose(f);
  }
  if (file_is_plain_text(path)) return for_each_block_plain(path, n_threads, consume);
  GzReader r(path);
  if (!r.f) return sai_set_error(SAI_ERR_ARG, "cannot open VCF %s", path);
  std::vector<char> buf(size_t(8) << 20);
  size_t have = 0;
  for (;;) {
    if (have == buf.size()) buf.resize(buf.size() * 2);  // one line longer than the buffer
    const int got = gzread(r.f, buf.data() + have, static_cast<unsigned>(std::min<size_t>(buf.size() - have, size_t(1) << 30)));
    if (got < 0) return sai_set_error(SAI_ERR_ARG, "read error in %s", path);
    have += static_cast<size_t>(got);
    const bool eof = got == 0;
    size_t usable = have;
    if (!eof) {
      usable = 0;
      for (size_t i = have; i > 0; --i)
        if (buf[i - 1] == '\n') { usable = i; break; }
      if (usable == 0) continue;  // no complete line yet
    }
    if (usable) {
      const int rc = consume(buf.data(), buf.data() + usable);
      if (rc < 0) return rc;
      if (rc > 0) return SAI_OK;  // consumer has seen enough
    }
    const size_t rest = have - usable;
    if (rest) memmove(buf.data(), buf.data() + usable, rest);
    have = rest;
    if (eof) break;
  }
  return SAI_OK;
}

int parse_header(const char* p, const char* eol, const char* path, int32_t n_samples, const char* const* sample_names,
                 const int32_t* ploidy, Selection& sel) {
  const char* le = eol;
  if (le > p && le[-1] == '\r') --le;
  std::vector<std::string> names;
  const char* q = p;
  int c = 0;
  while (q <= le) {
    const char* t = find_tab(q, le);
    if (c >= 9) names.emplace_back(q, static_cast<size_t>(t - q));
    q = t + 1;
    ++c;
  }
  sel.slot_of_col.assign(names.size(), -1);
  sel.ploidy.assign(static_cast<size_t>(n_samples), 1);
  sel.n_out = n_samples;
  std::unordered_map<std::string, int32_t> index;
  for (size_t i = 0; i < names.size(); ++i) index.emplace(names[i], static_cast<int32_t>(i));
  for (int32_t s = 0; s < n_samples; ++s) {
    auto it = index.find(sample_names[s]);
    if (it == index.end()) return sai_set_error(SAI_ERR_ARG, "samples not found in %s: %s", path, sample_names[s]);
    if (sel.slot_of_col[static_cast<size_t>(it->second)] >= 0)
      return sai_set_error(SAI_ERR_ARG, "sample %s requested twice", sample_names[s]);
    sel.slot_of_col[static_cast<size_t>(it->second)] = s;
    sel.ploidy[static_cast<size_t>(s)] = ploidy[s];
    sel.max_col = std::max(sel.max_col, it->second);
  }
  return SAI_OK;
}

}  // namespace

// No C++ exception may cross the C ABI (the caller is ctypes: it would be std::terminate and the
// Python process would die): allocation and thread-creation failures come back as a status.
template <typename F>
static int guarded(const char* what, F&& body) {
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


