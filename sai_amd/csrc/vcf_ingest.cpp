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

namespace {

}  // namespace

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
  // ends with the member that holds the first record beyond it (-1: to the end of the file)
  int64_t file_begin = 0, file_stop = -1;
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
  bool region_end = false;
  WorkerPool readers(std::max(1, std::min(st->n_threads, 8)));
  while (file_off < total && !region_end) {
    int b;
    {
      std::unique_lock<std::mutex> lk(st->m);
      b = static_cast<int>(st->produced % 2);
      st->cv.wait(lk, [&] { return st->state[b] == 0 || st->cancel; });
      if (st->cancel) return SAI_OK;
    }
    unsigned char* dst = st->bufs[b];
    size_t want = std::min(st->cap - 8, total - file_off);  // 8 bytes of zero padding behind the data
    if (st->file_stop >= 0)  // a region: up to its last member (a member is < 64 KiB + header), not the whole buffer
      want = std::min(want, static_cast<size_t>(st->file_stop) + (size_t(1) << 17) > file_off
                                ? static_cast<size_t>(st->file_stop) + (size_t(1) << 17) - file_off : size_t(0));
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
      const bool last_of_region = st->file_stop >= 0 && file_off + off >= static_cast<size_t>(st->file_stop);
      off += static_cast<size_t>(bsize);
      if (last_of_region) {
        region_end = true;
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

struct sai_vcf_block {
  int32_t n_samples = 0;
  int64_t n_matched = 0;      // records of the chromosome inside the region
  int64_t n_anc_entries = 0;  // ancestral-allele entries loaded for the chromosome/region
  std::vector<int32_t> pos;
  std::vector<int8_t> dosage;  // [record][sample]
};

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
    // starts last -- two short reads instead of the whole file
    if (idx.present && idx.first_voff != ~0ull) {
      int64_t lo = -1;
      int rc = for_each_block_from(path, 1, idx.first_voff, size_t(1) << 16, [&](const char* p, const char* end) -> int {
        const int r = scan(p, end);
        return (r != 0 || first >= 0) ? 1 : 0;
      });
      if (rc) return rc;
      lo = first;
      first = -1;
      rc = for_each_block_from(path, kScanThreads, idx.last_chunk_voff, size_t(4) << 20, scan);
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
      // always, so a coarse or stale bound costs bytes, never records: an entry equal to the window's
      // own (an index that fills empty windows from the previous one) is not taken as the end.
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

int sai_vcf_block_free(sai_vcf_block* block) {
  delete block;
  return SAI_OK;
}

}  // extern "C"
