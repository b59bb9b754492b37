// sai_inflate_bgzf: the members of a bgzip (BGZF) file inflated on the GPU.
//
// A BGZF member is an independent raw DEFLATE stream (RFC 1951) of at most 64 KiB of text, so the
// members of a batch are inflated side by side, ONE WAVEFRONT PER MEMBER.  DEFLATE is serial inside
// a stream; what the 64 lanes share is everything around the symbol decode:
//   * the compressed bytes are held 256 B at a time in one register per lane (one coalesced load,
//     the next 256 B prefetched) and fed to a wave-uniform bit buffer with v_readlane;
//   * the last 4 KiB of history live in LDS (older text is read back from HBM); a match of length L
//     is copied by L lanes at once
//     (overlapping matches read `src + k % dist`), a literal is one LDS byte store;
//   * the Huffman tables (10-bit / 8-bit primary look-up + a canonical walk for longer codes) are
//     built by all lanes: code counts with ballots, a symbol's rank inside its length with a prefix
//     popcount, every lane fills the table entries of its own symbols;
//   * finished text leaves the window for HBM 256 B at a time, one word per lane.
// Every lane executes the same decode on the same values, so nothing is broadcast or synchronised.
// (Whether the compiler keeps those values in scalar or in vector registers makes no difference to
// the rate -- measured both ways: a lone wavefront issues an instruction every few cycles either way.)
// Each access is bounded (window index masked, output against ISIZE, distance against what the
// member has produced, input against the member's length): a corrupt member ends with a non-zero
// status, never with a stray access.  A second kernel checks the CRC-32 of every member's text against
// its gzip trailer, as the host reader does after its own inflate.

#include "common.hpp"

namespace {

constexpr int kLitBits = 10;
constexpr int kDistBits = 8;
constexpr int kClBits = 7;
constexpr int kMaxLit = 288;
constexpr int kMaxDist = 32;

enum : int32_t {
  kOk = 0,
  kBadBlockType = 1,
  kBadCodeLengths = 2,
  kOutputOverrun = 3,
  kBadDistance = 4,
  kInputOverrun = 5,
  kSizeMismatch = 6,
  kBadSymbol = 7,
  kBadStored = 8,
  kCrcMismatch = 9,
};

template <int BITS, int SYMS>
struct HuffLds {
  uint16_t tab[1 << BITS];  // (len << 9) | symbol for codes of up to BITS bits, 0 otherwise
  uint16_t sorted[SYMS];    // symbols in canonical order (for the longer codes)
  uint16_t count[16];       // codes per length
};
using LitLds = HuffLds<kLitBits, kMaxLit>;
using DistLds = HuffLds<kDistBits, kMaxDist>;  // also holds the code-length code (7-bit table, 19 symbols)

// W = bytes of history kept in LDS.  The decode is a chain of dependent instructions, so its rate
// is the number of wavefronts in flight, and LDS decides that number: W = 32768 (DEFLATE's whole
// window) allows 4 per CU, 16384 allows 8, 4096 allows 16 (the register file allows no more).  A match
// that reaches further back than W is served from the text the wavefront has already written to HBM
// (one global load; a fence whenever the source is younger than the last fence).  Measured on 503 MB
// of VCF text: 36 / 65 / 77 / 91 GB/s for W = 32768 / 16384 / 8192 / 4096, and the same order on
// lines that copy their predecessor 8, 14 or 20 KB back (106 / 98 / 98 GB/s at W = 4096).
template <int W>
struct WaveLds {
  uint8_t win[W];
  LitLds lit;
  DistLds dist;
  uint8_t lens[kMaxLit + kMaxDist];  // literal/length code lengths, the distance ones right behind them
  uint8_t cl_lens[32];
};
static_assert(sizeof(WaveLds<16384>) <= 20480, "eight wavefronts per CU need <= 20 KiB each");
static_assert(sizeof(WaveLds<4096>) <= 8192, "twenty wavefronts per CU need <= 8 KiB each");

struct InflateArgs {
  const uint8_t* comp;
  int64_t n_comp;  // bytes of `comp` that may be read
  const sai_bgzf_member* members;
  int32_t n_members;
  uint8_t* text;
  int64_t n_text;  // bytes of `text` that may be written
  int32_t* status;
};

// ---- wave-uniform bit reader ---------------------------------------------------------------------
struct BitIn {
  const uint32_t* words;  // aligned base of the member's stream
  int64_t n_words;        // words that may be loaded
  uint32_t cur, nxt;      // lane l holds word 64 * chunk + l of the current / next chunk
  int widx;               // next word to enter the bit buffer
  uint64_t buf;
  int cnt;
  int lane;
  int origin_bits;  // position of the aligned base relative to the member's first byte, in bits (a member is < 2^20 bits)

  __device__ __forceinline__ uint32_t load_chunk(int chunk) const {
    const int64_t i = static_cast<int64_t>(chunk) * 64 + lane;
    return i < n_words ? words[i] : 0u;
  }
  __device__ __forceinline__ void start(const uint8_t* comp, int64_t n_comp, int64_t member_off, int64_t byte_off, int lane_) {
    lane = lane_;
    const int64_t aligned = byte_off & ~int64_t(3);
    origin_bits = static_cast<int>(aligned - member_off) * 8;
    words = reinterpret_cast<const uint32_t*>(comp + aligned);
    n_words = (n_comp - aligned) >> 2;  // the caller pads the buffer to a multiple of 4
    cur = load_chunk(0);
    nxt = load_chunk(1);
    widx = 0;
    buf = 0;
    cnt = 0;
    fill();
    const int skip = static_cast<int>(byte_off - aligned) * 8;
    buf >>= skip;
    cnt -= skip;
  }
  // at least 32 bits in the buffer afterwards
  __device__ __forceinline__ void fill() {
    while (cnt <= 32) {
      const int l = __builtin_amdgcn_readfirstlane(widx & 63);
      const uint32_t w = __builtin_amdgcn_readlane(cur, l);
      buf |= static_cast<uint64_t>(w) << cnt;
      cnt += 32;
      ++widx;
      if ((widx & 63) == 0) {
        cur = nxt;
        nxt = load_chunk((widx >> 6) + 1);
      }
    }
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return static_cast<uint32_t>(buf) & ((1u << n) - 1u); }
  __device__ __forceinline__ void drop(int n) {
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ uint32_t take(int n) {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  // bits of the member consumed so far
  __device__ __forceinline__ int pos_bits() const { return origin_bits + widx * 32 - cnt; }
};

__device__ __forceinline__ uint32_t reverse_bits(uint32_t code, int len) { return __brev(code) >> (32 - len); }

// Canonical Huffman tables of `n` code lengths (s.lens[base ...]), built by the whole wavefront.
// Returns false for an over-subscribed set of lengths.
template <int BITS, int MAXSYM, typename H>
__device__ bool build_tables(const uint8_t* lens, int n, H& h, int lane) {
  constexpr int kPer = (MAXSYM + 63) / 64;
  int my_len[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int s = k * 64 + lane;
    my_len[k] = s < n ? lens[s] : 0;
  }
  for (int i = lane; i < (1 << BITS); i += 64) h.tab[i] = 0;
  // codes per length, first code and first slot of every length (uniform), rank of my symbols
  uint32_t code = 0, slot = 0, left = 1;
  bool over = false;
  int my_code[kPer], my_slot[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) my_code[k] = my_slot[k] = 0;
  for (int len = 1; len <= 15; ++len) {
    uint32_t count = 0;
    code <<= 1;
    left <<= 1;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const bool mine = my_len[k] == len;
      const uint64_t m = __ballot(mine);
      if (mine) {
        const uint32_t rank = count + __popcll(m & ((1ull << lane) - 1ull));
        my_code[k] = static_cast<int>(code + rank);
        my_slot[k] = static_cast<int>(slot + rank);
      }
      count += __popcll(m);
    }
    if (count > left) over = true;
    left -= count > left ? left : count;
    if (lane == 0) h.count[len] = static_cast<uint16_t>(count);
    code += count;
    slot += count;
  }
  if (over) return false;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int len = my_len[k];
    if (len == 0) continue;
    const int s = k * 64 + lane;
    h.sorted[my_slot[k]] = static_cast<uint16_t>(s);
    if (len <= BITS) {
      const uint32_t rev = reverse_bits(static_cast<uint32_t>(my_code[k]), len);
      const uint16_t e = static_cast<uint16_t>((len << 9) | s);
      for (uint32_t i = rev; i < (1u << BITS); i += 1u << len) h.tab[i] = e;
    }
  }
  __syncthreads();  // one wavefront per workgroup: orders the LDS writes before the look-ups
  return true;
}

// The primary-table entry of the code at the head of the bit buffer (0 = a longer code).  Every lane
// reads the same entry: readfirstlane hands it to the scalar unit, so that the symbol arithmetic
// and the branches of the decode loop run there.  Only peeks: the look-up can be issued early.
template <int BITS, typename H>
__device__ __forceinline__ uint32_t peek_entry(const BitIn& in, const H& h) {
  return __builtin_amdgcn_readfirstlane(h.tab[in.peek(BITS)]);
}

// The symbol of entry `e` (consumes its bits); -1 for a bit pattern that is no code.  Needs >= 15
// bits in the buffer.
template <typename H>
__device__ __forceinline__ int take_symbol(BitIn& in, const H& h, uint32_t e) {
  if (e) {
    in.drop(e >> 9);
    return e & 511;
  }
  // codes longer than the primary table: walk the canonical code one bit at a time (rare symbols)
  int code = 0, first = 0, index = 0;
  uint32_t bits = static_cast<uint32_t>(in.buf);
  for (int len = 1; len <= 15; ++len) {
    code |= static_cast<int>(bits & 1u);
    bits >>= 1;
    const int count = __builtin_amdgcn_readfirstlane(h.count[len]);
    if (code - count < first) {
      in.drop(len);
      return __builtin_amdgcn_readfirstlane(h.sorted[index + (code - first)]);
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

template <int BITS, typename H>
__device__ __forceinline__ int decode_symbol(BitIn& in, const H& h) {
  return take_symbol(in, h, peek_entry<BITS>(in, h));
}

template <int W>
__global__ __launch_bounds__(64) void inflate_bgzf_kernel(InflateArgs a) {
  constexpr int kWin = W;
  __shared__ WaveLds<W> s;
  const int lane = threadIdx.x;
  const int m = blockIdx.x;
  if (m >= a.n_members) return;
  const sai_bgzf_member mem = a.members[m];
  const int isize = static_cast<int>(mem.isize);
  int32_t err = kOk;
  // the member's own bounds inside the two buffers (checked on the host as well)
  if (mem.data_off < 0 || mem.data_off + static_cast<int64_t>(mem.data_len) > a.n_comp || mem.out_off < 0 ||
      mem.isize > 65536u || mem.data_len > (1u << 17) || mem.out_off + static_cast<int64_t>(mem.isize) > a.n_text) {
    if (lane == 0) a.status[m] = kSizeMismatch;
    return;
  }
  uint8_t* dst = a.text + mem.out_off;
  const bool dst_aligned = (reinterpret_cast<uintptr_t>(dst) & 3u) == 0;
  BitIn in;
  in.start(a.comp, a.n_comp, mem.data_off, mem.data_off, lane);
  const int bit_limit = static_cast<int>(mem.data_len) * 8;
  int out_pos = 0, flushed = 0, fenced = 0;  // text produced / stored to HBM / known to be visible there
  bool last = false;

  auto flush_groups = [&]() {
    while (out_pos - flushed >= 256) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(&s.win[(flushed + 4 * lane) & (kWin - 1)]);
      uint8_t* d = dst + flushed + 4 * lane;
      if (dst_aligned) {
        *reinterpret_cast<uint32_t*>(d) = w;
      } else {
        d[0] = static_cast<uint8_t>(w);
        d[1] = static_cast<uint8_t>(w >> 8);
        d[2] = static_cast<uint8_t>(w >> 16);
        d[3] = static_cast<uint8_t>(w >> 24);
      }
      flushed += 256;
    }
  };

  while (!last && err == kOk) {
    in.fill();
    if (in.pos_bits() > bit_limit) { err = kInputOverrun; break; }
    last = in.take(1) != 0;
    const uint32_t type = in.take(2);
    if (type == 3) { err = kBadBlockType; break; }
    if (type == 0) {
      // stored: LEN / NLEN on the next byte boundary, then LEN raw bytes
      in.drop(in.cnt & 7);
      in.fill();
      const uint32_t len = in.take(16);
      in.fill();
      const uint32_t nlen = in.take(16);
      if ((len ^ nlen) != 0xFFFFu) { err = kBadStored; break; }
      const int64_t byte0 = mem.data_off + (in.pos_bits() >> 3);  // absolute offset of the raw bytes
      if (byte0 + len > mem.data_off + static_cast<int64_t>(mem.data_len)) { err = kInputOverrun; break; }
      if (out_pos + len > isize) { err = kOutputOverrun; break; }
      for (uint32_t done = 0; done < len;) {  // through the window in pieces that fit it
        const uint32_t step = len - done < static_cast<uint32_t>(W / 2) ? len - done : static_cast<uint32_t>(W / 2);
        for (uint32_t k = lane; k < step; k += 64) s.win[(out_pos + k) & (kWin - 1)] = a.comp[byte0 + done + k];
        __syncthreads();
        out_pos += step;
        done += step;
        flush_groups();
      }
      in.start(a.comp, a.n_comp, mem.data_off, byte0 + len, lane);
      continue;
    }
    int n_lit, n_dist;
    if (type == 1) {
      n_lit = 288;
      n_dist = 32;
      for (int i = lane; i < 288; i += 64) s.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
      if (lane < 32) s.lens[288 + lane] = 5;
      __syncthreads();
    } else {
      n_lit = static_cast<int>(in.take(5)) + 257;
      n_dist = static_cast<int>(in.take(5)) + 1;
      const int n_cl = static_cast<int>(in.take(4)) + 4;
      if (n_lit > 286 || n_dist > 30) { err = kBadCodeLengths; break; }
      // the code-length code: 3 bits per length, in the permuted order of RFC 1951 3.2.7
      if (lane < 32) s.cl_lens[lane] = 0;
      __syncthreads();
      for (int i = 0; i < n_cl; ++i) {
        in.fill();
        const uint32_t v = in.take(3);
        // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const int pos = i < 3 ? 16 + i : (i == 3 ? 0 : ((i & 1) ? (19 - i) >> 1 : 6 + (i >> 1)));
        if (lane == 0) s.cl_lens[pos] = static_cast<uint8_t>(v);
      }
      __syncthreads();
      if (!build_tables<kClBits, 64>(s.cl_lens, 19, s.dist, lane)) { err = kBadCodeLengths; break; }
      // the literal/length and distance code lengths, run-length coded with the code-length code
      int i = 0, prev = 0;
      const int total = n_lit + n_dist;
      while (i < total) {
        in.fill();
        const int sym = decode_symbol<kClBits>(in, s.dist);
        if (sym < 0 || sym > 18) { err = kBadCodeLengths; break; }
        if (sym < 16) {
          if (lane == 0) s.lens[i] = static_cast<uint8_t>(sym);
          prev = sym;
          ++i;
          continue;
        }
        int rep, val = 0;
        if (sym == 16) {
          if (i == 0) { err = kBadCodeLengths; break; }
          val = prev;
          rep = 3 + static_cast<int>(in.take(2));
        } else if (sym == 17) {
          rep = 3 + static_cast<int>(in.take(3));
        } else {
          rep = 11 + static_cast<int>(in.take(7));
        }
        if (i + rep > total) { err = kBadCodeLengths; break; }
        for (int k = lane; k < rep; k += 64) s.lens[i + k] = static_cast<uint8_t>(val);
        i += rep;
        prev = val;
      }
      if (err != kOk) break;
      __syncthreads();
      if (s.lens[256] == 0) { err = kBadCodeLengths; break; }  // no end-of-block code
    }
    if (!build_tables<kLitBits, kMaxLit>(s.lens, n_lit, s.lit, lane)) { err = kBadCodeLengths; break; }
    if (!build_tables<kDistBits, 64>(s.lens + n_lit, n_dist, s.dist, lane)) { err = kBadCodeLengths; break; }

    // ---- the symbols of the block --------------------------------------------------------------
    // No input check per symbol: every symbol either produces text (bounded by ISIZE) or ends the
    // block, and the next block header is checked against the member's length -- a stream that runs
    // off its end into the bytes of the next member stops there at the latest.
    // `e` = the table entry of the NEXT symbol, looked up while the copy of the current match is
    // still on its way through the LDS (the look-up only peeks at the bits)
    in.fill();
    uint32_t e = peek_entry<kLitBits>(in, s.lit);
    for (;;) {
      int sym = take_symbol(in, s.lit, e);
      if (sym < 0) { err = kBadSymbol; break; }
      if (sym < 256) {
        if (out_pos >= isize) { err = kOutputOverrun; break; }
        s.win[out_pos & (kWin - 1)] = static_cast<uint8_t>(sym);  // all lanes, one address: no exec-mask juggling
        ++out_pos;
        in.fill();
        e = peek_entry<kLitBits>(in, s.lit);
      } else if (sym == 256) {
        break;
      } else {
        sym -= 257;
        if (sym >= 29) { err = kBadSymbol; break; }
        int len;
        if (sym < 8) {
          len = 3 + sym;
        } else if (sym == 28) {
          len = 258;
        } else {
          const int x = (sym - 4) >> 2;
          len = 3 + ((4 + (sym & 3)) << x) + static_cast<int>(in.take(x));
        }
        in.fill();
        const int dsym = decode_symbol<kDistBits>(in, s.dist);
        if (dsym < 0 || dsym >= 30) { err = kBadSymbol; break; }
        int dist;
        if (dsym < 4) {
          dist = 1 + dsym;
        } else {
          const int x = (dsym >> 1) - 1;
          dist = 1 + ((2 + (dsym & 1)) << x) + static_cast<int>(in.take(x));
        }
        if (dist > out_pos) { err = kBadDistance; break; }
        if (out_pos + len > isize) { err = kOutputOverrun; break; }
        const int src = out_pos - dist;
        if (W < 32768 && dist > W) {
          // further back than the LDS history: the text is in HBM already (everything but the last
          // few hundred bytes is); a fence whenever the source is younger than the last one makes this wavefront's own stores
          // visible to all its lanes
          if (src + len > fenced) {
            __threadfence();
            fenced = flushed;
          }
          uint8_t b[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int k = j * 64 + lane;
            b[j] = k < len ? __hip_atomic_load(dst + src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : uint8_t(0);
          }
          in.fill();
          e = peek_entry<kLitBits>(in, s.lit);
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int k = j * 64 + lane;
            if (k < len) s.win[(out_pos + k) & (kWin - 1)] = b[j];
          }
          out_pos += len;
          if (out_pos - flushed >= 256) {
            __syncthreads();
            flush_groups();
          }
          continue;
        }
        // first 64 bytes of the match: loads, then the next symbol's look-up, then the stores
        // (a match that overlaps its own output repeats with period `dist`)
        int k0 = lane;
        if (dist < len) k0 = dist == 1 ? 0 : lane % dist;  // uniform branch; the division only where the text repeats
        // The history window is shared by the lanes of this one wavefront: a byte stored by one lane (a
        // literal, an earlier match) is read here by another.  The hardware executes a wavefront's LDS
        // operations in order; the wavefront-scope fence states that order to the compiler as well (no
        // instruction is emitted for it) instead of leaving it to what alias analysis cannot prove.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t b0 = 0;
        if (lane < len) b0 = s.win[(src + k0) & (kWin - 1)];
        in.fill();
        e = peek_entry<kLitBits>(in, s.lit);
        if (lane < len) s.win[(out_pos + lane) & (kWin - 1)] = b0;
        if (len > 64) {
          uint8_t b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = (j + 1) * 64 + lane;
            b[j] = k < len ? s.win[(src + (dist >= len ? k : k % dist)) & (kWin - 1)] : 0;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = (j + 1) * 64 + lane;
            if (k < len) s.win[(out_pos + k) & (kWin - 1)] = b[j];
          }
        }
        out_pos += len;
      }
      if (out_pos - flushed >= 256) {
        __syncthreads();
        flush_groups();
      }
    }
  }
  if (err == kOk) {
    if (out_pos != isize) err = kSizeMismatch;
    else if (in.pos_bits() > bit_limit) err = kInputOverrun;
  }
  if (err == kOk) {
    __syncthreads();
    flush_groups();
    for (int k = flushed + lane; k < out_pos; k += 64) dst[k] = s.win[k & (kWin - 1)];
  }
  if (lane == 0) a.status[m] = err;
}

// ---- CRC-32 of every member's text ---------------------------------------------------------------
// One wavefront per member again, but this part is parallel inside a member: every lane takes a
// contiguous slice of the text (just written, so it comes from L2), runs the table-driven CRC over it
// eight bytes per step (slicing-by-8, tables in LDS), and the 64 slice CRCs are combined with the
// GF(2) algebra of zlib's crc32_combine: crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B).

constexpr uint32_t kCrcPoly = 0xEDB88320u;

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__host__ __device__ constexpr uint32_t gf2_multmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (uint32_t m = 1u << 31; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

struct X2nTable {
  uint32_t v[32];  // x^(2^k) mod P
  constexpr X2nTable() : v() {
    uint32_t p = 1u << 30;  // x^1
    v[0] = p;
    for (int k = 1; k < 32; ++k) v[k] = p = gf2_multmod(p, p);
  }
};
__constant__ const X2nTable kX2n;

// x^(8 n) mod P
__device__ inline uint32_t gf2_shift_op(uint32_t n_bytes) {
  uint32_t p = 1u << 31;  // x^0
  for (int k = 3; n_bytes; n_bytes >>= 1, ++k)
    if (n_bytes & 1u) p = gf2_multmod(kX2n.v[k & 31], p);
  return p;
}

__global__ __launch_bounds__(64) void crc_members_kernel(InflateArgs a) {
  __shared__ uint32_t tab[8][256];  // slicing-by-8: eight bytes per step, the eight look-ups independent
  const int lane = threadIdx.x;
  const int m = blockIdx.x;
  if (m >= a.n_members) return;
  if (a.status[m] != kOk) return;  // uniform: nothing sensible to check
  const sai_bgzf_member mem = a.members[m];
  for (int i = lane; i < 256; i += 64) {
    uint32_t c = static_cast<uint32_t>(i);
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
    tab[0][i] = c;
  }
  __syncthreads();
  for (int t = 1; t < 8; ++t) {
    for (int i = lane; i < 256; i += 64) {
      const uint32_t c = tab[t - 1][i];
      tab[t][i] = (c >> 8) ^ tab[0][c & 0xFFu];
    }
    __syncthreads();
  }
  const uint8_t* p0 = a.text + mem.out_off;
  const uint8_t* p1 = p0 + mem.isize;
  // slices on the 8-byte grid of the address space: only the first and the last one have ragged ends
  const uintptr_t base = reinterpret_cast<uintptr_t>(p0) & ~uintptr_t(7);
  const uint32_t span = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p1) - base);
  const uint32_t slice = ((span + 63u) / 64u + 7u) & ~7u;
  const uint8_t* lo = reinterpret_cast<const uint8_t*>(base) + static_cast<size_t>(lane) * slice;
  const uint8_t* hi = lo + slice;
  if (lo < p0) lo = p0;
  if (hi > p1) hi = p1;
  uint32_t crc = 0;
  uint32_t after = 0;
  if (lo < hi) {
    uint32_t c = 0xFFFFFFFFu;
    const uint8_t* q = lo;
    while (q < hi && (reinterpret_cast<uintptr_t>(q) & 7u)) c = tab[0][(c ^ *q++) & 0xFFu] ^ (c >> 8);
    for (; q + 8 <= hi; q += 8) {
      const u32x2 w = *reinterpret_cast<const u32x2*>(q);
      const uint32_t x = c ^ w.x, y = w.y;
      c = tab[7][x & 0xFFu] ^ tab[6][(x >> 8) & 0xFFu] ^ tab[5][(x >> 16) & 0xFFu] ^ tab[4][x >> 24] ^
          tab[3][y & 0xFFu] ^ tab[2][(y >> 8) & 0xFFu] ^ tab[1][(y >> 16) & 0xFFu] ^ tab[0][y >> 24];
    }
    while (q < hi) c = tab[0][(c ^ *q++) & 0xFFu] ^ (c >> 8);
    crc = c ^ 0xFFFFFFFFu;
    after = static_cast<uint32_t>(p1 - hi);
  }
  // shift every slice CRC by the bytes behind it and fold (an empty slice contributes nothing)
  uint32_t part = (lo < hi) ? gf2_multmod(gf2_shift_op(after), crc) : 0u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part ^= __shfl_xor(part, off);
  if (lane == 0 && part != mem.crc) a.status[m] = kCrcMismatch;
}

}  // namespace

extern "C" int sai_inflate_bgzf(sai_ctx* ctx, const void* comp, int64_t n_comp_bytes, const sai_bgzf_member* members,
                                int32_t n_members, void* text, int64_t n_text_bytes, int32_t* status, void* stream) {
  if (int rc = enter(ctx)) return rc;
  if (n_members < 0 || n_comp_bytes < 0 || n_text_bytes < 0) return fail(SAI_ERR_ARG, "negative size");
  if (n_members == 0) return SAI_OK;
  if (!comp || !members || !text || !status) return fail(SAI_ERR_ARG, "NULL buffer");
  if ((reinterpret_cast<uintptr_t>(comp) & 3u) || (n_comp_bytes & 3))
    return fail(SAI_ERR_ARG, "the compressed buffer must be 4-byte aligned and padded to a multiple of 4 bytes");
  InflateArgs a;
  a.comp = static_cast<const uint8_t*>(comp);
  a.n_comp = n_comp_bytes;
  a.members = members;
  a.n_members = n_members;
  a.text = static_cast<uint8_t*>(text);
  a.n_text = n_text_bytes;
  a.status = status;
  static const int window = [] {  // SAI_INFLATE_WINDOW = 8192 / 16384 / 32768: more history in LDS (for A/B runs)
    const char* e = std::getenv("SAI_INFLATE_WINDOW");
    const int v = e ? std::atoi(e) : 0;
    return v == 32768 || v == 16384 || v == 8192 ? v : 4096;
  }();
  const dim3 grid(static_cast<unsigned>(n_members));
  if (window == 4096) hipLaunchKernelGGL(inflate_bgzf_kernel<4096>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a);
  else if (window == 8192) hipLaunchKernelGGL(inflate_bgzf_kernel<8192>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a);
  else if (window == 32768) hipLaunchKernelGGL(inflate_bgzf_kernel<32768>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(inflate_bgzf_kernel<16384>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a);
  if (int rc = check_launch("inflate_bgzf")) return rc;
  hipLaunchKernelGGL(crc_members_kernel, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("crc_members");
}
