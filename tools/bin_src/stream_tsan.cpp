// ThreadSanitizer driver for the host-side readers (not part of the library): streams a VCF through
// sai_vcf_stream_* with small staging buffers (many hand-overs between producer and consumer, many
// runs of the worker pool), loads it with sai_vcf_load, scans it, and closes one stream early.
//   g++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude tools/bin_src/stream_tsan.cpp \
//       sai_amd/csrc/host_core.cpp sai_amd/csrc/vcf_ingest.cpp sai_amd/csrc/vcf_stream.cpp sai_amd/csrc/bgzf_stream.cpp \
//       sai_amd/csrc/narrow.cpp -lz -lpthread -ldl -o /tmp/stream_tsan
//   /tmp/stream_tsan file.vcf[.gz] chrom sample [sample...]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "saihip.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const char* path = argv[1];
  const char* chrom = argv[2];
  std::vector<const char*> names(argv + 3, argv + argc);
  std::vector<int32_t> ploidy(names.size(), 2);
  const int64_t cap = 1 << 16;
  std::vector<char> b0(cap), b1(cap);
  long long lines = 0, bytes = 0;
  for (int round = 0; round < 3; ++round) {
    sai_vcf_stream* st = nullptr;
    if (sai_vcf_stream_open(path, chrom, -1, -1, (int32_t)names.size(), names.data(), ploidy.data(), nullptr, 6, b0.data(),
                            b1.data(), cap, &st)) {
      std::fprintf(stderr, "open: %s\n", sai_last_error());
      return 1;
    }
    for (int k = 0;; ++k) {
      int32_t buf = 0, done = 0;
      int64_t nb = 0, nl = 0;
      const int64_t* off;
      const int32_t *len, *pos;
      const uint8_t *flip, *gi;
      if (sai_vcf_stream_next(st, &buf, &nb, &nl, &off, &len, &pos, &flip, &gi, &done)) {
        std::fprintf(stderr, "next: %s\n", sai_last_error());
        return 1;
      }
      if (done) break;
      const char* text = buf == 0 ? b0.data() : b1.data();
      for (int64_t i = 0; i < nl; ++i) bytes += text[off[i]] + len[i] + pos[i] + flip[i] + gi[i];  // touch what the producer wrote
      lines += nl;
      if (round == 2 && k == 1) break;  // give up early: close() must stop and join the producer
    }
    sai_vcf_stream_close(st);
  }
  sai_vcf_block* blk = nullptr;
  if (sai_vcf_load(path, chrom, -1, -1, (int32_t)names.size(), names.data(), ploidy.data(), nullptr, 6, &blk)) return 1;
  int64_t n_rec = 0, n_match = 0, n_anc = 0;
  sai_vcf_block_info(blk, &n_rec, &n_match, &n_anc);
  sai_vcf_block_free(blk);
  int64_t first = 0, last = 0;
  if (sai_vcf_scan(path, chrom, &first, &last)) return 1;
  std::printf("lines %lld (checksum %lld), loaded %lld records, scan %lld..%lld\n", lines, bytes, (long long)n_rec, (long long)first,
              (long long)last);
  return 0;
}
