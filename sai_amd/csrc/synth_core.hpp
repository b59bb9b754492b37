// synth-v1: the counter-based synthetic data generator, shared by the device
// kernels (synth.hip) and the host entry points (host_core.cpp): every byte is a pure function of
// (seed, chromosome, site, population stream, individual), identical on host and device.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define SAI_HD __host__ __device__ __forceinline__
#else
#define SAI_HD inline
#endif

SAI_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

SAI_HD uint64_t stream_key(uint64_t seed, int32_t chrom, int32_t stream) {
  return mix64(seed ^ (static_cast<uint64_t>(static_cast<uint32_t>(chrom)) << 40) ^
               (static_cast<uint64_t>(static_cast<uint32_t>(stream)) << 32));
}

SAI_HD int32_t synth_gap(uint64_t seed, int32_t chrom, int64_t site) {
  const uint64_t h = mix64(stream_key(seed, chrom, 0) + static_cast<uint64_t>(site));
  return 1 + static_cast<int32_t>(static_cast<uint32_t>(h >> 32) % 49u);
}

struct SiteModel {
  uint64_t key;        // per (site, population stream) hashing key
  uint32_t threshold;  // allele is ALT when a 32-bit uniform < threshold
  int32_t fixed;       // -1: draw; else dosage forced to this value
};

// pop_stream: 0 = ref, 1 = tgt, >= 2 = sources
SAI_HD SiteModel site_model(uint64_t seed, int32_t chrom, int64_t site, int32_t pop_stream, int32_t ploidy) {
  const uint64_t hs = mix64(stream_key(seed, chrom, 1) + static_cast<uint64_t>(site));
  const bool intro = (static_cast<uint32_t>(hs >> 32) % 1000u) == 0u;
  const double u = static_cast<double>(static_cast<uint32_t>(hs)) * (1.0 / 4294967296.0);
  SiteModel m;
  m.key = mix64(stream_key(seed, chrom, 2 + pop_stream) + static_cast<uint64_t>(site));
  m.fixed = -1;
  double p;
  if (intro) {
    if (pop_stream == 0) { m.fixed = 0; p = 0.0; }
    else if (pop_stream == 1) { p = 0.2 + 0.7 * u; }
    else { m.fixed = ploidy; p = 1.0; }
  } else {
    const double u2 = u * u;
    p = u2 * u2;
  }
  m.threshold = static_cast<uint32_t>(p * 4294967296.0 >= 4294967295.0 ? 4294967295.0 : p * 4294967296.0);
  return m;
}

SAI_HD int8_t synth_genotype(const SiteModel& m, int32_t ind, int32_t ploidy, uint32_t miss_thr) {
  if (miss_thr != 0u) {
    const uint64_t hm = mix64(m.key ^ 0xD1B54A32D192ED03ull ^ (static_cast<uint64_t>(static_cast<uint32_t>(ind)) << 1));
    if (static_cast<uint32_t>(hm >> 32) < miss_thr) return static_cast<int8_t>(-ploidy);
  }
  if (m.fixed >= 0) return static_cast<int8_t>(m.fixed);
  int d = 0;
  for (int a = 0; a < ploidy; a += 2) {
    const uint64_t h = mix64(m.key + static_cast<uint64_t>(static_cast<uint32_t>(ind)) +
                             (static_cast<uint64_t>(a >> 1) << 32));
    d += static_cast<uint32_t>(h) < m.threshold;
    if (a + 1 < ploidy) d += static_cast<uint32_t>(h >> 32) < m.threshold;
  }
  return static_cast<int8_t>(d);
}

SAI_HD uint32_t miss_threshold(int32_t missing_per_million) {
  return static_cast<uint32_t>((static_cast<uint64_t>(missing_per_million) << 32) / 1000000ull);
}
