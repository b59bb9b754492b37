// Host-only part of libsaihip: error text, version queries and the host side of the synth-v1
// generator.  Plain C++ (no HIP): together with vcf_ingest.cpp it is also built on its own with
// -fsanitize=address,undefined (libsaihost_san.so, __graft_entry__.build(sanitize=True)) so the code
// that parses untrusted files runs under the sanitizers in the CPU test suite.

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "saihip.h"
#include "synth_core.hpp"

namespace {

thread_local char g_err[512] = "";

}  // namespace

extern "C" int sai_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define fail sai_set_error

extern "C" {

int sai_abi_version(void) { return SAI_ABI_VERSION; }
const char* sai_build_arch(void) { return "gfx950"; }
const char* sai_last_error(void) { return g_err; }

int sai_synth_fill_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t pop_stream,
                        int32_t n_ind, int32_t ploidy, int32_t missing_per_million, int8_t* out) {
  if (n_sites < 0 || site0 < 0 || n_ind < 0 || pop_stream < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (ploidy < 1 || ploidy > 8) return fail(SAI_ERR_ARG, "ploidy must be 1..8");
  if (missing_per_million < 0 || missing_per_million > 1000000) return fail(SAI_ERR_ARG, "missing_per_million out of range");
  if (n_sites == 0 || n_ind == 0) return SAI_OK;
  if (!out) return fail(SAI_ERR_ARG, "NULL buffer");
  const uint32_t mt = miss_threshold(missing_per_million);
  for (int64_t s = 0; s < n_sites; ++s) {
    const SiteModel m = site_model(seed, chrom, site0 + s, pop_stream, ploidy);
    int8_t* row = out + s * n_ind;
    for (int32_t i = 0; i < n_ind; ++i) row[i] = synth_genotype(m, i, ploidy, mt);
  }
  return SAI_OK;
}

int sai_synth_gaps_host(uint64_t seed, int32_t chrom, int64_t site0, int64_t n_sites, int32_t* gaps) {
  if (n_sites < 0 || site0 < 0) return fail(SAI_ERR_ARG, "negative argument");
  if (n_sites > 0 && !gaps) return fail(SAI_ERR_ARG, "NULL buffer");
  for (int64_t i = 0; i < n_sites; ++i) gaps[i] = synth_gap(seed, chrom, site0 + i);
  return SAI_OK;
}

}  // extern "C"
