"""numpy statement of the "synth-v1" generator -- TEST INFRASTRUCTURE ONLY.

The generator is this repo's own, not the reference's; this file is the
independent definition the HIP/host implementation in sai_amd/csrc/synth.hip is tested
against.  All arithmetic is uint64 wrap-around / IEEE f64.

  mix64            splitmix64 finaliser
  stream_key       mix64(seed ^ chrom << 40 ^ stream << 32); streams: 0 gaps, 1 site class,
                   2 + p genotypes of population stream p (0 ref, 1 tgt, 2.. sources)
  gap(i)           1 + (hi32(mix64(key0 + i)) mod 49);  pos = cumsum(gap)
  site class       hs = mix64(key1 + i); introgressed iff hi32(hs) mod 1000 == 0;
                   u = lo32(hs) / 2^32
  allele prob p    introgressed: ref 0 (dosage fixed 0), tgt 0.2 + 0.7u, sources dosage = ploidy;
                   background: (u*u)*(u*u) for every population
  genotype         site key k = mix64(key_{2+p} + i); per pair of alleles a hash
                   h = mix64(k + ind + (pair << 32)); allele ALT iff lo32(h) < T (then hi32(h)),
                   T = trunc(p * 2^32)
  missing          hm = mix64(k ^ 0xD1B54A32D192ED03 ^ (ind << 1)); missing (-ploidy) iff
                   hi32(hm) < trunc(rate_per_million * 2^32 / 1e6)
"""

from __future__ import annotations

import numpy as np

U64 = np.uint64
_M = (1 << 64) - 1


def mix64(z):
    z = np.asarray(z, dtype=U64)
    with np.errstate(over="ignore"):
        z = z + U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


def stream_key(seed, chrom, stream):
    return mix64(U64((int(seed) ^ (int(chrom) << 40) ^ (int(stream) << 32)) & _M))


def gaps(seed, chrom, site0, n_sites):
    i = np.arange(site0, site0 + n_sites, dtype=U64)
    with np.errstate(over="ignore"):
        h = mix64(stream_key(seed, chrom, 0) + i)
    return (1 + ((h >> U64(32)) % U64(49))).astype(np.int32)


def genotypes(seed, chrom, site0, n_sites, pop_stream, n_ind, ploidy, missing_per_million=0):
    """int8 [n_sites][n_ind] in reference order."""
    i = np.arange(site0, site0 + n_sites, dtype=U64)
    with np.errstate(over="ignore"):
        hs = mix64(stream_key(seed, chrom, 1) + i)
        key = mix64(stream_key(seed, chrom, 2 + pop_stream) + i)
    intro = ((hs >> U64(32)) % U64(1000)) == 0
    u = (hs & U64(0xFFFFFFFF)).astype(np.float64) * (1.0 / 4294967296.0)
    u2 = u * u
    p = u2 * u2
    if pop_stream == 1:
        p = np.where(intro, 0.2 + 0.7 * u, p)
    thr = np.minimum(p * 4294967296.0, 4294967295.0).astype(U64)
    ind = np.arange(n_ind, dtype=U64)[None, :]
    d = np.zeros((n_sites, n_ind), dtype=np.int64)
    for a in range(0, ploidy, 2):
        with np.errstate(over="ignore"):
            h = mix64(key[:, None] + ind + U64((a >> 1) << 32))
        d += (h & U64(0xFFFFFFFF)) < thr[:, None]
        if a + 1 < ploidy:
            d += (h >> U64(32)) < thr[:, None]
    if pop_stream == 0:
        d[intro] = 0
    elif pop_stream >= 2:
        d[intro] = ploidy
    if missing_per_million:
        mt = U64((int(missing_per_million) << 32) // 1000000)
        hm = mix64(key[:, None] ^ U64(0xD1B54A32D192ED03) ^ (ind << U64(1)))
        d[(hm >> U64(32)) < mt] = -ploidy
    return d.astype(np.int8)
