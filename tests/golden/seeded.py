"""Seeded input builders shared by make_golden.py (which feeds them to the reference) and the
tests (which feed the same arrays to the oracle / the HIP path).  Only numpy Generator methods
with a stable stream are used."""

import numpy as np


def fourpop_inputs(seed, n_sites, n_ref, n_tgt, src_sizes, n_out, ploidies, miss):
    """(ref, tgt, [src...], out or None) int64 dosage matrices; ploidies = [ref, tgt, [src...], out]."""
    rng = np.random.default_rng(seed)
    p = rng.random(n_sites) ** 2

    def block(n_ind, ploidy, shift):
        q = np.clip(p + shift * rng.random(n_sites), 0, 1)
        g = (rng.random((n_sites, n_ind, ploidy)) < q[:, None, None]).sum(axis=2).astype(np.int64)
        if miss:
            g[rng.random((n_sites, n_ind)) < miss] = -ploidy
        return g

    ref = block(n_ref, ploidies[0], -0.3)
    tgt = block(n_tgt, ploidies[1], 0.2)
    srcs = [block(n, pl, 0.5) for n, pl in zip(src_sizes, ploidies[2])]
    out = block(n_out, ploidies[3], -0.6) if n_out else None
    return ref, tgt, srcs, out
