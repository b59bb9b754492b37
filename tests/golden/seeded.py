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


def fuzz_scenario(seed):
    """A seeded random chromosome + config: several ref / tgt populations, 1-2 sources, optional
    outgroup, ploidy 1-4, missing calls, random window grid, optional chunk bounds."""
    rng = np.random.default_rng(seed)
    n_sites = int(rng.integers(200, 2500))
    pos = np.cumsum(rng.integers(1, 60, n_sites)).astype(np.int32)
    n_ref, n_tgt, n_src = int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 3))
    with_out = bool(rng.random() < 0.5)
    anc = True if with_out else bool(rng.random() < 0.5)
    p = rng.random(n_sites) ** float(rng.choice([1, 2, 4]))

    def pop(n_ind, ploidy, fixed=False):
        g = rng.binomial(ploidy, np.broadcast_to(p[:, None], (n_sites, n_ind))).astype(np.int64)
        if fixed:
            g[rng.random(n_sites) < 0.25] = ploidy
        miss = rng.random(g.shape) < float(rng.choice([0.0, 0.02, 0.2]))
        g[miss] = -ploidy
        return g

    pl = {"ref": {}, "tgt": {}, "src": {}}
    gts = {"ref": {}, "tgt": {}, "src": {}, "outgroup": {}}
    for grp, n, prefix in (("ref", n_ref, "R"), ("tgt", n_tgt, "T"), ("src", n_src, "S")):
        for i in range(n):
            ploidy = int(rng.integers(1, 5))
            pl[grp][f"{prefix}{i}"] = ploidy
            gts[grp][f"{prefix}{i}"] = pop(int(rng.integers(1, 40 if grp != "src" else 4)), ploidy, fixed=grp == "src")
    if with_out:
        pl["outgroup"] = {"O": int(rng.integers(1, 3))}
        gts["outgroup"]["O"] = pop(int(rng.integers(1, 6)), pl["outgroup"]["O"])
    ops = ["=", "<", ">", "<=", ">="]

    def uq():
        return {
            "ref": {k: float(rng.choice([0.05, 0.3, 1.0])) for k in pl["ref"]},
            "tgt": {k: float(rng.choice([0.0, 0.2, 0.5, 0.95])) for k in pl["tgt"]},
            "src": {k: f"{rng.choice(ops)}{rng.choice([0, 0.5, 1])}" for k in pl["src"]},
        }

    stats = {"U": uq(), "Q": uq()}
    if anc and rng.random() < 0.7:
        for name in ("fd", "df", "Danc", "Dplus", "DD"):
            if rng.random() < 0.7:
                stats[name] = True
    win = int(rng.integers(500, 6000))
    step = int(rng.integers(100, win + 1))
    start = end = None
    if rng.random() < 0.4:  # a chunk the way ChunkGenerator cuts it: window-aligned bounds
        start = int(pos[n_sites // 4] // step * step + 1)
        end = start + int(rng.integers(1, 6)) * step + win - step - 1
    return dict(pos=pos, gts=gts, pl=pl, stats=stats, win=win, step=step, start=start, end=end, anc=anc, with_out=with_out)


def many_sources_scenario(seed):
    """``fuzz_scenario``'s shape with SEVEN to TEN source populations (the reference loops over any number of
    them, stat_utils.py:114-119, 141-152): one or two target populations, optional outgroup, ploidy 1-3, missing
    calls, both polarity modes, thresholds that let some windows through all the sources' conditions."""
    rng = np.random.default_rng(seed)
    n_sites = int(rng.integers(600, 1800))
    pos = np.cumsum(rng.integers(1, 60, n_sites)).astype(np.int32)
    n_src = int(rng.integers(7, 11))
    n_tgt = int(rng.integers(1, 3))
    with_out = bool(seed % 2)
    anc = True if with_out else bool(rng.random() < 0.5)
    p = rng.random(n_sites) ** 3
    intro = rng.random(n_sites) < 0.12  # sites every source carries (and the reference population lacks)

    def pop(n_ind, ploidy, role):
        pp = p.copy()
        if role == "ref":
            pp[intro] = 0.0
        elif role == "src":
            pp[intro] = 1.0
        g = rng.binomial(ploidy, np.broadcast_to(pp[:, None], (n_sites, n_ind))).astype(np.int64)
        g[rng.random(g.shape) < float(rng.choice([0.0, 0.01]))] = -ploidy
        return g

    pl = {"ref": {"R0": 2}, "tgt": {}, "src": {}}
    gts = {"ref": {"R0": pop(int(rng.integers(5, 30)), 2, "ref")}, "tgt": {}, "src": {}, "outgroup": {}}
    for i in range(n_tgt):
        ploidy = int(rng.integers(1, 4))
        pl["tgt"][f"T{i}"] = ploidy
        gts["tgt"][f"T{i}"] = pop(int(rng.integers(3, 25)), ploidy, "tgt")
    for i in range(n_src):
        ploidy = int(rng.integers(1, 4))
        pl["src"][f"S{i}"] = ploidy
        gts["src"][f"S{i}"] = pop(int(rng.integers(1, 4)), ploidy, "src")
    if with_out:
        pl["outgroup"] = {"O": 2}
        gts["outgroup"]["O"] = pop(int(rng.integers(1, 6)), 2, "out")

    def uq():
        return {
            "ref": {"R0": float(rng.choice([0.1, 0.3]))},
            "tgt": {k: float(rng.choice([0.0, 0.2, 0.9])) for k in pl["tgt"]},
            "src": {k: str(rng.choice(["=1", ">=0.5", ">0", "<=1", ">=0.75"])) for k in pl["src"]},
        }

    stats = {"U": uq(), "Q": uq()}
    if anc:
        for name in ("fd", "Dplus", "DD"):
            if rng.random() < 0.8:
                stats[name] = True
    win = int(rng.integers(1500, 6000))
    step = int(rng.integers(500, win + 1))
    return dict(pos=pos, gts=gts, pl=pl, stats=stats, win=win, step=step, start=None, end=None, anc=anc, with_out=with_out)


def siteset_scenario(kind, seed):
    """Populations whose site sets differ and / or repeat a position (what the reference's
    WindowGenerator resolves per window with intersect1d + isin, window_generator.py:193-231).

    kind: "ragged"      every population lacks a random ~12 % of the base sites (no repeats)
          "dup_u"       all populations share positions, some positions occur twice (a VCF with split
                        multiallelic records); U only
          "dup_rare"    the same with few repeats and few candidates (the reference then often runs
                        through and reports candidate positions shifted by the repeats)
          "dup_uq"      the same with U and Q configured
          "dup_uneven"  a position repeated in the target population only
          "unsorted"    all populations share one position array that is NOT ascending (blocks of an
                        unsorted VCF swapped): the reference's matrices keep the file order while its
                        `pos` is intersect1d's sorted array, so candidate positions come out permuted
          "unsorted_mixed"  every population has its OWN file order (and lacks a few sites): row k of one
                        population's window selection meets row k of another's and the k-th smallest position
    Returns dict(pos={group: {pop: int32}}, gts={group: {pop: int64}}, pl, stats, win, step, anc)."""
    rng = np.random.default_rng(seed)
    n_sites = int(rng.integers(300, 700))
    base = np.cumsum(rng.integers(1, 40, n_sites)).astype(np.int32) + 50
    p = rng.random(n_sites) ** 2
    intro = rng.random(n_sites) < (0.02 if kind == "dup_rare" else 0.15)
    dup = np.zeros(n_sites, bool)
    if kind.startswith("dup"):
        dup = rng.random(n_sites) < (0.012 if kind == "dup_rare" else 0.06)
    rows = np.repeat(np.arange(n_sites), 1 + dup.astype(np.int64))  # a duplicated site contributes two rows

    def pop(n_ind, ploidy, role):
        pp = p.copy()
        if role == "ref":
            pp[intro] = 0.0
        elif role == "src":
            pp[intro] = 1.0
        g = rng.binomial(ploidy, np.broadcast_to(pp[rows][:, None], (len(rows), n_ind))).astype(np.int64)
        g[rng.random(g.shape) < 0.02] = -ploidy
        return g

    pl = {"ref": {"R": 2}, "tgt": {"T0": 2, "T1": 1}, "src": {"S0": 2, "S1": 2}}
    sizes = {"R": 9, "T0": 8, "T1": 6, "S0": 1, "S1": 2}
    pos, gts = {}, {}
    for grp, role in (("ref", "ref"), ("tgt", "tgt"), ("src", "src")):
        pos[grp], gts[grp] = {}, {}
        for name, ploidy in pl[grp].items():
            g = pop(sizes[name], ploidy, role)
            keep = np.ones(len(rows), bool)
            if kind in ("ragged", "unsorted_mixed"):
                drop_site = rng.random(n_sites) < (0.12 if kind == "ragged" else 0.05)
                keep = ~drop_site[rows]
            elif kind == "dup_uneven" and name != "T0":
                keep = np.concatenate([[True], rows[1:] != rows[:-1]])  # only T0 keeps the second rows
            pos[grp][name] = base[rows][keep]
            gts[grp][name] = g[keep]
            if kind == "unsorted":  # the same file order in every population: a few blocks of records swapped
                order = np.arange(n_sites)
                cuts = np.sort(np.random.default_rng(seed + 1000).choice(np.arange(20, n_sites - 20), 6, replace=False))
                parts = np.split(order, cuts)
                order = np.concatenate([parts[i] for i in (0, 2, 1, 3, 5, 4, 6)])
                pos[grp][name], gts[grp][name] = pos[grp][name][order], gts[grp][name][order]
            if kind == "unsorted_mixed":  # a different swap of blocks of records in every population
                n_rows = len(pos[grp][name])
                own = np.random.default_rng(seed * 31 + len(pos) * 7 + len(pos[grp])).permutation(7)
                cuts = np.sort(rng.choice(np.arange(20, n_rows - 20), 6, replace=False))
                order = np.concatenate([np.split(np.arange(n_rows), cuts)[i] for i in own])
                pos[grp][name], gts[grp][name] = pos[grp][name][order], gts[grp][name][order]
    uq = lambda tgt: {"ref": {"R": 0.3}, "tgt": dict(tgt), "src": {"S0": "=1", "S1": ">=0.5"}}  # noqa: E731
    stats = {"U": uq({"T0": 0.2, "T1": 0.0})}
    if kind not in ("dup_u", "dup_rare"):
        stats["Q"] = uq({"T0": 0.9, "T1": 0.5})
    return dict(pos=pos, gts=gts, pl=pl, stats=stats, win=1500, step=500, anc=bool(seed % 2))
