"""GPU tests at BASELINE.json's full sizes.  The oracle cannot hold 10^7 x 2002 int64 genotypes,
so parity at size is shown by (a) size-independent properties -- conservation of the byte sums
against an independent torch reduction, prefix-sum identities between the per-site flags and the
per-window records, shard invariance (a site sub-range generated and scored on its own gives the
same windows), additivity of non-overlapping windows -- and (b) the oracle run on windows sampled
from the full-size block (their genotypes are copied back from HBM and un-tiled)."""

import numpy as np
import pytest

from conftest import same_f64

pytestmark = pytest.mark.gpu

SEED = 20260630


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from sai_amd.engine import Engine

    return Engine.get(0)


def untile(pop, lo, hi):
    """int64 [hi-lo][n_ind] genotypes of sites [lo, hi) of a TiledPop (copied from HBM)."""
    t0, t1 = lo // 64, (hi + 63) // 64
    raw = pop.tiles[t0 * pop.n_ind * 64 : t1 * pop.n_ind * 64].cpu().numpy()
    blk = raw.reshape(t1 - t0, pop.n_ind, 64).transpose(0, 2, 1).reshape(-1, pop.n_ind)
    return blk[lo - t0 * 64 : hi - t0 * 64].astype(np.int64)


def run_config(eng, config, n_sites, sizes, win, step, specs, chrom=1, mpm=0, site0=0):
    from sai_amd import _ffi
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    block = synth_block(eng, SEED + config, chrom, n_sites, sizes[0], sizes[1], sizes[2:], missing_per_million=mpm,
                        site0=site0)  # fmt: skip
    windows = default_windows(int(block.pos[0]), int(block.pos[-1]), win, step)
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
    scorer = ResidentScorer(eng, block, windows, sets, cap_u=1 << 22, cap_q=1 << 22)
    scorer.step()
    return block, windows, scorer, scorer.results()


def check_sampled_windows_against_oracle(block, windows, scorer, res, specs, sample):
    from oracle import sai_oracle as O

    lo = scorer.lo.cpu().numpy()
    hi = scorer.hi.cpu().numpy()
    pos = block.pos.cpu().numpy()
    for wi in sample:
        a, b = int(lo[wi]), int(hi[wi])
        assert (a, b) == (np.searchsorted(pos, windows[wi][0]), np.searchsorted(pos, windows[wi][1], side="right"))
        mats = [untile(p, a, b) for p in block.pops]
        for si, s in enumerate(specs):
            kw = dict(ref_gts=mats[0], tgt_gts=mats[1], src_gts_list=mats[2:], ref_ploidy=2, tgt_ploidy=2,
                      src_ploidy_list=[2] * (len(mats) - 2), pos=pos[a:b], w=s["w"], y_list=s["y_list"],
                      anc_allele_available=s["anc"])  # fmt: skip
            eu = O.u_stat(x=s["x"], **kw)
            eq = O.q_stat(quantile=s["quantile"], **kw)
            rec = res.records[si, wi]
            assert rec["n_sites"] == b - a and rec["u_count"] == eu["value"]
            assert res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()
            assert same_f64(rec["q"], eq["value"]), (wi, si, rec["q"], eq["value"])
            assert res.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()


def check_identities(eng, block, scorer, res):
    """Conservation + prefix-sum identities, everything recomputed with plain torch ops."""
    import torch

    counts = eng.site_counts(block.pops).to(torch.int64)  # the scorer itself runs the fused pass
    for p, pop in enumerate(block.pops):
        total, called = 0, 0
        step = 1 << 30  # bytes per slice, keeps the torch temporaries small
        for o in range(0, pop.tiles.numel(), step):
            t = pop.tiles[o : o + step]
            total += int(t.clamp(min=0).to(torch.int32).sum(dtype=torch.int64))
            called += int((t >= 0).sum())
        pad = pop.tiles.numel() - pop.n_sites * pop.n_ind  # zero bytes of the last tile count as called
        assert int(counts[p, :, 0].sum()) == total
        assert int(counts[p, :, 1].sum()) == called - pad
        assert int(counts[p, :, 1].max()) <= pop.n_ind
    lo, hi = scorer.lo.to(torch.int64), scorer.hi.to(torch.int64)
    flag_bytes = scorer.flag_bytes()  # one byte per set and site from the flag planes
    for si in range(scorer.n_sets):
        fl = flag_bytes[si]
        zero = torch.zeros(1, dtype=torch.int64, device=fl.device)
        cu = torch.cat([zero, torch.cumsum(((fl >> 1) & 1).to(torch.int64), 0)])
        cc = torch.cat([zero, torch.cumsum((fl & 1).to(torch.int64), 0)])
        assert (cu[hi] - cu[lo]).cpu().numpy().tolist() == res.records[si]["u_count"].tolist()
        assert (cc[hi] - cc[lo]).cpu().numpy().tolist() == res.records[si]["n_cond"].tolist()
        assert ((fl & 2) <= ((fl & 1) << 1)).all()  # a U candidate always satisfies the condition
        q = res.records[si]["q"]
        assert np.array_equal(np.isnan(q), res.records[si]["n_cond"] == 0)
        assert np.all((q[~np.isnan(q)] >= 0) & (q[~np.isnan(q)] <= 1))
        assert np.all(res.records[si]["n_cdd_q"][res.records[si]["n_cond"] > 0] >= 1)
    assert res.records[0]["n_sites"].tolist() == (hi - lo).cpu().numpy().tolist()


C3_SPECS = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=True),
            dict(w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=False)]  # fmt: skip


def test_c3_full_size(eng):
    """configs[2]: 1e7 sites, 1000 ref / 1000 tgt / 2 src, 50kb/25kb, U + Q95 (both polarity modes)."""
    n = 10_000_000
    block, windows, scorer, res = run_config(eng, 3, n, [1000, 1000, 2], 50000, 25000, C3_SPECS)
    assert len(windows) in range(9900, 10100) and res.records.shape == (2, len(windows))
    check_identities(eng, block, scorer, res)
    rng = np.random.default_rng(1)
    hot = np.argsort(-res.records[0]["u_count"])[:6].tolist()
    sample = sorted(set(hot + rng.integers(0, len(windows), 10).tolist() + [0, len(windows) - 1]))
    check_sampled_windows_against_oracle(block, windows, scorer, res, C3_SPECS, sample)
    assert res.records[0]["u_count"].sum() > 1000  # the workload really exercises the statistic

    # shard invariance: sites [a, b) generated on their own (as another GPU would) give identical windows
    a = 4_000_000 + 17
    b = a + 1_500_000
    sub, sw, ssc, sres = run_config(eng, 3, b - a, [1000, 1000, 2], 50000, 25000, C3_SPECS, site0=a)
    pos = block.pos.cpu().numpy()
    assert np.array_equal(sub.pos.cpu().numpy(), pos[a:b])
    index = {w: i for i, w in enumerate(windows)}
    inner = [(i, index[w]) for i, w in enumerate(sw) if w in index and w[0] >= pos[a] and w[1] <= pos[b - 1]]
    assert len(inner) > 1400
    for si in range(2):
        for i, j in inner:
            ra, rb = sres.records[si, i], res.records[si, j]
            assert ra.tobytes() == rb.tobytes()
            assert sres.u_list(si, i).tolist() == res.u_list(si, j).tolist()
            assert sres.q_list(si, i).tolist() == res.q_list(si, j).tolist()


def test_c2_full_size(eng):
    """configs[1]: 1e6 sites, 200 ref / 200 tgt / 2 src, 50kb/10kb windows, U; with 0.1 % missing calls."""
    specs = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=True)]
    block, windows, scorer, res = run_config(eng, 2, 1_000_000, [200, 200, 2], 50000, 10000, specs, mpm=1000)
    assert len(windows) in range(2450, 2550)
    check_identities(eng, block, scorer, res)
    check_sampled_windows_against_oracle(block, windows, scorer, res, specs, list(range(0, len(windows), 40)))
    # additivity: 50 kb windows every 10 kb -> window k and k+5 are disjoint and tile the chromosome;
    # their U counts must add up to the count over the union (checked through the flags)
    import torch

    fl = scorer.flag_bytes()[0]
    lo, hi = scorer.lo.cpu().numpy(), scorer.hi.cpu().numpy()
    for k in (3, 500, 2000):
        chain = list(range(k, min(k + 50, len(windows)), 5))
        assert all(hi[chain[i]] == lo[chain[i + 1]] for i in range(len(chain) - 1))
        union = int(((fl[lo[chain[0]] : hi[chain[-1]]] >> 1) & 1).sum())
        assert union == int(res.records[0]["u_count"][chain].sum())


def test_c4_whole_genome_shards(eng):
    """configs[3] on one GPU: chromosomes of 5e6 sites scored as contiguous window-range shards
    (the multi-GPU decomposition) equal the unsharded run; two chromosomes stand for the 22."""
    from sai_amd.distributed import my_chunk_indices

    specs = C3_SPECS[:1]
    for chrom in (1, 22):
        n = 5_000_000
        block, windows, scorer, res = run_config(eng, 4, n, [1000, 1000, 2], 50000, 25000, specs, chrom=chrom)
        pos = block.pos.cpu().numpy()
        world = 8
        covered = 0
        for rank in (0, 3, 7):
            idx = my_chunk_indices(len(windows), rank, world)
            a = int(np.searchsorted(pos, windows[idx[0]][0]))
            b = int(np.searchsorted(pos, windows[idx[-1]][1], side="right"))
            sub, sw, ssc, sres = run_config(eng, 4, b - a, [1000, 1000, 2], 50000, 25000, specs, chrom=chrom, site0=a)
            # the shard's own grid (chunk re-derivation, window_generator.py:133-142) contains its windows
            pick = {w: i for i, w in enumerate(sw)}
            for j in idx:
                i = pick[windows[j]]
                assert sres.records[0, i].tobytes() == res.records[0, j].tobytes()
                assert sres.u_list(0, i).tolist() == res.u_list(0, j).tolist()
                covered += 1
        assert covered >= 3 * (len(windows) // world)


def test_c5_two_sources_sweep(eng):
    """configs[4]: two source populations of one diploid each, 18 parameter sets (3x3 y grid x
    {"=", ">="}) answered from ONE site_counts pass; full size through identities, and every set
    against the oracle on sampled windows."""
    import torch

    from sai_amd import _ffi
    from sai_amd.resident import default_windows, synth_block

    n = 10_000_000
    block = synth_block(eng, SEED + 5, 1, n, 1000, 1000, [1, 1])
    windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 50000, 25000)
    specs = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[(op, y1), (op, y2)], anc=True)
             for op in ("=", ">=") for y1 in (0.0, 0.5, 1.0) for y2 in (0.0, 0.5, 1.0)]  # fmt: skip
    assert len(specs) == 18
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
    counts = eng.site_counts(block.pops)  # the only pass over the 20 GB of genotypes
    tgt_freq, flags, _ = eng.site_flags(counts, block.ploidies, sets)
    lo, hi = eng.window_bounds(block.pos, [w[0] for w in windows], [w[1] for w in windows])
    res = eng.window_stats(tgt_freq, flags, sets, lo, hi, pos=block.pos, cap_hint=1 << 22)
    assert res.records.shape == (18, len(windows))
    lo64, hi64 = lo.to(torch.int64), hi.to(torch.int64)
    zero = torch.zeros(1, dtype=torch.int64, device=flags.device)
    flag_bytes = eng.flag_bytes(flags, block.n_sites, sets, tgt_freq)
    for si in range(18):
        cu = torch.cat([zero, torch.cumsum(((flag_bytes[si] >> 1) & 1).to(torch.int64), 0)])
        assert (cu[hi64] - cu[lo64]).cpu().numpy().tolist() == res.records[si]["u_count"].tolist()
    # ">= 0" on both sources accepts every valid site; "= 1, = 1" is the strictest
    assert res.records[9]["n_cond"].sum() >= res.records[8]["n_cond"].sum()

    class S:  # what check_sampled_windows_against_oracle needs from a scorer
        pass

    s = S()
    s.lo, s.hi = lo, hi
    hot = np.argsort(-res.records[8]["u_count"])[:3].tolist()
    check_sampled_windows_against_oracle(block, windows, s, res, specs, sorted(set(hot + [7, 5000])))


def test_c3_packed2_equals_int8(eng):
    """configs[2] once more on the optional 2-bit layout: every record and candidate list equal to
    the int8 run."""
    from sai_amd import _ffi
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    block = synth_block(eng, SEED + 3, 1, 10_000_000, 1000, 1000, [2], missing_per_million=1000)
    windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 50000, 25000)
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in C3_SPECS]
    a = ResidentScorer(eng, block, windows, sets, cap_u=1 << 22, cap_q=1 << 22)
    b = ResidentScorer(eng, block, windows, sets, cap_u=1 << 22, cap_q=1 << 22, layout="packed2")
    a.step()
    b.step()
    ra, rb = a.results(), b.results()
    assert ra.records.tobytes() == rb.records.tobytes()
    assert np.array_equal(ra.offsets, rb.offsets) and np.array_equal(ra.cdd_u, rb.cdd_u) and np.array_equal(ra.cdd_q, rb.cdd_q)
    assert sum(p.data.numel() for p in b.packed) * 3.5 < sum(p.tiles.numel() for p in block.pops)


# ------------------------------------------------------------------------------------------
# exactly what bench.py times ("GPU tests on exactly what the bench runs")
# ------------------------------------------------------------------------------------------


def _bench_scorer(eng, wl, steps=3):
    """The scorer bench.py builds for a workload on one GPU: build_synth_shard(world=1) -> one block
    of the job's chromosome pieces -> ResidentScorer(overlap=True), `steps` pipelined steps."""
    from sai_amd.resident import ResidentScorer
    from sai_amd.sharding import build_synth_shard

    block, lay, win_counts = build_synth_shard(eng, wl, 0, 1)
    scorer = ResidentScorer(eng, block, [(s, e) for _, s, e in lay.windows], wl.params(), cap_u=1 << 22, cap_q=1 << 22,
                            overlap=True, window_segment=lay.window_segment)  # fmt: skip
    for _ in range(steps):
        scorer.step()
    return block, lay, scorer, scorer.results()


def check_piece_windows_against_oracle(block, lay, scorer, res, specs, sample):
    """check_sampled_windows_against_oracle for a block of several pieces: the window's site range is
    searched inside its own piece (positions ascend per piece only)."""
    from oracle import sai_oracle as O

    lo, hi = scorer.lo.cpu().numpy(), scorer.hi.cpu().numpy()
    pos = block.pos.cpu().numpy()
    for wi in sample:
        _, start, end = lay.windows[wi]
        s0, s1 = lay.segments[int(lay.window_segment[wi])]
        a, b = int(lo[wi]), int(hi[wi])
        assert (a, b) == (s0 + np.searchsorted(pos[s0:s1], start), s0 + np.searchsorted(pos[s0:s1], end, side="right")), wi
        mats = [untile(p, a, b) for p in block.pops]
        for si, s in enumerate(specs):
            kw = dict(ref_gts=mats[0], tgt_gts=mats[1], src_gts_list=mats[2:], ref_ploidy=2, tgt_ploidy=2,
                      src_ploidy_list=[2] * (len(mats) - 2), pos=pos[a:b], w=s["w"], y_list=s["y_list"],
                      anc_allele_available=s["anc"])  # fmt: skip
            eu, eq = O.u_stat(x=s["x"], **kw), O.q_stat(quantile=s["quantile"], **kw)
            rec = res.records[si, wi]
            assert rec["n_sites"] == b - a and rec["u_count"] == eu["value"], (wi, si)
            assert res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()
            assert same_f64(rec["q"], eq["value"]), (wi, si, rec["q"], eq["value"])
            assert res.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()


def test_c5_fused_sweep_pipelined_full_size(eng):
    """configs[4] as bench.py --workload c5 runs it: 1e7 sites, the 18 sets riding in ONE fused site
    pass, steps pipelined on two streams -- identities over the whole block and the oracle on sampled
    windows, among them heavy ones (sets 0 and 9 select ~620 sites per window: the workgroup select)."""
    import bench

    wl = bench.make_workload("c5")
    block, lay, scorer, res = _bench_scorer(eng, wl)
    assert scorer.fused and scorer.overlap and len(scorer.chunks) == 1 and res.records.shape == (18, len(lay.windows))
    assert len(lay.windows) in range(9900, 10100)
    check_identities(eng, block, scorer, res)
    heavy = [si for si in range(18) if int(np.median(res.records[si]["n_cond"])) > 256]
    assert 0 in heavy and 9 in heavy  # ("=0", "=0") and (">=0", ">=0")
    rng = np.random.default_rng(5)
    sample = sorted({int(np.argmax(res.records[0]["n_cond"])), int(np.argmax(res.records[9]["n_cond"])),
                     int(np.argmax(res.records[8]["u_count"])), 0, len(lay.windows) - 1, *rng.integers(0, len(lay.windows), 3).tolist()})  # fmt: skip
    assert len(sample) >= 6
    check_piece_windows_against_oracle(block, lay, scorer, res, wl.specs, sample)
    # the pipelined steps repeat themselves: the un-pipelined scorer on the same block gives the same bytes
    from sai_amd.resident import ResidentScorer

    plain = ResidentScorer(eng, block, [(s, e) for _, s, e in lay.windows], wl.params(), cap_u=1 << 22, cap_q=1 << 22,
                           window_segment=lay.window_segment)  # fmt: skip
    plain.step()
    want = plain.results()
    assert want.records.tobytes() == res.records.tobytes()
    assert np.array_equal(want.cdd_u, res.cdd_u) and np.array_equal(want.cdd_q, res.cdd_q)


def test_c4_multi_piece_block_full_size(eng):
    """configs[3]'s block as bench.py --workload c4 lays it out on one GPU, with three full-size
    chromosomes (3 x 5e6 sites, 30 GB): pieces back to back at tile boundaries, one site pass and one
    windows stage for all of them.  Windows next to every piece boundary against the oracle, and the
    identities over the whole block."""
    import bench

    wl = bench.make_workload("c4", chroms=3)
    block, lay, scorer, res = _bench_scorer(eng, wl)
    assert len(lay.pieces) == 3 and block.segments is not None and res.records.shape == (1, len(lay.windows))
    assert len(lay.windows) in range(14900, 15100) and block.n_real_sites == 3 * 5_000_000
    check_identities(eng, block, scorer, res)
    seg = np.asarray(lay.window_segment)
    edges = np.flatnonzero(np.diff(seg)) + 1  # first window of pieces 1 and 2
    assert len(edges) == 2
    sample = sorted({0, len(lay.windows) - 1, *[int(e) + d for e in edges for d in (-2, -1, 0, 1)],
                     *np.argsort(-res.records[0]["u_count"])[:3].tolist()})  # fmt: skip
    check_piece_windows_against_oracle(block, lay, scorer, res, wl.specs, sample)
    # the last window of a piece ends inside its piece; the first of the next starts at that piece's first site
    lo, hi = scorer.lo.cpu().numpy(), scorer.hi.cpu().numpy()
    for e in edges:
        assert hi[e - 1] <= lay.segments[seg[e - 1]][1] and lo[e] == lay.segments[seg[e]][0]
    assert res.records[0]["u_count"].sum() > 1000
