"""GPU tests of the sharded resident job: a rank's share of a whole-genome
window list -- several chromosome pieces laid back to back in one block, searched per segment --
gives, merged over the ranks, exactly the bytes of the one-GPU run of the same job; and bench.py's
N > 1 branch runs end to end with two ranks on the box's one GPU (gloo moves the rows; RCCL refuses
two ranks on one device) and with one rank on real RCCL."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, same_f64

SEED = 20260630

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from sai_amd.engine import Engine

    return Engine.get(0)


def small_job(name, sites=60_000, chroms=5):
    import bench

    wl = bench.make_workload(name, sites=sites, chroms=chroms if name in ("c4", "c2x22") else 0)
    wl.n_ref, wl.n_tgt = 130, 70
    wl.missing_per_million = 2000
    for s in wl.specs:
        s.update(w=0.05, x=0.3)
    return wl


def run_rank(eng, wl, rank, world, overlap=False, through_row=True):
    """One rank's pass; returns (WindowResults or None, layout).  ``through_row`` sends the results
    through pack_row / RowLayout.unpack, the bytes a gather moves."""
    import torch

    from sai_amd.resident import ResidentScorer
    from sai_amd.sharding import build_synth_shard

    block, lay, counts = build_synth_shard(eng, wl, rank, world)
    if block is None:
        return None, lay, counts
    scorer = ResidentScorer(eng, block, [(s, e) for _, s, e in lay.windows], wl.params(), cap_u=1 << 18, cap_q=1 << 18,
                            overlap=overlap, window_segment=lay.window_segment)  # fmt: skip
    scorer.step()
    scorer.step()  # the pipelined form needs a second step + flush to have run a stage
    res = scorer.results()
    if through_row:
        layout = scorer.row_layout()
        row = torch.empty((layout.nbytes,), dtype=torch.uint8, device=eng.device)
        with scorer.window_stream():
            scorer.pack_row(row, layout)
        torch.cuda.synchronize()
        from sai_amd.resident import RowLayout

        back = RowLayout.from_header(layout.header()).unpack(row.cpu().numpy())
        assert back.records.tobytes() == res.records.tobytes()
        assert back.cdd_u.tobytes() == res.cdd_u.tobytes() and back.cdd_q.tobytes() == res.cdd_q.tobytes()
        assert np.array_equal(back.offsets, res.offsets)
        res = back
    return res, lay, counts


@pytest.mark.parametrize("name,world", [("c4", 3), ("c4", 8), ("c5", 4), ("c2", 2)])
def test_merged_shards_equal_the_one_gpu_job(eng, name, world):
    from sai_amd.sharding import merge_rank_results, plan_shards

    wl = small_job(name)
    one, lay1, counts = run_rank(eng, wl, 0, 1)
    n_sets = len(wl.specs)
    assert one.records.shape == (n_sets, sum(counts)) and one.records["u_count"].sum() > 0 and one.cdd_q.size > 0
    per_rank, n_pieces = [], 0
    for r in range(world):
        res, lay, c = run_rank(eng, wl, r, world, overlap=(r % 2 == 1))
        assert c == counts
        per_rank.append(res)
        n_pieces += len(lay.pieces)
        # a rank holds only its own sites + halo
        assert sum(lay.n_sites) < len(wl.chroms) * wl.n_sites / world + (len(lay.pieces) + 1) * 4000
    merged = merge_rank_results(per_rank, plan_shards(counts, world), n_sets)
    assert merged.records.tobytes() == one.records.tobytes()
    assert merged.cdd_u.tobytes() == one.cdd_u.tobytes() and merged.cdd_q.tobytes() == one.cdd_q.tobytes()
    assert np.array_equal(merged.offsets, one.offsets)
    if name == "c4":
        assert n_pieces >= world + len(wl.chroms) - 1  # ranks really span chromosome boundaries


@pytest.mark.parametrize("name,sites,chroms", [("c4", 50_000, 4), ("c2x22", 20_000, 22)])
def test_multi_piece_block_equals_per_chromosome_blocks(eng, name, sites, chroms):
    """The one-rank block of a 4-chromosome job (4 segments, one site pass, one windows stage)
    against four single-chromosome scorers on separately generated blocks; and the 22-piece block of
    `bench.py --workload c2x22` (C2's populations and window grid) against 22 single-chromosome runs."""
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    wl = small_job(name, sites=sites, chroms=chroms)
    assert len(wl.chroms) == chroms
    one, lay, counts = run_rank(eng, wl, 0, 1, through_row=False)
    assert len(lay.pieces) == chroms
    g = 0
    for chrom, n_w in zip(wl.chroms, counts):
        block = synth_block(eng, wl.seed, chrom, wl.n_sites, wl.n_ref, wl.n_tgt, wl.src_sizes,
                            missing_per_million=wl.missing_per_million)  # fmt: skip
        windows = default_windows(int(block.pos[0]), int(block.pos[-1]), wl.win_len, wl.win_step)
        assert len(windows) == n_w
        sc = ResidentScorer(eng, block, windows, wl.params(), cap_u=1 << 18, cap_q=1 << 18)
        sc.step()
        res = sc.results()
        assert one.records[:, g : g + n_w].tobytes() == res.records.tobytes()
        for w in range(n_w):
            assert one.u_list(0, g + w).tolist() == res.u_list(0, w).tolist()
            assert one.q_list(0, g + w).tolist() == res.q_list(0, w).tolist()
        g += n_w


def test_more_than_sixteen_sets_in_one_scorer(eng):
    """C5's 18 parameter sets: one scorer (two set chunks) against engine-level calls."""
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    wl = small_job("c5", sites=80_000)
    block = synth_block(eng, wl.seed, 1, wl.n_sites, wl.n_ref, wl.n_tgt, wl.src_sizes, missing_per_million=2000)
    windows = default_windows(int(block.pos[0]), int(block.pos[-1]), wl.win_len, wl.win_step)
    sets = wl.params()
    assert len(sets) == 18
    sc = ResidentScorer(eng, block, windows, sets, cap_u=16, cap_q=16)  # far too small: results(grow=True) enlarges
    sc.step()
    with pytest.raises(RuntimeError, match="too small"):
        sc.results()
    res = sc.results(grow=True)
    counts = eng.site_counts(block.pops)
    tgt_freq, flags, _ = eng.site_flags(counts, block.ploidies, sets)
    lo, hi = eng.window_bounds(block.pos, [w[0] for w in windows], [w[1] for w in windows])
    ref = eng.window_stats(tgt_freq, flags, sets, lo, hi, pos=block.pos)
    assert res.records.tobytes() == ref.records.tobytes()
    assert res.cdd_u.tobytes() == ref.cdd_u.tobytes() and res.cdd_q.tobytes() == ref.cdd_q.tobytes()
    assert np.array_equal(res.offsets, ref.offsets)
    assert len({int(r["n_cond"].sum()) for r in res.records}) > 3  # the sets really differ


def test_more_sets_than_one_call_carries(eng):
    """45 parameter sets: more than SAI_MAX_SETS = SAI_FUSED_SETS = 20, so the scorer reduces the
    genotypes once (site_counts), evaluates the sets in three site_flags launches that each write their
    own columns of the shared plane rows (row stride 135 words), and runs three windows stages on column
    slices of those rows -- every record and list against the per-window oracle on sampled windows, and
    against one-set-at-a-time scorers (the fused single-launch path)."""
    from oracle import sai_oracle as O
    from sai_amd import _ffi
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    rng = np.random.default_rng(45)
    block = synth_block(eng, SEED + 9, 1, 60_000, 40, 30, [2], missing_per_million=3000)
    windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 20_000, 10_000)
    specs = [dict(w=float(rng.choice([0.05, 0.2, 0.5, 1.0])), x=float(rng.choice([0.0, 0.3, 0.6])),
                  quantile=float(rng.choice([0.5, 0.9, 0.95, 1.0])), y_list=[(str(rng.choice(["=", ">=", "<="])), float(rng.choice([0.0, 0.5, 1.0])))],
                  anc=bool(s % 3)) for s in range(45)]  # fmt: skip
    sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]
    sc = ResidentScorer(eng, block, windows, sets, cap_u=1 << 20, cap_q=1 << 20, overlap=True)
    assert not sc.fused and len(sc.chunks) == 3 and tuple(sc.flags.shape) == ((60_000 + 63) // 64, 135)
    for _ in range(3):
        sc.step()
    res = sc.results()
    assert res.records.shape == (45, len(windows))
    for si in (0, 19, 20, 39, 40, 44):  # both sides of every chunk boundary
        one = ResidentScorer(eng, block, windows, [sets[si]], cap_u=1 << 20, cap_q=1 << 20)
        assert one.fused
        one.step()
        want = one.results()
        assert want.records[0].tobytes() == res.records[si].tobytes(), si
        for wi in range(0, len(windows), 7):
            assert want.u_list(0, wi).tolist() == res.u_list(si, wi).tolist()
            assert want.q_list(0, wi).tolist() == res.q_list(si, wi).tolist()
    from test_hip_fullsize import untile

    lo, hi = sc.lo.cpu().numpy(), sc.hi.cpu().numpy()
    pos = block.pos.cpu().numpy()
    n_checked = 0
    for wi in (0, len(windows) // 2, len(windows) - 1):
        a, b = int(lo[wi]), int(hi[wi])
        mats = [untile(p, a, b) for p in block.pops]
        for si in range(0, 45, 4):
            s = specs[si]
            kw = dict(ref_gts=mats[0], tgt_gts=mats[1], src_gts_list=mats[2:], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2],
                      pos=pos[a:b], w=s["w"], y_list=s["y_list"], anc_allele_available=s["anc"])  # fmt: skip
            eu, eq = O.u_stat(x=s["x"], **kw), O.q_stat(quantile=s["quantile"], **kw)
            rec = res.records[si, wi]
            assert rec["u_count"] == eu["value"] and res.u_list(si, wi).tolist() == eu["cdd_pos"].tolist()
            assert same_f64(rec["q"], eq["value"]) and res.q_list(si, wi).tolist() == np.asarray(eq["cdd_pos"]).astype(np.int64).tolist()
            n_checked += int(eu["value"] > 0)
    assert n_checked > 5


@pytest.mark.parametrize("trial", range(int(os.environ.get("SAI_SCORER_FUZZ", "8"))))
def test_scorer_forms_agree_on_random_blocks(eng, trial):
    """Random blocks (1 .. 6 000 sites, some in several pieces), populations, 1 .. 25 parameter sets, window
    grids with empty and very wide windows: the resident scorer in all its forms -- fused pass, counts
    handed in, more sets than one launch carries, packed2, pipelined over three steps, rebound to the same
    block -- returns the bytes of the engine-level calls (site_counts + site_flags + window_stats, which
    tests/test_hip_kernels.py pins to the oracle).  SAI_SCORER_FUZZ=300 was run once on the GPU box."""
    import torch

    from sai_amd import _ffi
    from sai_amd.engine import TiledPop
    from sai_amd.resident import ResidentBlock, ResidentScorer

    rng = np.random.default_rng(9000 + trial)
    n_src = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 90)), int(rng.integers(1, 90))] + [int(rng.integers(1, 4)) for _ in range(n_src)]
    n_pieces = int(rng.choice([1, 1, 2, 3]))
    piece_sites = [int(rng.integers(1, 2500)) for _ in range(n_pieces)]
    tile0 = np.concatenate([[0], np.cumsum([(n + 63) // 64 for n in piece_sites])])
    n_total = int(tile0[-1]) * 64
    max_dosage = int(rng.choice([1, 2, 2, 3]))
    mats, pos = [], np.zeros(n_total, dtype=np.int32)
    for n_ind in sizes:
        g = np.zeros((n_total, n_ind), dtype=np.int8)
        for k, n in enumerate(piece_sites):
            blk = rng.integers(0, max_dosage + 1, size=(n, n_ind)).astype(np.int8)
            blk[rng.random(blk.shape) < 0.05] = -max_dosage
            g[tile0[k] * 64 : tile0[k] * 64 + n] = blk
        mats.append(g)
    segments = []
    for k, n in enumerate(piece_sites):
        pos[tile0[k] * 64 : tile0[k] * 64 + n] = np.cumsum(rng.integers(1, 60, n)) + 10
        segments.append((int(tile0[k]) * 64, int(tile0[k]) * 64 + n))
    pops = [eng.tile(m) for m in mats]
    block = ResidentBlock(pops, [int(rng.integers(1, 4)) if max_dosage > 2 else 2 for _ in sizes], torch.from_numpy(pos).to(eng.device),
                          segments=segments if n_pieces > 1 or rng.random() < 0.3 else None)  # fmt: skip
    if block.segments is None and n_pieces == 1:
        block = ResidentBlock([TiledPop(p.tiles, piece_sites[0], p.n_ind) for p in pops], block.ploidies, block.pos[: piece_sites[0]])
    windows, wseg = [], []
    for k, (a, b) in enumerate(segments):
        p_lo, p_hi = int(pos[a]), int(pos[b - 1])
        win, step = int(rng.integers(50, 4000)), int(rng.integers(25, 3000))
        for start in range(max(p_lo - win, 1), p_hi + step, step):
            windows.append((start, start + win - 1))
            wseg.append(k)
        windows.append((1, p_hi + 100))  # everything
        wseg.append(k)
        windows.append((p_hi + 1000, p_hi + 2000))  # nothing
        wseg.append(k)
    n_sets = int(rng.choice([1, 2, 5, 18, 21, 25]))
    ops = ["=", "<", ">", "<=", ">="]
    sets = [_ffi.make_params(float(rng.choice([0.05, 0.3, 1.0])), float(rng.choice([0.0, 0.2, 0.6])), float(rng.choice([0.0, 0.5, 0.95, 1.0])),
                             [(str(rng.choice(ops)), float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(n_src)], bool(rng.integers(2)))
            for _ in range(n_sets)]  # fmt: skip
    kw = dict(cap_u=1 << 20, cap_q=1 << 20, window_segment=wseg if block.segments is not None else None)
    # engine-level reference
    counts = eng.site_counts(block.pops)
    tgt_freq, planes, _ = eng.site_flags(counts, block.ploidies, sets)
    ref_sc = ResidentScorer(eng, block, windows, sets[:1], **kw)
    ref_sc.step()
    ref_sc.results()
    want = eng.window_stats(tgt_freq, planes, sets, ref_sc.lo, ref_sc.hi, pos=block.pos, cap_hint=1 << 18)

    def same(res, what):
        assert res.records.tobytes() == want.records.tobytes(), what
        assert np.array_equal(res.offsets, want.offsets), what
        assert np.array_equal(res.cdd_u, want.cdd_u) and np.array_equal(res.cdd_q, want.cdd_q), what

    plain = ResidentScorer(eng, block, windows, sets, **kw)
    assert plain.fused == (n_sets <= _ffi.SAI_FUSED_SETS)
    plain.step()
    same(plain.results(grow=True), "plain")
    piped = ResidentScorer(eng, block, windows, sets, overlap=True, **kw)
    for _ in range(3):
        piped.step()
    same(piped.results(grow=True), "pipelined")
    given = ResidentScorer(eng, block, windows, sets, counts_in=counts, **kw)
    given.step()
    same(given.results(grow=True), "counts handed in")
    plain.rebind(block, sets, counts_in=counts)
    plain.step()
    same(plain.results(grow=True), "rebound to counts")
    plain.rebind(block, sets)
    plain.step()
    same(plain.results(grow=True), "rebound back")
    if max_dosage <= 2 and n_sets <= _ffi.SAI_FUSED_SETS:
        packed = ResidentScorer(eng, block, windows, sets, layout="packed2", **kw)
        packed.step()
        same(packed.results(grow=True), "packed2")


def _bench(args, env_extra=None, nproc=1):
    env = dict(os.environ)
    env.update(env_extra or {})
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
               "127.0.0.1", "--master-port", "29577", "bench.py", "--gpus", str(nproc), *args]  # fmt: skip
    else:
        cmd = [sys.executable, "bench.py", *args]
    res = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


REDUCED = ["--workload", "c4", "--sites", "200000", "--chroms", "5", "--steps", "3", "--warmup", "1", "--cpu-sites", "0"]


@pytest.mark.parametrize("gather", ["step", "end"])
def test_bench_two_ranks_on_one_gpu_equal_one_rank(gather):
    """bench.py's N > 1 branch with two processes on the box's one GPU (real kernels; gloo moves
    the rows) against the one-process run of the same reduced job."""
    one = _bench(REDUCED)
    two = _bench([*REDUCED, "--gather", gather], {"SAI_BENCH_DEVICE": "0", "SAI_BENCH_BACKEND": "gloo"}, nproc=2)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["gather"] == gather
    for k in ("windows_total", "u_sum", "q_finite", "cdd_u_entries", "cdd_q_entries", "chromosomes", "parameter_sets"):
        assert two["config"][k] == one["config"][k], k
    assert one["config"]["u_sum"] > 0 and one["config"]["windows_total"] > 500
    assert two["config"]["windows_rank0"] in (one["config"]["windows_total"] // 2, one["config"]["windows_total"] // 2 + 1)
    assert two["config"]["pieces_rank0"] == 3 and one["config"]["pieces_rank0"] == 5
    assert "configs[3]" in two["config"]["workload"] and "REDUCED" in two["config"]["workload"]
    assert two["roofline"]["frac"] > 0 and two["value"] > 0
    assert "2 rank(s) match" in two["config"]["gather_check"] and one["config"]["gather_check"] is None
    ranks = two["config"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and sum(r["windows"] for r in ranks) == one["config"]["windows_total"]
    assert all(r["site_pass_avg_ms"] > 0 and r["ms_per_step_wall"] > 0 for r in ranks) and len(one["config"]["per_rank"]) == 1


def test_plain_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher on the command line (how the driver starts the
    N = 1 run): bench.py starts its ranks itself as a child job and relays rank 0's line; same totals
    as the one-process run, and the line names the one-GPU job it is to be divided by."""
    one = _bench(REDUCED)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAI_BENCH_DEVICE="0", SAI_BENCH_BACKEND="gloo")
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *REDUCED], cwd=str(ROOT), env=env, capture_output=True,
                         text=True, timeout=900)  # fmt: skip
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    for k in ("windows_total", "u_sum", "q_finite", "cdd_u_entries", "cdd_q_entries", "chromosomes", "parameter_sets"):
        assert two["config"][k] == one["config"][k], k
    assert "2 rank(s) match" in two["config"]["gather_check"]
    base = two["config"]["one_gpu_base"]
    assert base["workload_id"] == "c4" and "note" in base  # a reduced job has no stored base
    assert one["config"]["one_gpu_base"] is None


def test_bench_without_ancestral_alleles_is_a_different_job():
    """--anc false (SURVEY 8d's second row): every set matches the sources against 1 - y as well and
    inverts those sites; same windows, other counts, and the line says so."""
    args = ["--workload", "c3", "--sites", "400000", "--steps", "2", "--warmup", "1", "--cpu-sites", "0", "--score-path", "off"]
    yes, no = _bench(args), _bench([*args, "--anc", "false"])
    assert no["config"]["anc_allele_available"] is False and "anc_allele_available=False" in no["config"]["workload"]
    assert no["config"]["windows_total"] == yes["config"]["windows_total"]
    # y = 1: the mirror 1 - y = 0 also matches the many sites where the source carries no derived allele; they
    # are inverted (ref' = 1 - ref) and pass `ref' < w` only where the reference is nearly fixed, so the synthetic
    # job gains few windows, never loses one
    assert no["config"]["q_finite"] >= yes["config"]["q_finite"] and no["config"]["u_sum"] >= yes["config"]["u_sum"] > 0
    assert no["roofline"]["traffic"] is None  # a reduced job has no stored counter figure


def test_bench_one_rank_on_real_rccl():
    """The same branches -- process group with device_id, header all_gather, the per-pass gather on
    the window stream, the MAX reduction -- with one rank on real RCCL."""
    one = _bench(REDUCED)
    rccl = _bench([*REDUCED[:-2], "--cpu-sites", "20000", "--cpu-run-seconds", "0.3"], {"SAI_BENCH_FORCE_DIST": "1"})
    for k in ("windows_total", "u_sum", "q_finite", "cdd_u_entries", "cdd_q_entries"):
        assert rccl["config"][k] == one["config"][k], k
    # the line of a job with a process group is as complete as the plain one
    assert rccl["cpu_baseline"]["value"] > 0 and rccl["cpu_baseline"]["kind"] == "port"
    assert rccl["roofline"]["frac"] > 0 and "traffic" in rccl["roofline"] and "one_gpu_base" in rccl["config"]
    (r0,) = rccl["config"]["per_rank"]
    assert r0["rank"] == 0 and r0["windows"] == one["config"]["windows_total"] and r0["sites"] == one["config"]["sites_rank0"]
    assert 0 < r0["site_pass_avg_ms"] <= r0["ms_per_step_wall"] * 1.05 and abs(r0["site_pass_avg_ms"] - rccl["roofline"]["avg_launch_ms"]) < 1e-3
    assert rccl["config"]["gather"] == "step" and rccl["config"]["gather_row_bytes"][0] > 24 * one["config"]["windows_total"]
    assert "1 rank(s) match" in rccl["config"]["gather_check"]
    # the record proves by itself what the group was and where the rank ran: RCCL, the world size the
    # group saw, the device's PCI bus id / UUID, and what every per-pass gather took on the stream it was issued on
    coll = rccl["config"]["collective"]
    assert coll["backend"] == "nccl" and coll["world_size_seen_by_group"] == 1 and coll["distinct_devices"] is True
    assert coll["rccl_version"] and coll["rccl_version"][0].isdigit() and coll["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert r0["device_index"] == 0 and len(r0["pci_bus_id"].split(":")) == 3 and len(r0["uuid"]) == 32 and r0["hostname"]
    assert r0["gathers_timed"] == 3 and 0 < r0["gather_avg_ms_on_stream"] <= r0["gather_max_ms_on_stream"] < r0["ms_per_step_wall"] * 3
    assert one["config"]["collective"] is None and one["config"]["per_rank"][0]["pci_bus_id"] == r0["pci_bus_id"]


def test_bench_default_line_has_the_contract_fields():
    line = _bench(["--workload", "c2", "--steps", "5", "--warmup", "1", "--cpu-sites", "20000", "--cpu-run-seconds", "0.3"])
    assert line["metric"].startswith("windows/sec") and line["unit"] == "windows/s" and line["n_gpus"] == 1
    assert line["dtype"] == "i8" and line["vs_baseline"] is None and line["higher_is_better"] is True
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"}
    # full-size C2: the counters are read in this very run (two rocprofv3 --pmc child passes), and what moved is
    # the algorithmic bytes within a percent
    src = line["roofline"]["traffic_source"]
    assert "measured in this run" in src or "live measurement unavailable" in src, src  # a box without a usable rocprofv3 says so
    assert 1.0 <= line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"] < 1.01
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cpu_model"] and line["cpu_baseline"]["cores"] >= 1
    cb = line["cpu_baseline"]
    assert cb["runs"] == 3 and cb["min"] <= cb["value"] <= cb["max"] and len(cb["wall_s"]) == 3 and min(cb["wall_s"]) >= 0.3
    assert line["config"]["anc_allele_available"] is True
    sp = line["score_path"]
    assert sp["windows"] == line["config"]["windows_total"] and sp["value"] > 0
    assert sp["item_protocol"]["same_bytes_as_native"] is True


def test_launch_carried_events_time_the_pass_and_hand_it_over(eng):
    """sai_plan_set_pass_events: the site pass of a plan stamps its two events in its own dispatch packet.  The
    pipelined scorer uses them as the hand-over to the windows stage and -- in timed steps -- as the pass's
    duration: as many pairs as timed steps, every duration positive and below the wall time of the run, the
    results those of the scorer without the pipeline."""
    import time

    import torch

    from sai_amd.engine import LaunchEvent

    wl = small_job("c2", sites=400_000, chroms=1)
    got = run_rank(eng, wl, 0, 1, overlap=False, through_row=False)[0]
    from sai_amd.resident import ResidentScorer
    from sai_amd.sharding import build_synth_shard

    block, lay, _ = build_synth_shard(eng, wl, 0, 1)
    scorer = ResidentScorer(eng, block, [(s, e) for _, s, e in lay.windows], wl.params(), cap_u=1 << 20, cap_q=1 << 20, overlap=True,
                            window_segment=lay.window_segment)  # fmt: skip
    for _ in range(2):
        scorer.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(25):
        scorer.step(time_counts=True)
    for _ in range(2):
        scorer.step()  # untimed steps after timed ones must not re-stamp the timed pairs
    res = scorer.results()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms = scorer.site_pass_ms()
    assert len(ms) == 25 and all(0.0 < m < wall_ms for m in ms) and sum(ms) < wall_ms, (ms, wall_ms)
    # the pairs behind the durations are a ring per buffer set, read and reused, not one pair per timed step
    pairs = [p for ring in scorer._timing for p in ring]
    assert len(pairs) <= 9 and all(isinstance(a, LaunchEvent) and isinstance(b, LaunchEvent) and not busy for a, b, busy in pairs)
    assert scorer.site_pass_ms() == ms
    scorer.close()
    assert not any(scorer._timing)
    assert res.records.tobytes() == got.records.tobytes() and res.cdd_q.tobytes() == got.cdd_q.tobytes()
    ev = LaunchEvent(eng)  # never stamped: a query says "done" (nothing pending), elapsed_time between unstamped events is an error
    assert ev.query() is True


# ---- the product entry point over several worker processes ----------------------------


def _sai_score(args, out, workers, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, "-m", "sai_amd", "score", *args, "--output", str(out), "--num-workers", str(workers)],
                         cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)  # fmt: skip
    assert res.returncode == 0, res.stderr[-3000:]
    return res


SCORE_INPUTS = {
    "plain": ["--vcf", "tests/data/test.data.vcf", "--chr-name", "21", "--win-len", "10000", "--win-step", "5000",
              "--config", "tests/data/test.uq.config.yaml"],  # fmt: skip
    "bgzip+tbi": ["--vcf", "{tmp}/indexed.vcf.gz", "--chr-name", "21", "--win-len", "10000", "--win-step", "5000",
                  "--config", "tests/data/test.uq.config.yaml"],  # fmt: skip
    "outgroup": ["--vcf", "tests/data/test.with.outgroup.vcf.gz", "--chr-name", "1", "--win-len", "8000", "--win-step", "4000",
                 "--anc-alleles", "tests/data/test.with.outgroup.anc.alleles", "--config", "tests/data/test.with.outgroup.config.yaml"],  # fmt: skip
    "mixed ploidy": ["--vcf", "tests/data/test.mixed.ploidy.data.vcf.gz", "--chr-name", "21", "--win-len", "10000", "--win-step", "5000",
                     "--anc-alleles", "tests/data/test.mixed.ploidy.data.anc.alleles", "--config", "tests/data/test_mixed_ploidy.config.yaml"],  # fmt: skip
}


@pytest.mark.parametrize("case", list(SCORE_INPUTS))
def test_sai_score_with_two_workers_writes_the_one_process_files(case, tmp_path):
    """Plain `python -m sai_amd score ... --num-workers 2` (no launcher on the command line): the process
    starts its two ranks as a child job, both on this box's one GPU with gloo for the gather, each reads
    and scores its own 8 chunks (sai.py:91), rank 0 writes -- TSV, .U.log and .Q.log byte-identical to
    the one-process run of the same command."""
    if case == "bgzip+tbi":
        from test_ingest_native import write_bgzf, write_tbi

        vcf = tmp_path / "indexed.vcf.gz"
        write_bgzf(vcf, open(ROOT / "tests/data/test.data.vcf", "rb").read(), np.random.default_rng(3), max_block=700)
        write_tbi(vcf)
    args = [a.format(tmp=tmp_path) for a in SCORE_INPUTS[case]]
    one, two = tmp_path / "one" / "s.tsv", tmp_path / "two" / "s.tsv"
    _sai_score(args, one, 1)
    _sai_score(args, two, 2, {"SAI_AMD_DIST_BACKEND": "gloo"})
    assert len(one.read_text().splitlines()) > 2
    names = sorted(p.name for p in one.parent.iterdir())
    assert names == sorted(p.name for p in two.parent.iterdir()) and "s.tsv" in names
    for name in names:
        assert (two.parent / name).read_bytes() == (one.parent / name).read_bytes(), name


def test_sai_score_workers_from_the_environment_and_the_function(tmp_path):
    """SAI_AMD_GPUS=2 without the flag, and score(num_workers=3) from Python (uneven shares): same files."""
    args = SCORE_INPUTS["plain"]
    one, env_two, fn_three = tmp_path / "a" / "s.tsv", tmp_path / "b" / "s.tsv", tmp_path / "c" / "s.tsv"
    _sai_score(args, one, 1)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAI_AMD_DIST_BACKEND="gloo", SAI_AMD_GPUS="2")
    res = subprocess.run([sys.executable, "-m", "sai_amd", "score", *args, "--output", str(env_two)], cwd=str(ROOT), env=env,
                         capture_output=True, text=True, timeout=900)  # fmt: skip
    assert res.returncode == 0, res.stderr[-3000:]
    code = ("import sai_amd.stats; from sai_amd.sai import score; score(vcf_file='tests/data/test.data.vcf', chr_name='21', win_len=10000, "
            f"win_step=5000, anc_allele_file=None, output_file={str(fn_three)!r}, config='tests/data/test.uq.config.yaml', num_workers=3)")
    env.pop("SAI_AMD_GPUS")
    res = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    for other in (env_two, fn_three):
        for sfx in (".tsv", ".U.log", ".Q.log"):
            assert other.with_suffix(sfx).read_bytes() == one.with_suffix(sfx).read_bytes(), (other, sfx)


def test_score_rank_on_real_rccl(tmp_path):
    """The sharded route's collectives on RCCL itself with the one rank a one-GPU box can give: process group
    with device_id + the gloo status group, the status exchange, the size all_gather and the padded gather of
    the packed batches in HBM."""
    code = f"""
import os, sys
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch.distributed as dist
import sai_amd.stats
from sai_amd import distributed as D
from sai_amd.sai import chunk_preprocessor_for, load_config, write_headers
from sai_amd.generators import ChunkGenerator
assert D.init_process_group("nccl", force=True) == (0, 1)
assert dist.get_backend() == "nccl" and D._STATUS_GROUP is not None and dist.get_backend(D._STATUS_GROUP) == "gloo"
cfg = load_config("tests/data/test.uq.config.yaml")
out = {str(tmp_path / 'rccl.tsv')!r}
gen = ChunkGenerator(vcf_file="tests/data/test.data.vcf", chr_name="21", window_size=10000, step_size=5000, num_chunks=4)
pre = chunk_preprocessor_for(cfg, "tests/data/test.data.vcf", 10000, 5000, out, None)
write_headers(out, cfg.statistics, cfg.ploidies)
res = D.run_sharded(pre, gen, as_items=False)
assert len(res) == 4
D.shutdown_process_group()
"""
    res = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    one = tmp_path / "one.tsv"
    _sai_score(SCORE_INPUTS["plain"], one, 1)
    for sfx in (".tsv", ".U.log", ".Q.log"):
        assert (tmp_path / "rccl.tsv").with_suffix(sfx).read_bytes() == one.with_suffix(sfx).read_bytes(), sfx
