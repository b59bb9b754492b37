"""The N > 1 branch of bench.py end to end on CPU: ``gloo`` process groups of 2, 3 and 4 ranks run
bench.run_passes -- shard plan (chunk_generator.py:111-142 applied to the whole job's window list),
per-rank site ranges with the win-step halo, row packing, the per-pass gather (both forms),
decoding and merging on rank 0 -- on tiny synthetic jobs, and what rank 0 ends up with is compared
byte for byte with the one-rank run of the same job.

There is no CPU compute path in the product, so the kernels' place is taken here by a test double
with ResidentScorer's interface that answers every window with the oracle from the SAME synthetic
bytes (the library's host generator).  The GPU counterpart (same job, real kernels, two ranks on the
box's one GPU) is tests/test_hip_sharded.py."""

import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def tiny_workload(name="c4"):
    import bench

    wl = bench.make_workload(name, sites=2600, chroms=5 if name == "c4" else 0)
    wl.n_ref, wl.n_tgt = 12, 10
    wl.win_len, wl.win_step = 3000, 1500  # ~120 sites per window, ~43 windows per chromosome
    wl.missing_per_million = 20000
    for s in wl.specs:  # thresholds that make the tiny populations produce candidates
        s.update(w=0.3, x=0.3, quantile=0.9)
    return wl


class OracleScorer:
    """ResidentScorer's interface (what bench.run_passes uses), computed by the oracle on host
    copies of exactly the sites the shard layout says this rank holds."""

    overlap = False

    def __init__(self, wl, lay, all_pos):
        from sai_amd import _ffi

        lib = _ffi.load_host()
        self.wl, self.lay = wl, lay
        self.n_windows, self.n_sets = len(lay.windows), len(wl.specs)
        self.pieces = []
        for pc, a, n in zip(lay.pieces, lay.site0, lay.n_sites):
            chrom = int(wl.chroms[pc.chrom_index])
            mats = []
            for stream, n_ind in enumerate(wl.pop_sizes):
                m = np.empty((n, n_ind), dtype=np.int8)
                _ffi.check(lib.sai_synth_fill_host(wl.seed, chrom, a, n, stream, n_ind, wl.ploidy, wl.missing_per_million,
                                                   m.ctypes.data_as(C.c_void_p)), lib)  # fmt: skip
                mats.append(m.astype(np.int64))
            self.pieces.append((all_pos[pc.chrom_index][a : a + n], mats))
        self.after_stage = None
        self._res = None
        self._k = 0

    def window_stream(self):
        import contextlib

        return contextlib.nullcontext()

    def flush(self):
        pass

    def step(self, time_counts=False):
        from oracle import sai_oracle as O
        from sai_amd.engine import RECORD_DTYPE, WindowResults

        rec = np.zeros((self.n_sets, self.n_windows), dtype=RECORD_DTYPE)
        lists_u = [[None] * self.n_windows for _ in range(self.n_sets)]
        lists_q = [[None] * self.n_windows for _ in range(self.n_sets)]
        for wi, ((_, start, end), k) in enumerate(zip(self.lay.windows, self.lay.window_segment)):
            pos, mats = self.pieces[k]
            lo, hi = np.searchsorted(pos, start, "left"), np.searchsorted(pos, end, "right")
            sub = [m[lo:hi] for m in mats]
            for si, s in enumerate(self.wl.specs):
                kw = dict(ref_gts=sub[0], tgt_gts=sub[1], src_gts_list=sub[2:], ref_ploidy=self.wl.ploidy,
                          tgt_ploidy=self.wl.ploidy, src_ploidy_list=[self.wl.ploidy] * (len(sub) - 2), pos=pos[lo:hi],
                          w=s["w"], y_list=s["y_list"], anc_allele_available=s["anc"])  # fmt: skip
                u = O.u_stat(x=s["x"], **kw)
                q = O.q_stat(quantile=s["quantile"], **kw)
                _, _, cond = O.matching_loci(sub[0], sub[1], sub[2:], s["w"], s["y_list"], [self.wl.ploidy] * len(sub), s["anc"])
                rec[si, wi] = (hi - lo, u["value"], int(cond.sum()), len(q["cdd_pos"]), q["value"])
                lists_u[si][wi] = np.asarray(u["cdd_pos"], dtype=np.int32)
                lists_q[si][wi] = np.asarray(q["cdd_pos"], dtype=np.int32)
        cat = lambda ll: np.concatenate([a for row in ll for a in row]) if self.n_windows else np.zeros(0, np.int32)  # noqa: E731
        off = np.zeros((self.n_sets, self.n_windows, 2), dtype=np.int64)
        for k, name in enumerate(("u_count", "n_cdd_q")):
            flat = rec[name].reshape(-1).astype(np.int64)
            off[:, :, k] = (np.cumsum(flat) - flat).reshape(self.n_sets, self.n_windows)
        self._res = WindowResults(rec, off, cat(lists_u), cat(lists_q))
        index, self._k = self._k, self._k + 1
        if self.after_stage is not None:
            self.after_stage(index)

    def results(self):
        return self._res

    def row_layout(self):
        from sai_amd import _ffi
        from sai_amd.resident import RowLayout

        m = _ffi.SAI_MAX_SETS
        chunks = [min(m, self.n_sets - s0) for s0 in range(0, self.n_sets, m)]
        nu, nq, s0 = [], [], 0
        for c in chunks:
            nu.append(int(self._res.records["u_count"][s0 : s0 + c].sum()))
            nq.append(int(self._res.records["n_cdd_q"][s0 : s0 + c].sum()))
            s0 += c
        return RowLayout(self.n_sets, self.n_windows, chunks, nu, nq)

    def pack_row(self, row, layout):
        import torch

        raw = self._res.records.tobytes() + self._res.cdd_u.tobytes() + self._res.cdd_q.tobytes()
        assert len(raw) == layout.nbytes
        row[: len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))


def run_job(wl, rank, world, gather_mode, steps=2, warmup=1):
    """What bench.main does between set-up and the JSON line, with the oracle double: returns the
    merged global WindowResults on rank 0."""
    import torch
    import torch.distributed as dist

    import bench
    from sai_amd import _ffi
    from sai_amd.distributed import RowGather, gather_padded
    from sai_amd.sharding import layout_shard, merge_rank_results, piece_site_range, plan_shards, synth_chrom_windows

    all_pos, all_windows = synth_chrom_windows(_ffi.load_host(), wl)
    counts = [len(w) for w in all_windows]
    plan = plan_shards(counts, world)
    lay = layout_shard(plan[rank], wl.chroms, all_windows,
                       lambda pc: piece_site_range(all_pos[pc.chrom_index], all_windows[pc.chrom_index], pc.w0, pc.w1))  # fmt: skip
    scorer = OracleScorer(wl, lay, all_pos) if plan[rank] else None
    layout = None
    if scorer is not None:
        scorer.step()
        layout = scorer.row_layout()
    gather = RowGather(layout, torch.device("cpu"))
    n_rows = max(steps, warmup, 1) if gather_mode == "end" else 1
    ring = torch.zeros((n_rows, max(gather.sizes[gather.rank], 1)), dtype=torch.uint8)

    def gather_ring(n):
        if n == 0:
            return None
        rows = gather_padded(ring[:n, : gather.sizes[gather.rank]].reshape(-1), [n * s for s in gather.sizes])
        return None if rows is None else [r.reshape(n, -1)[n - 1] if r.numel() else r for r in rows]

    def fence():
        if gather.on:
            dist.barrier()

    out = bench.run_passes(scorer, gather, lambda k: ring[k % n_rows], steps, warmup, gather_mode, fence, gather_ring)
    if rank != 0:
        assert out["rows"] is None or not gather.on
        return None, plan, lay
    if not gather.on:  # one rank: the scorer's own results are the job's results
        return merge_rank_results([scorer.results()], plan, len(wl.specs)), plan, lay
    return merge_rank_results(gather.decode(out["rows"]), plan, len(wl.specs)), plan, lay


def _worker(rank, world, port, out_dir, name, gather_mode):
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))  # fmt: skip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_bench_sharded_cpu import run_job, tiny_workload

    res, plan, lay = run_job(tiny_workload(name), rank, world, gather_mode)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), rec=res.records.view(np.uint8), off=res.offsets, u=res.cdd_u, q=res.cdd_q)
    # every rank holds its own range + the halo: neighbouring ranks overlap by < one window of sites
    np.save(os.path.join(out_dir, f"sites{rank}.npy"), np.array([sum(lay.n_sites), len(lay.windows), len(lay.pieces)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,gather_mode", [("c4", 2, "step"), ("c4", 4, "step"), ("c4", 3, "end"), ("c5", 2, "step")])
def test_sharded_job_equals_one_rank_job(tmp_path, name, world, gather_mode):
    import torch.multiprocessing as mp

    wl = tiny_workload(name)
    single, plan1, lay1 = run_job(wl, 0, 1, "step")
    assert single.records.shape == (len(wl.specs), len(lay1.windows)) and single.records["u_count"].sum() > 0
    assert np.isfinite(single.records["q"]).sum() > 0 and single.cdd_q.size > 0
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), name, gather_mode), nprocs=world, join=True)
    got = np.load(tmp_path / "merged.npz")
    assert got["rec"].tobytes() == single.records.tobytes()  # every record, every f64 bit
    assert np.array_equal(got["off"], single.offsets)
    assert got["u"].tobytes() == single.cdd_u.tobytes() and got["q"].tobytes() == single.cdd_q.tobytes()
    # the shards: all windows exactly once, contiguous, balanced; each rank holds only its sites + halo
    sites = [np.load(tmp_path / f"sites{r}.npy") for r in range(world)]
    n_w = [int(s[1]) for s in sites]
    assert sum(n_w) == len(lay1.windows) and max(n_w) - min(n_w) <= 1
    total_sites = len(wl.chroms) * wl.n_sites
    halo = 2 * (wl.win_len // 20)  # synth-v1: one site per ~25 bp; a piece re-reads < win_len of its neighbour
    for s in sites:
        assert int(s[0]) <= total_sites // world + (int(s[2]) + 1) * halo + wl.win_len // 10


def test_plan_shards_matches_the_chunk_rule():
    from sai_amd.sharding import plan_shards
    from sai_amd.utils.windows import split_index_ranges

    for counts in ([5000] * 22, [7, 0, 3, 11], [1], [0, 0], [3, 3, 3]):
        for world in (1, 2, 3, 4, 8, 9):
            plan = plan_shards(counts, world)
            assert len(plan) == world
            flat = [(pc.chrom_index, w) for pieces in plan for pc in pieces for w in range(pc.w0, pc.w1)]
            assert flat == [(c, w) for c, n in enumerate(counts) for w in range(n)]
            sizes = [sum(pc.n_windows for pc in pieces) for pieces in plan]
            want = [b - a for a, b in split_index_ranges(sum(counts), world)] if sum(counts) else []
            assert sizes[: len(want)] == want and not any(sizes[len(want) :])
            for pieces in plan:  # g0 = position in the global list; pieces of a rank are consecutive
                for a, b in zip(pieces, pieces[1:]):
                    assert b.g0 == a.g0 + a.n_windows and b.chrom_index > a.chrom_index and b.w0 == 0


def test_row_layout_round_trip():
    from sai_amd.engine import RECORD_DTYPE
    from sai_amd.resident import RowLayout

    rng = np.random.default_rng(4)
    n_sets, n_w = 18, 7
    rec = np.zeros((n_sets, n_w), dtype=RECORD_DTYPE)
    rec["u_count"] = rng.integers(0, 4, (n_sets, n_w))
    rec["n_cdd_q"] = rng.integers(0, 3, (n_sets, n_w))
    rec["q"] = rng.random((n_sets, n_w))
    u = rng.integers(1, 10**6, int(rec["u_count"].sum())).astype(np.int32)
    q = rng.integers(1, 10**6, int(rec["n_cdd_q"].sum())).astype(np.int32)
    lay = RowLayout(n_sets, n_w, [16, 2], [int(rec["u_count"][:16].sum()), int(rec["u_count"][16:].sum())],
                    [int(rec["n_cdd_q"][:16].sum()), int(rec["n_cdd_q"][16:].sum())])  # fmt: skip
    assert RowLayout.from_header(lay.header() + [0] * 5).header() == lay.header()
    row = np.frombuffer(rec.tobytes() + u.tobytes() + q.tobytes(), dtype=np.uint8)
    res = lay.unpack(row)
    assert res.records.tobytes() == rec.tobytes()
    for s in range(n_sets):
        for w in range(n_w):
            assert len(res.u_list(s, w)) == rec["u_count"][s, w] and len(res.q_list(s, w)) == rec["n_cdd_q"][s, w]
    assert np.array_equal(np.concatenate([res.u_list(s, w) for s in range(n_sets) for w in range(n_w)]), u)
    with pytest.raises(ValueError):
        lay.unpack(row[:-4])


# ---- bench.main itself, N = 2 over gloo: the whole line ------------------------------


class OracleDevice:
    """Stand-in for bench.HipDevice (the one seam of bench.main that touches the GPU): the shard layout is the
    product's, the windows are answered by the oracle from the same synthetic bytes."""

    name = "oracle-double"

    def start(self, local_rank):
        import torch

        self.device = torch.device("cpu")

    def process_group_options(self, backend):
        return {}

    def collective_device(self, backend):
        return self.device

    def adapt_workload(self, wl):
        wl.n_ref, wl.n_tgt = 12, 10
        wl.win_len, wl.win_step = 3000, 1500
        wl.missing_per_million = 20000
        for s in wl.specs:
            s.update(w=0.3, x=0.3, quantile=0.9)

    def build(self, wl, rank, world, args):
        from types import SimpleNamespace

        from sai_amd import _ffi
        from sai_amd.sharding import layout_shard, piece_site_range, plan_shards, synth_chrom_windows

        all_pos, all_windows = synth_chrom_windows(_ffi.load_host(), wl)
        counts = [len(w) for w in all_windows]
        plan = plan_shards(counts, world)
        lay = layout_shard(plan[rank], wl.chroms, all_windows,
                           lambda pc: piece_site_range(all_pos[pc.chrom_index], all_windows[pc.chrom_index], pc.w0, pc.w1))  # fmt: skip
        if not plan[rank]:
            return None, lay, counts, None
        scorer = OracleScorer(wl, lay, all_pos)
        scorer.step()
        n = int(sum(lay.n_sites))
        return SimpleNamespace(n_real_sites=n, genotype_bytes=n * sum(wl.pop_sizes)), lay, counts, scorer

    def synchronize(self):
        pass

    def identity(self):
        return {"device_index": None, "pci_bus_id": f"cpu-double:{os.getpid()}", "uuid": "", "name": "oracle double"}

    def empty_rows(self, n_rows, row_bytes):
        import torch

        return torch.zeros((n_rows, row_bytes), dtype=torch.uint8)

    def current_stream_context(self):
        import contextlib

        return contextlib.nullcontext()

    def site_pass_ms(self, scorer):
        return [0.5, 1.5]

    def genotype_bytes(self, block, layout):
        return block.genotype_bytes

    def stream_read_probe(self, block):
        return 1.0

    def release(self):
        pass


def _bench_main_worker(rank, world, port, out_dir, argv):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SAI_BENCH_BACKEND="gloo")  # fmt: skip
    os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # a foreign launcher's environment: bench.main must set it itself
    fd = os.open(os.path.join(out_dir, f"stdout{rank}.txt"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(fd, 1)
    import bench
    from test_bench_sharded_cpu import OracleDevice

    bench.main(argv, device=OracleDevice())


def test_the_two_rank_line_is_complete(tmp_path):
    """bench.main with two gloo ranks: the N > 1 line carries what the N = 1 line carries -- `roofline` (with a
    stored-counter `traffic` source when the job is full size; a reduced job says none), `cpu_baseline` timed by
    rank 0 before it joined the group, `config.per_rank` (every rank's wall time, site-pass average, windows and
    sites) and `config.one_gpu_base` -- and exactly one line comes out, from rank 0."""
    import json

    import torch.multiprocessing as mp

    argv = ["--gpus", "2", "--workload", "c4", "--sites", "2600", "--chroms", "5", "--steps", "2", "--warmup", "1",
            "--cpu-sites", "2600", "--cpu-runs", "2", "--cpu-run-seconds", "0.2", "--cpu-workers", "2"]  # fmt: skip
    mp.spawn(_bench_main_worker, args=(2, _free_port(), str(tmp_path), argv), nprocs=2, join=True)
    assert (tmp_path / "stdout1.txt").read_text() == ""
    (text,) = [ln for ln in (tmp_path / "stdout0.txt").read_text().splitlines() if ln.startswith("{")]
    line = json.loads(text)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["dtype"] == "i8" and line["steps"] == 2
    cfg = line["config"]
    assert cfg["workload_id"] == "c4" and "c4" in cfg["job"] and "--workload c4" in cfg["job"]
    ranks = cfg["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and sum(r["windows"] for r in ranks) == cfg["windows_total"]
    assert ranks[0]["windows"] == cfg["windows_rank0"] and ranks[0]["sites"] == cfg["sites_rank0"]
    assert all(r["ms_per_step_wall"] > 0 and r["site_pass_avg_ms"] == 1.0 and r["sites"] > 0 for r in ranks)
    assert max(r["ms_per_step_wall"] for r in ranks) <= line["ms_per_step"] * 1.0001 + 1e-3
    # WHERE every rank ran and what its per-pass gather took: the record of the one-shot 8-GPU run
    # must show by itself that the group saw N ranks on N distinct devices, and attribute a slow step
    assert [r["local_rank"] for r in ranks] == [0, 1] and all(r["hostname"] for r in ranks)
    assert len({r["pci_bus_id"] for r in ranks}) == 2 and all(r["pci_bus_id"].startswith("cpu-double:") for r in ranks)
    assert all(r["gathers_timed"] == 2 and r["gather_avg_ms_host_call"] > 0 for r in ranks)  # --steps 2, one gather per pass
    assert all(r["gather_avg_ms_on_stream"] is None for r in ranks)  # the double has no stream events
    coll = cfg["collective"]
    assert coll["backend"] == "gloo" and coll["backend_requested"] == "gloo" and coll["world_size_seen_by_group"] == 2
    assert coll["distinct_devices"] is True and len(coll["devices"]) == 2 and coll["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    base = cfg["one_gpu_base"]
    assert base["workload_id"] == "c4" and base["command"] == "python bench.py --workload c4" and "note" in base
    assert "2 rank(s) match" in cfg["gather_check"] and len(cfg["source_digest"]) == 16
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["algorithmic_bytes_per_launch"] > 0 and roof["rank"] == 0 and "traffic" in roof and roof["avg_launch_ms"] == 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 2 and cb["value"] > 0 and "before it joined the process group" in cb["sample"]
    assert cfg["u_sum"] > 0 and cfg["q_finite"] > 0


def test_three_ranks_report_where_they_ran(tmp_path):
    import json

    import torch.multiprocessing as mp

    argv = ["--gpus", "3", "--workload", "c4", "--sites", "2600", "--chroms", "4", "--steps", "1", "--warmup", "1", "--cpu-sites", "0",
            "--gather", "end"]  # fmt: skip
    mp.spawn(_bench_main_worker, args=(3, _free_port(), str(tmp_path), argv), nprocs=3, join=True)
    (text,) = [ln for ln in (tmp_path / "stdout0.txt").read_text().splitlines() if ln.startswith("{")]
    cfg = json.loads(text)["config"]
    assert [r["rank"] for r in cfg["per_rank"]] == [0, 1, 2] and cfg["collective"]["world_size_seen_by_group"] == 3
    assert cfg["collective"]["distinct_devices"] and all(r["gathers_timed"] == 0 for r in cfg["per_rank"])  # one gather at the end


def test_ranks_that_share_a_device_are_told_apart_from_ranks_that_do_not():
    """collective_record: N RCCL ranks must sit on N distinct (host, PCI bus id) pairs -- the rehearsal shape (every
    rank on device 0) or a rank without an identity reads as NOT distinct, which ends an RCCL run with exit code 3."""
    import bench

    ranks = [{"hostname": "n0", "pci_bus_id": f"0000:{i:02x}:00.0"} for i in range(8)]
    ok = bench.collective_record("nccl", ranks, backend_seen="nccl", world_seen=8)
    assert ok["distinct_devices"] and ok["world_size_seen_by_group"] == 8 and ok["hosts"] == ["n0"] and len(ok["devices"]) == 8
    same = bench.collective_record("nccl", ranks[:3] + [dict(ranks[0])], backend_seen="nccl", world_seen=4)
    assert not same["distinct_devices"]
    two_hosts = bench.collective_record("nccl", [ranks[0], {"hostname": "n1", "pci_bus_id": ranks[0]["pci_bus_id"]}], backend_seen="nccl", world_seen=2)
    assert two_hosts["distinct_devices"]
    assert not bench.collective_record("nccl", [ranks[0], {"hostname": "n0"}], backend_seen="nccl", world_seen=2)["distinct_devices"]


def test_static_traffic_and_one_gpu_base_of_the_full_size_job(monkeypatch):
    """The figures an N > 1 line takes from profiles/: rank 0's share of the stored counters, and the one-GPU
    base only when it was measured on this tree's sources (else null, the stale number kept visible)."""
    import argparse

    import bench

    wl = bench.make_workload("c4")
    args = argparse.Namespace(traffic="auto", sites=0, chroms=0, scaling="strong", layout="int8", anc="true")
    whole, src = bench.static_traffic(wl, args, 1, 110_000_000)
    share, src8 = bench.static_traffic(wl, args, 8, 13_752_000)
    if whole is None:  # the stored counters are another tree's: said so, not handed out
        assert "other sources" in src and share is None
    else:
        assert whole > 2.2e11 and "not measured in this run" in src
        assert share == int(whole * 13_752_000 / 110_000_000) and "rank 0's share" in src8
    assert bench.static_traffic(wl, argparse.Namespace(**{**vars(args), "sites": 1000}), 8, 10) == (None, None)
    monkeypatch.setattr(bench, "source_digest", lambda: "f" * 16)
    none, why = bench.static_traffic(wl, args, 1, 110_000_000)
    assert none is None and "other sources" in why
    monkeypatch.undo()
    base = bench.one_gpu_base(wl, args)
    assert base["workload_id"] == "c4"
    if base["value"] is not None:
        assert base["source_digest"] == bench.source_digest() and base["value"] > 1e6
    else:
        assert base["stale"]["value"] > 1e6 and "other sources" in base["note"]
    monkeypatch.setattr(bench, "source_digest", lambda: "0" * 16)
    stale = bench.one_gpu_base(wl, args)
    assert stale["value"] is None and stale["stale"]["value"] > 1e6
