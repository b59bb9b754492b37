#!/usr/bin/env python3
"""Benchmark of the sliding-window U/Q hot path on MI355X.

    python bench.py                      # N = 1: BASELINE.json configs[2] (C3), the metric's configuration
    python bench.py --workload c2|c4|c5  # the other configs (C4 on one GPU holds 220 GB)
    python bench.py --workload c2x22     # 22 chromosomes of C2's size as ONE block of 22 pieces
    python bench.py --gpus N             # N > 1: configs[3] (C4) over N GPUs; starts its own ranks (a child
                                         # torch.distributed.run on 127.0.0.1) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     # the same job under the driver's launcher

Workloads (BASELINE.json `configs`; all synthetic "synth-v1" data generated
in place in HBM by the counter-based generator, so a shard holds exactly the bytes a one-GPU run
holds at those sites):

  c2  1e6 sites, 200 ref / 200 tgt / 2 src diploids, 50 kb windows every 10 kb, U
  c2x22  22 chromosomes of c2's size, resident as one block of 22 pieces: one pass, one windows stage per step
  c3  1e7 sites, 1,000 / 1,000 / 2, 50 kb every 25 kb, U + Q95            (default for --gpus 1)
  c4  22 chromosomes x 5e6 sites, same populations and windows as c3       (default for --gpus N>1)
  c5  1e7 sites, two source populations of 1 diploid, Q95 sweep over 18 (op, y1, y2) sets

N > 1 is STRONG scaling: the job is fixed, its global window list (the chromosomes' lists end to
end) is cut into N contiguous ranges by the reference's chunk rule (chunk_generator.py:111-142, the
first `n % N` ranks get one window more), each rank generates only the sites of its range plus the
`win_len - win_step` halo, and there is no data-path collective: ONE gather per pass brings the
24-byte window records and the CSR candidate lists to rank 0 (RCCL over xGMI), issued on the stream
the windows stage ran on (`--gather step`, the default; `--gather end` keeps the K passes' rows in
HBM and gathers them once before the closing fence).

One step = one pass of the whole path over the resident block: site pass (site_counts + the fused
per-site decision) -> window_bounds -> window statistics -> copy of the records to pinned host
memory (+ the gather).  Rank 0 prints one JSON line.

`roofline` prices the dominant kernel (site_counts) with HIP events on the launch stream:
algorithmic bytes = resident sites x (n_ref + n_tgt + n_src) genotype bytes per launch of rank 0.
`cpu_baseline` times the numpy oracle (the reference's per-window structure) on the host cores over
a bounded site prefix of the same chromosome, before the GPU is initialised.  `score_path` is the
rate of the product entry point on the same resident block: FeaturePreprocessor.score_windows (the
same fused pass and windows stage) + the native TSV / log writer, next to the item-dictionary route.
"""

from __future__ import annotations

import argparse
import ctypes as C
import gc
import datetime
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

SEED0 = 20260630  # + config number
METRIC = "windows/sec (whole node) + achieved HBM GB/s, 50kb windows over 1e7 sites"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
UQ = dict(w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=True)


def make_workload(name: str, sites=None, chroms=None, scaling: str = "strong", world: int = 1):
    """The SynthWorkload of a BASELINE config; ``sites`` / ``chroms`` scale it down for tests and
    rehearsals (the JSON line then says so)."""
    from sai_amd.sharding import SynthWorkload

    if name == "c2":
        wl = SynthWorkload("c2", SEED0 + 2, [1], int(1e6), 200, 200, [2], 50000, 10000, [dict(UQ)],
                           description="synthetic chr: 1e6 sites, 200 ref/200 tgt/2 src diploids, 50kb/10kb windows, U stat "
                           "(BASELINE.json configs[1])")  # fmt: skip
    elif name == "c2x22":
        # small chromosomes the way a real run meets them: 22 of C2's size resident as ONE block of 22 pieces
        # (sharding.build_synth_shard), one site pass and one windows stage per step -- the ramp and the tail
        # of a launch are paid once per 8.8 GB, not once per 0.4 GB
        wl = SynthWorkload("c2x22", SEED0 + 2, list(range(1, 23)), int(1e6), 200, 200, [2], 50000, 10000, [dict(UQ)],
                           description="22 synthetic chromosomes of C2's size (1e6 sites, 200 ref/200 tgt/2 src diploids, 50kb/10kb "
                           "windows, U stat) as ONE multi-piece block: what BASELINE.json configs[1] looks like inside a whole-genome run")  # fmt: skip
    elif name == "c3":
        wl = SynthWorkload("c3", SEED0 + 3, [1], int(1e7), 1000, 1000, [2], 50000, 25000, [dict(UQ)],
                           description="synthetic chr: 1e7 sites, 1000 ref/1000 tgt/2 src diploids, 50kb/25kb windows, U+Q95 "
                           "(BASELINE.json configs[2])")  # fmt: skip
    elif name == "c4":
        wl = SynthWorkload("c4", SEED0 + 4, list(range(1, 23)), int(5e6), 1000, 1000, [2], 50000, 25000, [dict(UQ)],
                           description="whole-genome synthetic: 22 chroms x 5e6 sites, 1000 ref/1000 tgt/2 src diploids, "
                           "50kb/25kb windows, U+Q95, windows sharded over the GPUs (BASELINE.json configs[3])")  # fmt: skip
    elif name == "c5":
        specs = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[(op, y1), (op, y2)], anc=True)
                 for op in ("=", ">=") for y1 in (0.0, 0.5, 1.0) for y2 in (0.0, 0.5, 1.0)]  # fmt: skip
        wl = SynthWorkload("c5", SEED0 + 5, [1], int(1e7), 1000, 1000, [1, 1], 50000, 25000, specs,
                           description="two source populations (src1+src2, 1 diploid each), Q95 sweep over an 18-set "
                           "(op, y1, y2) grid from one genotype pass, 1e7 sites, 1000 ref/1000 tgt, 50kb/25kb windows "
                           "(BASELINE.json configs[4])")  # fmt: skip
    else:
        raise ValueError(f"unknown workload {name}")
    if scaling == "weak" and world > 1:  # the round-1 form: one chromosome of this size per GPU
        wl.chroms = list(range(1, len(wl.chroms) * world + 1))
        wl.description += f"; WEAK scaling: {len(wl.chroms)} such chromosomes"
    if sites:
        wl.n_sites = int(sites)
        wl.description += f"; REDUCED to {wl.n_sites} sites per chromosome"
    if chroms:
        wl.chroms = list(range(1, int(chroms) + 1))
        wl.description += f"; REDUCED to {len(wl.chroms)} chromosomes"
    return wl


# ------------------------------------------------------------------------------------------
# CPU baseline: the oracle under a process pool over ChunkGenerator-style chunks
# ------------------------------------------------------------------------------------------

_CPU = {}


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a
    GPU box shows all host CPUs but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(int(int(quota) / int(period)), 1))
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                n = min(n, max(q // per, 1))
        except (OSError, ValueError):
            pass
    return n


def cpu_model() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_chunk(bounds):
    from oracle import sai_oracle as O

    d = _CPU
    return len(
        O.run_chunk("1", {"ref": d["ref"]}, {"tgt": d["tgt"]}, d["src"], d["win_len"], d["win_step"], d["stats"],
                    d["ploidies"], d["anc"], start=bounds[0], end=bounds[1])  # fmt: skip
    )


def cpu_baseline(wl, n_sites: int, workers: int, runs: int = 3, min_run_s: float = 5.0) -> dict:
    """windows/s of the numpy oracle on a site prefix of the job's first chromosome (same seed and
    generator as the GPU run).  Runs before any HIP call so that forking the pool is safe."""
    import multiprocessing as mp
    from concurrent.futures import ThreadPoolExecutor

    from oracle import sai_oracle as O
    from sai_amd import _ffi

    lib = _ffi.load()
    chrom = int(wl.chroms[0])

    def host_block(stream, n_ind):
        out = np.empty((n_sites, n_ind), dtype=np.int8)
        step = max(n_sites // (workers * 4), 1)

        def fill(s0):
            n = min(step, n_sites - s0)
            _ffi.check(lib.sai_synth_fill_host(wl.seed, chrom, s0, n, stream, n_ind, wl.ploidy, 0,
                                               out[s0:].ctypes.data_as(C.c_void_p)))  # fmt: skip

        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(fill, range(0, n_sites, step)))
        return out.astype(np.int64)  # the reference's resident layout (utils.py:410)

    gaps = np.empty(n_sites, dtype=np.int32)
    _ffi.check(lib.sai_synth_gaps_host(wl.seed, chrom, 0, n_sites, gaps.ctypes.data_as(C.c_void_p)))
    pos = np.cumsum(gaps).astype(np.int32)
    src_names = ["src"] if len(wl.src_sizes) == 1 else [f"src{i + 1}" for i in range(len(wl.src_sizes))]
    # the product's configuration of this workload: U and Q as the reference configures them (two
    # statistics, each recomputing its frequencies); a sweep contributes its first parameter set
    s0 = wl.specs[0]
    y = {n: yy for n, yy in zip(src_names, s0["y_list"])}
    _CPU.update(
        ref=O.Chrom(pos, host_block(0, wl.n_ref)),
        tgt=O.Chrom(pos, host_block(1, wl.n_tgt)),
        src={n: O.Chrom(pos, host_block(2 + i, k)) for i, (n, k) in enumerate(zip(src_names, wl.src_sizes))},
        win_len=wl.win_len, win_step=wl.win_step, anc=bool(s0["anc"]),
        stats={
            "U": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["x"]}, "src": dict(y)},
            "Q": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["quantile"]}, "src": dict(y)},
        },
        ploidies={"ref": {"ref": wl.ploidy}, "tgt": {"tgt": wl.ploidy}, "src": {n: wl.ploidy for n in src_names}},
    )  # fmt: skip
    windows = O.split_windows([int(pos[0]), int(pos[-1])], wl.win_len, wl.win_step)
    windows = [w for w in windows if w[1] <= int(pos[-1])]  # only windows fully inside the prefix
    chunks = O.split_window_ranges(windows, workers * 8)  # 8 chunks per worker (sai.py:91)
    # three runs of whole passes over the chunk list, each at least `min_run_s` long: one pass of the
    # prefix takes 1.5-2 s, and single passes scattered by +-30 % from call to call
    rates, walls, passes = [], [], []
    with mp.get_context("fork").Pool(workers) as pool:
        for _ in range(runs):
            t0, n_pass = time.perf_counter(), 0
            while True:
                done = sum(pool.map(_cpu_chunk, chunks, chunksize=1))
                assert done == len(windows), (done, len(windows))
                n_pass += 1
                dt = time.perf_counter() - t0
                if dt >= min_run_s:
                    break
            rates.append(n_pass * len(windows) / dt)
            walls.append(dt)
            passes.append(n_pass)
    _CPU.clear()
    order = sorted(rates)
    return {
        "value": round(order[len(order) // 2], 2),  # median of the runs
        "unit": "windows/s",
        "cores": workers,
        "kind": "port",
        "cpu_model": cpu_model(),
        "runs": len(rates),
        "min": round(order[0], 2),
        "max": round(order[-1], 2),
        "wall_s": [round(w, 2) for w in walls],
        "cpu_s": round(sum(walls) * workers, 1),
        "sample": f"first {n_sites} sites of chromosome {chrom} of {wl.name} ({len(windows)} windows, "
        f"{wl.n_ref}/{wl.n_tgt}/{'+'.join(map(str, wl.src_sizes))} diploids, int64 matrices, U+Q95), "
        f"multiprocessing.Pool({workers}) over {len(chunks)} chunks; median of {len(rates)} runs of "
        f"{'/'.join(map(str, passes))} whole passes (>= {min_run_s:g} s each)",
    }


# ------------------------------------------------------------------------------------------
# the timed passes (shared by the GPU run and the CPU rehearsal of tests/test_bench_sharded_cpu.py)
# ------------------------------------------------------------------------------------------


def run_passes(scorer, gather, row_of, steps: int, warmup: int, gather_mode: str, fence, new_ring=None, new_event=None) -> dict:
    """W untimed + K timed passes of ``scorer`` with the per-pass gather of its row.

    ``scorer`` = a ResidentScorer (or None on a rank without windows); ``gather`` = a RowGather;
    ``row_of(k)`` = the uint8 tensor pass k's row is packed into; ``fence()`` = barrier +
    synchronize on both sides of the timed region; ``new_event()`` = an event of the device's current
    stream with ``record()`` / ``elapsed_time()`` (None: host clock only).  Returns the wall time, what
    rank 0 received last (a list of per-rank rows) and what every per-pass gather of the timed passes took:
    ``gather_host_ms`` (the call, on the host) and ``gather_events`` (an event pair around it on the stream
    it was issued on) -- so that ONE record of a multi-GPU run says whether a step that is too long spent
    its time in the collective."""
    layout = gather.layouts[gather.rank]
    dist_on = gather.on
    got = {"rows": None, "timed": False}
    queue: list[int] = []
    gather_host_ms: list = []
    gather_events: list = []

    def gather_row(k: int) -> None:
        pair = (new_event(), new_event()) if (got["timed"] and new_event is not None) else None
        if pair:
            pair[0].record()
        t0 = time.perf_counter()
        got["rows"] = gather.gather(row_of(k))
        if got["timed"]:
            gather_host_ms.append((time.perf_counter() - t0) * 1e3)
        if pair:
            pair[1].record()
            gather_events.append(pair)

    def on_stage(_index: int) -> None:  # on the stream the windows stage ran on, right after it
        k = queue.pop(0)
        if not dist_on:
            return
        scorer.pack_row(row_of(k), layout)
        if gather_mode == "step":
            gather_row(k)

    def one_pass(k: int, timed: bool) -> None:
        if scorer is None:  # a rank without windows still takes part in every collective
            if dist_on and gather_mode == "step":
                gather_row(k)
            return
        queue.append(k)
        scorer.step(time_counts=timed)

    def finish(n_rows: int) -> None:
        if scorer is not None:
            scorer.flush()  # the pipelined form holds the last pass's windows stage back until asked
        if dist_on and gather_mode == "end":
            got["rows"] = new_ring(n_rows)

    if scorer is not None:
        scorer.after_stage = on_stage
    for k in range(warmup):
        one_pass(k, False)
    finish(max(warmup, 1) if warmup else 0)  # also sets up RCCL's channels outside the timed region
    fence()
    got["timed"] = True
    t0 = time.perf_counter()
    for k in range(steps):
        one_pass(k, True)
    finish(steps)
    fence()
    return {"dt": time.perf_counter() - t0, "rows": got["rows"], "gather_host_ms": gather_host_ms, "gather_events": gather_events}


# ------------------------------------------------------------------------------------------
# GPU run
# ------------------------------------------------------------------------------------------


def score_path_rate(eng, wl, block, lay, repeats: int = 3) -> dict:
    """The product entry point on the resident block: FeaturePreprocessor.run_windows (fused site
    pass + windows stage through ResidentScorer, the kernels timed above) + the reference's item
    dictionaries + process_items' TSV / .U.log / .Q.log text, written to a scratch directory."""
    import torch

    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.generators import WindowGenerator
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.sai import write_headers

    src_names = ["src"] if len(wl.src_sizes) == 1 else [f"src{i + 1}" for i in range(len(wl.src_sizes))]
    s0 = wl.specs[0]
    ystr = {n: f"{op}{y:g}" for n, (op, y) in zip(src_names, s0["y_list"])}
    stats = StatConfig({"U": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["x"]}, "src": dict(ystr)},
                        "Q": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["quantile"]}, "src": dict(ystr)}})  # fmt: skip
    ploidies = PloidyConfig({"ref": {"ref": wl.ploidy}, "tgt": {"tgt": wl.ploidy}, "src": {n: wl.ploidy for n in src_names}})
    n = lay.n_sites[0]
    pos_host = block.pos[:n].cpu().numpy()
    wg = WindowGenerator.from_resident(
        str(wl.chroms[0]), pos_host, block.pos[:n], {"ref": _trim(block.pops[0], n)}, {"tgt": _trim(block.pops[1], n)},
        {nm: _trim(p, n) for nm, p in zip(src_names, block.pops[2:])}, wl.win_len, wl.win_step, ploidies,
    )  # fmt: skip
    # `score` gets its blocks from WindowGenerator.device_blocks, whose placement search hands the generator an output
    # arena (memory of another class than the populations': where the scorers of the region write, placement.py); the
    # block of this process was settled when it was built, so its arena -- what the timed scorer has left of it -- goes
    # to the generator by hand
    arena = (block.extra or {}).get("output_arena")
    if arena is not None:
        wg.__dict__["_output_arena"] = arena
    times, plain, first_calls, rows_of_calls = [], [], [], []
    ROW = 12  # calls in a row per repeat: the device needs several calls after an idle stretch to return to its steady rate
    with tempfile.TemporaryDirectory() as tmp:
        out, out_items = os.path.join(tmp, "scores.tsv"), os.path.join(tmp, "items.tsv")
        fp = FeaturePreprocessor(out, stats, anc_allele_available=s0["anc"])
        fp_items = FeaturePreprocessor(out_items, stats, anc_allele_available=s0["anc"])
        n_rows = 0
        for _ in range(repeats + 1):
            # The item route of the repeat before kept the host busy -- and the GPU idle -- for ~80 ms; a device
            # that idle has clocked its memory down, and the first pass after it ran 0.4 ms slower than the same
            # pass in a loop.  `score` never meets the GPU that cold (the ingest runs right before), so every timed
            # call follows an untimed one of the same kind.
            write_headers(out, stats, ploidies)
            fp.score_and_write(wg)
            torch.cuda.synchronize()
            # what `score` runs after the ingest (rows written while later windows are scored), as a run meets it:
            # one call after the other (a chromosome after a chromosome); the first timed call is reported too
            in_a_row = []
            for _k in range(ROW):
                write_headers(out, stats, ploidies)
                t0 = time.perf_counter()
                fp.score_and_write(wg)
                in_a_row.append(time.perf_counter() - t0)
            first_calls.append(in_a_row[0])
            rows_of_calls.append(in_a_row)
            t0, t1 = 0.0, min(in_a_row)
            # the same work as two calls, nothing overlapped (round 4's form), for comparison
            write_headers(out_items, stats, ploidies)
            fp_items.score_windows(wg)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            batch = fp_items.score_windows(wg)
            t3 = time.perf_counter()
            fp_items.write_batches([batch])
            t4 = time.perf_counter()
            plain.append((t4 - t2, t3 - t2, t4 - t3))
            same_plain = all(open(out + sfx, "rb").read() == open(out_items + sfx, "rb").read() for sfx in ("",)) and all(
                open(os.path.join(tmp, f"scores.{k}.log"), "rb").read() == open(os.path.join(tmp, f"items.{k}.log"), "rb").read()
                for k in ("U", "Q"))  # fmt: skip
            write_headers(out_items, stats, ploidies)
            t5 = time.perf_counter()
            items = fp_items.items_from_batch(batch)  # the reference's item protocol, for comparison
            t6 = time.perf_counter()
            fp_items.process_items(items)
            t7 = time.perf_counter()
            times.append((t1 - t0, t6 - t5, t7 - t6))
            n_rows = len(items)
        same = same_plain and all(open(out + sfx, "rb").read() == open(out_items + sfx, "rb").read() for sfx in ("",))
        same = same and all(
            open(os.path.join(tmp, f"scores.{k}.log"), "rb").read() == open(os.path.join(tmp, f"items.{k}.log"), "rb").read()
            for k in ("U", "Q")
        )
        text_bytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.startswith("scores"))
    total, build, items_write = min(times[1:])
    two_calls, gpu, write = min(plain[1:])
    return {
        "value": round(n_rows / total, 1),
        "unit": "windows/s",
        "windows": n_rows,
        "ms_total": round(total * 1e3, 2),
        "ms_first_call_after_idle": round(min(first_calls[1:]) * 1e3, 2),
        # the k-th call of a row (best over the repeats): how the rate returns after the ~100 ms the device idled
        "ms_by_call_in_row": [round(min(r[k] for r in rows_of_calls[1:]) * 1e3, 2) for k in range(ROW)],
        # every row as measured, the untimed repeat first (it ran before any item-route leg of this function)
        "ms_rows": [[round(v * 1e3, 2) for v in r] for r in rows_of_calls],
        "output_arena_bytes_used": None if arena is None else [int(arena.used), int(arena.tensor.numel())],
        "parts": FeaturePreprocessor.PARTS,
        # the same work as score_windows + write_batches, one after the other (round 4's product path)
        "ms_as_two_calls": round(two_calls * 1e3, 2),
        "ms_gpu_score_windows": round(gpu * 1e3, 2),
        "ms_native_text": round(write * 1e3, 2),
        "host_us_per_window": round(write / max(n_rows, 1) * 1e6, 3),
        "output_bytes": text_bytes,
        "item_protocol": {
            "ms_item_dicts": round(build * 1e3, 2),
            "ms_process_items_text": round(items_write * 1e3, 2),
            "windows_per_s": round(n_rows / (gpu + build + items_write), 1),
            "same_bytes_as_native": bool(same),
        },
        "what": "FeaturePreprocessor.score_and_write on the resident block = what `score` runs after the ingest (U and Q as two "
        "statistics, one fused pass per window range, TSV + .U.log + .Q.log written while the later ranges are scored); "
        "ms_as_two_calls = score_windows + write_batches one after the other; item_protocol = the same batch through "
        "items_from_batch + process_items; ms_total = the best of %d x %d calls in a row (chromosome after chromosome; "
        "ms_by_call_in_row = the k-th call of a row), ms_first_call_after_idle = the first of a row, right after one untimed call "
        "(the device has idled ~100 ms through the item route of the repeat before and runs its first passes slower)" % (repeats, ROW),
    }


def measure_traffic(argv: list, kernel: str, timeout_s: float = 90.0):
    """HBM bytes per launch of the dominant kernel, measured NOW: two short child runs of this same
    command under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (each with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes; the program itself follows `--`), corrected as the guide says for
    gfx950 (FETCH_SIZE counts 64 B per 128-byte request of a wide coalesced read: x 2; both in KiB).
    Returns (bytes, how) or (None, why not).  The caller has released its own HBM before."""
    import csv
    import glob
    import shutil
    import subprocess

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    keep = [a for a in argv if a not in ("--traffic",)]
    child = [sys.executable, str(Path(__file__).resolve()), *_without(keep, {"--steps": 1, "--warmup": 1, "--cpu-sites": 1, "--score-path": 1, "--traffic": 1}),
             "--steps", "3", "--warmup", "1", "--cpu-sites", "0", "--score-path", "off", "--traffic", "off"]  # fmt: skip
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", *child]
            try:  # its own session: a run that overstays is ended together with the program it profiles
                # SAI_AMD_PLACEMENT=0: the passes placement.py times while a block is built are launches of the same
                # kernel over two of the three populations -- they would be averaged into the bytes per launch
                proc = subprocess.Popen(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp", "SAI_AMD_PLACEMENT": "0"}, stdout=subprocess.DEVNULL,
                                        stderr=subprocess.DEVNULL, start_new_session=True)  # fmt: skip
            except OSError as exc:
                return None, f"rocprofv3 --pmc {counter}: {type(exc).__name__}"
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None, f"rocprofv3 --pmc {counter}: no result within {timeout_s:.0f} s"
            files = glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {rc})"
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r["Counter_Name"] == counter and f"{kernel}_kernel" in r["Kernel_Name"]]  # fmt: skip
            if not vals:
                return None, f"no {counter} rows for {kernel}"
            got[counter] = sum(vals) / len(vals)
    total = int(got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024)
    return total, (f"measured in this run: two child runs of this command (3 steps) under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; "
                   f"FETCH_SIZE {got['FETCH_SIZE']:.0f} KiB x 1024 x 2 + WRITE_SIZE {got['WRITE_SIZE']:.0f} KiB x 1024 per launch of {kernel}")  # fmt: skip


def _without(argv: list, flags: dict) -> list:
    """argv without the given flags and their values ({flag: number of values})."""
    out, skip = [], 0
    for a in argv:
        if skip:
            skip -= 1
            continue
        if a in flags:
            skip = flags[a]
            continue
        if any(a.startswith(f + "=") for f in flags):
            continue
        out.append(a)
    return out


def source_digest() -> str:
    """sha256 over the sources a stored figure depends on (kernels, C ABI, the resident scorer, the shard
    layout): the GPU box has no git history, so a stored number names the tree it was measured on this way."""
    import hashlib

    h = hashlib.sha256()
    files = sorted((ROOT / "sai_amd" / "csrc").glob("*")) + [ROOT / "include" / "saihip.h"] + [
        ROOT / "sai_amd" / f for f in ("engine.py", "resident.py", "sharding.py", "placement.py", "_ffi.py")]  # fmt: skip
    for f in files:
        if f.is_file():
            h.update(f.name.encode() + b"\0" + f.read_bytes())
    return h.hexdigest()[:16]


def one_gpu_base(wl, args) -> dict:
    """What an N > 1 line is to be divided by: the SAME job on one GPU.  `--gpus 1` runs C3, the
    configuration the metric is quoted on, `--gpus N` the whole-genome job C4 cut into N window ranges,
    so the strong-scaling denominator is C4 on one GPU (`python bench.py --workload c4`: 220 GB
    resident) as measured and kept under profiles/ -- per window the two jobs move the same bytes.
    The stored figure carries the digest of the sources it was measured on (`source_digest`) and the box;
    a figure from another tree is NOT handed out as this tree's base: `value` is then null and the stale
    number stays visible under `stale`."""
    rec = {"workload_id": wl.name, "command": f"python bench.py --workload {wl.name}", "value": None, "unit": "windows/s"}
    f = ROOT / "profiles" / "one_gpu_base.json"
    reduced = bool(args.sites or args.chroms or args.scaling != "strong" or args.layout != "int8" or getattr(args, "missing_per_million", 0))
    if reduced:
        rec["note"] = "reduced / non-default job: run the same arguments with --gpus 1 for the base"
    elif f.exists():
        stored = json.loads(f.read_text()).get(wl.name, {})
        if stored.get("source_digest") == source_digest():
            rec.update(stored)
        elif stored:
            rec["stale"] = stored
            rec["note"] = ("the stored one-GPU figure was measured on other sources than this tree's "
                           f"({stored.get('source_digest')} != {source_digest()}): run the command above on this tree for the base")  # fmt: skip
    return rec


RANK_RECORD_BYTES = 1024


def _placement_ms(block):
    pairs = ((getattr(block, "extra", None) or {}).get("placement") or {}).get("pairs") or []
    return [[p["ms"][0], p["ms_chosen"], "+".join(p["moved"]) or "as built"] for p in pairs if p.get("ms")] or None


def per_rank_figures(dist_on: bool, cdev, mine: dict) -> list:
    """One record per rank, all_gathered after the timed region (outside it) as JSON in fixed-size byte
    rows: the rank's own wall time per step, its site-pass average (HIP events), what its per-pass gather
    took, its windows and sites, and WHERE it ran -- host, local rank, device index, PCI bus id, UUID -- so
    that ONE record of the scaling run shows a straggler, a slow gather, or two ranks on one device."""
    import torch
    import torch.distributed as dist

    if not dist_on:
        return [mine]
    raw = json.dumps(mine).encode()
    if len(raw) > RANK_RECORD_BYTES:
        raise ValueError("per-rank record too long")
    row = torch.zeros((RANK_RECORD_BYTES,), dtype=torch.uint8)
    row[: len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    row = row.to(cdev)
    every = [torch.zeros_like(row) for _ in range(dist.get_world_size())]
    dist.all_gather(every, row)
    return [json.loads(bytes(e.cpu().tolist()).rstrip(b"\0").decode()) for e in every]


def collective_record(backend: str, per_rank: list, backend_seen=None, world_seen=None) -> dict:
    """What the job's process group is, as seen from inside it, and whether its ranks sit on distinct devices."""
    import torch
    import torch.distributed as dist

    rec = {
        "backend": backend_seen if backend_seen is not None else dist.get_backend(),
        "backend_requested": backend,
        "world_size_seen_by_group": world_seen if world_seen is not None else dist.get_world_size(),
        "rccl_version": None,
        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        "hosts": sorted({r.get("hostname") for r in per_rank}),
    }
    try:
        rec["rccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as exc:  # noqa: BLE001 - a CPU build of torch, a rehearsal without the library
        rec["rccl_version"] = f"unavailable ({type(exc).__name__})"
    where = [(r.get("hostname"), r.get("pci_bus_id") or r.get("uuid")) for r in per_rank]
    rec["devices"] = [f"{h}:{d}" for h, d in where]
    rec["distinct_devices"] = len(set(where)) == len(where) and all(d for _, d in where)
    return rec


def static_traffic(wl, args, world: int, sites_rank0: int):
    """(bytes, source) from profiles/traffic.json, or (None, None): the stored counter figure of this job; for
    N > 1 rank 0's share of it (the pass is a stream over the sites a rank holds: bytes scale with them)."""
    tfile = ROOT / "profiles" / "traffic.json"
    if args.traffic == "off" or not tfile.exists() or args.sites or args.chroms or args.scaling != "strong":
        return None, None
    rec = json.loads(tfile.read_text())
    key = wl.name + ("" if args.layout == "int8" else f":{args.layout}") + ("" if args.anc == "true" else ":noanc")
    if key not in rec:
        return None, None
    if rec[key].get("source_digest") != source_digest():  # counters of another tree are not this tree's traffic
        return None, (f"profiles/traffic.json[{key}] was measured on other sources ({rec[key].get('source_digest')} != {source_digest()}): "
                      "not handed out")  # fmt: skip
    whole = rec[key].get("site_counts_hbm_bytes_per_launch")
    src = f"profiles/traffic.json[{key}] ({rec[key].get('source', 'rocprofv3 --pmc passes of this command')}); not measured in this run"
    if world == 1 or whole is None:
        return whole, src
    total_sites = len(wl.chroms) * wl.n_sites
    return int(whole * sites_rank0 / total_sites), src + f"; static: rank 0's share ({sites_rank0} of {total_sites} sites) of the one-GPU job's counters"


def _trim(pop, n_sites):
    from sai_amd.engine import TiledPop

    return TiledPop(pop.tiles, n_sites, pop.n_ind)


def self_launch(n_ranks: int, argv: list, result_out, script: str = "") -> int:
    """`python bench.py --gpus N` without a launcher around it: build the library once, then run the N
    ranks as ONE child process tree (`python -m torch.distributed.run ... bench.py <same arguments>`,
    one rank per GPU, rendezvous on 127.0.0.1) whose stdout is this process's stdout, and return the
    child's exit code.  Nothing here touches the GPU -- no torch.cuda call, no library context -- and
    the child is started with subprocess, never exec'd over this process.  The same launcher starts the
    ranks of `sai score --num-workers N` (sai_amd/launcher.py); the reference starts its workers the
    same way from the parent that owns the task list (mp_pool.py:45-73)."""
    from sai_amd.launcher import launch_ranks

    return launch_ranks(n_ranks, argv, script=script or str(Path(__file__).resolve()), stdout=result_out, who="bench.py")


class HipDevice:
    """Everything of a bench run that touches the GPU, behind one seam: the library context, the resident
    synthetic shard and its scorer, device buffers, synchronisation and the event-timed site pass.
    tests/test_bench_sharded_cpu.py passes its own stand-in (the oracle answering from the same synthetic
    bytes) to ``main`` to run the N > 1 line end to end over gloo; nothing but a test can do that -- there
    is no flag or environment switch for it, and without the library and a gfx950 device this class raises."""

    name = "hip"

    def start(self, local_rank: int) -> None:
        import torch

        from sai_amd.engine import Engine

        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.eng = Engine.get(local_rank)
        self.device = self.eng.device

    def process_group_options(self, backend: str) -> dict:
        import torch

        return {"device_id": torch.device("cuda", self.local_rank)} if backend == "nccl" else {}

    def collective_device(self, backend: str):
        import torch

        return self.device if backend == "nccl" else torch.device("cpu")  # where the small collectives live

    def build(self, wl, rank: int, world: int, args):
        """(block, layout, windows per chromosome, scorer) -- the scorer has run one untimed step: the list
        sizes of a resident block are fixed, the row layout comes from them."""
        from sai_amd.resident import ResidentScorer
        from sai_amd.sharding import build_synth_shard

        block, lay, win_counts = build_synth_shard(self.eng, wl, rank, world)
        scorer = None
        if block is not None:
            scorer = ResidentScorer(self.eng, block, [(s, e) for _, s, e in lay.windows], wl.params(), cap_u=1 << 22,
                                    cap_q=1 << 22, layout=args.layout, overlap=args.overlap != "off",
                                    window_segment=lay.window_segment)  # fmt: skip
            scorer.step()
        return block, lay, win_counts, scorer

    def identity(self) -> dict:
        return self.eng.identity()

    def new_event(self):
        import torch

        ev = torch.cuda.Event(enable_timing=True)
        return ev

    def synchronize(self) -> None:
        import torch

        torch.cuda.synchronize()

    def empty_rows(self, n_rows: int, row_bytes: int):
        import torch

        return torch.empty((n_rows, row_bytes), dtype=torch.uint8, device=self.device)

    def current_stream_context(self):
        import torch

        return torch.cuda.stream(torch.cuda.current_stream())

    def site_pass_ms(self, scorer) -> list:
        return scorer.site_pass_ms()

    def genotype_bytes(self, block, layout: str) -> int:
        return block.genotype_bytes if layout == "int8" else block.packed2_bytes

    def stream_read_probe(self, block) -> float:
        """On-box ceiling: plain 16-B-per-lane streaming read of the ref block (outside the timed region)."""
        buf = block.pops[0].tiles[: min(block.pops[0].tiles.numel(), 10_000_000_000)]
        return self.eng.probe_stream_read(buf)

    def score_path(self, wl, block, lay) -> dict:
        return score_path_rate(self.eng, wl, block, lay)

    def release(self) -> None:
        import torch

        gc.collect()
        torch.cuda.empty_cache()


def main(argv=None, device=None) -> None:
    # stdout carries exactly one line, the JSON record: everything libraries print while the job runs
    # (RCCL's version banner at communicator creation, gloo's rank messages, ...) goes to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    argv = sys.argv[1:] if argv is None else list(argv)
    # the host driver of this pool only supports dmabuf IPC; without this RCCL's communicator set-up fails with
    # `hipIpcGetMemHandle: invalid argument`.  Set here, before anything has initialised HIP, so that ranks started
    # by ANY launcher have it -- the repo's own launcher sets it for its children too (sai_amd/launcher.py)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "c2", "c2x22", "c3", "c4", "c5"], default="auto",
                    help="auto = c3 on one GPU (the configuration the metric is quoted on), c4 on several; "
                    "`--workload c4` on one GPU runs the N > 1 job itself (220 GB resident), so ONE job can be used for every N")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = the fixed job sharded over the GPUs (default); weak = N chromosomes of the "
                    "workload's size, still sharded by contiguous window ranges")
    ap.add_argument("--sites", type=float, default=0, help="sites per chromosome (0 = the workload's own size)")
    ap.add_argument("--chroms", type=int, default=0, help="number of chromosomes (0 = the workload's own)")
    ap.add_argument("--missing-per-million", type=int, default=0,
                    help="missing calls per million genotypes in the synthetic populations (0 = none, the metric's job: the reference's "
                    "reader drops sites with missing calls by default); a side measurement of the stream loop's call-by-call form")
    ap.add_argument("--layout", choices=["int8", "packed2"], default="int8",
                    help="int8 = the SoA int8 block the metric is defined on (default); packed2 = the optional "
                    "2-bit layout (4x fewer genotype bytes; reported with its own algorithmic bytes)")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="pipeline every step's windows stage under the next step's site pass on a second stream "
                    "(auto = on)")
    ap.add_argument("--gather", choices=["step", "end"], default="step",
                    help="N>1: 'step' gathers records + candidate lists to rank 0 after every pass, as a real run "
                    "does (default); 'end' keeps the K passes' rows in HBM and gathers them once before the closing fence")
    ap.add_argument("--anc", choices=["true", "false"], default="true",
                    help="anc_allele_available of every parameter set: true = the headline; false = SURVEY 8(d)'s "
                    "second row (sources also matched against 1 - y, matching sites inverted; stat_utils.py:146-160)")
    ap.add_argument("--traffic", choices=["auto", "live", "static", "off"], default="auto",
                    help="roofline.traffic: live = two short child runs of this command under rocprofv3 --pmc (FETCH_SIZE, "
                    "WRITE_SIZE) after the timed region (each ended after 90 s at the latest); static = the figure kept in "
                    "profiles/traffic.json (N > 1: rank 0's share of it); auto = live on one GPU for a full-size workload "
                    "when rocprofv3 is there, else static")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed runs of the CPU baseline (the median is reported)")
    ap.add_argument("--cpu-run-seconds", type=float, default=5.0, help="minimum length of one CPU run (whole passes)")
    ap.add_argument("--cpu-sites", type=float, default=1e6, help="site prefix timed on the CPU (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="0 = usable cores (affinity mask capped by the cgroup quota)")
    ap.add_argument("--score-path", choices=["auto", "on", "off"], default="auto",
                    help="also time the product entry point (run_windows + items + text) on the resident block "
                    "(auto = on for one GPU and a one-chromosome workload)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only builds and starts the ranks
        sys.exit(self_launch(args.gpus, argv, result_out))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    args.gpus = world
    if device is None:
        device = HipDevice()

    # build first, before anything touches the GPU or joins a process group: hipcc children are
    # forked from a process that has not initialised HIP; concurrent ranks serialise on a file lock
    import __graft_entry__ as entry

    entry.build()

    name = args.workload if args.workload != "auto" else ("c3" if world == 1 else "c4")
    wl = make_workload(name, args.sites, args.chroms, args.scaling, world)
    if hasattr(device, "adapt_workload"):
        device.adapt_workload(wl)
    if args.missing_per_million:
        wl.missing_per_million = int(args.missing_per_million)
        wl.description += f"; {wl.missing_per_million} missing calls per million genotypes (NOT the metric's job)"
    if args.anc == "false":
        for spec in wl.specs:
            spec["anc"] = False
        wl.description += "; anc_allele_available=False (mirror match + inversion)"

    cpu = None
    if rank == 0 and args.cpu_sites > 0:
        # a one-GPU box grants a 16-CPU share of the host whatever nproc says.  In an N > 1 job rank 0 times the
        # baseline HERE, before it joins the process group: the other ranks wait for it in init_process_group
        # (timeout 10 minutes; the baseline takes about half a minute) and their cores are idle meanwhile
        workers = args.cpu_workers or min(usable_cores(), 16)
        cpu = cpu_baseline(wl, min(int(args.cpu_sites), wl.n_sites), workers, args.cpu_runs, args.cpu_run_seconds)
        if world > 1:
            cpu["sample"] += f"; timed by rank 0 of the {world}-rank job before it joined the process group"

    import torch
    import torch.distributed as dist

    if "SAI_BENCH_DEVICE" in os.environ:  # rehearsal knob: every rank on one device of a 1-GPU box
        local_rank = int(os.environ["SAI_BENCH_DEVICE"])
    # rehearsal knob: run the N>1 branches (process group, gather, reductions) with one rank, so that the
    # RCCL calls themselves execute on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("SAI_BENCH_FORCE_DIST"))
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")

    device.start(local_rank)
    backend = os.environ.get("SAI_BENCH_BACKEND", "nccl")  # "gloo" only for rehearsals on a 1-GPU box
    own_group = False
    if dist_on and not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10),
                                **device.process_group_options(backend))  # fmt: skip
        own_group = True
    elif dist_on:
        backend = dist.get_backend()
    from sai_amd.distributed import RowGather
    from sai_amd.sharding import merge_rank_results, plan_shards

    t_setup = time.perf_counter()
    block, lay, win_counts, scorer = device.build(wl, rank, world, args)
    total_windows = int(sum(win_counts))
    overlap = args.overlap != "off"
    row_layout = scorer.row_layout() if scorer is not None else None
    device.synchronize()
    t_setup = time.perf_counter() - t_setup

    cdev = device.collective_device(backend)
    gather = RowGather(row_layout, cdev)
    row_bytes = max(gather.sizes[gather.rank], 1)
    n_rows = max(args.steps, args.warmup, 1) if (dist_on and args.gather == "end") else 1
    ring = device.empty_rows(n_rows, row_bytes)

    def row_of(k: int):
        return ring[k % n_rows]

    def gather_ring(n: int):
        if n == 0:
            return None
        ctx = scorer.window_stream() if scorer is not None else device.current_stream_context()
        with ctx:
            from sai_amd.distributed import gather_padded

            rows = gather_padded(ring[:n, : gather.sizes[gather.rank]].reshape(-1), [n * s for s in gather.sizes])
        if rows is None:
            return None
        return [r.reshape(n, -1)[n - 1] if r.numel() else r for r in rows]  # the last pass's rows

    def fence() -> None:
        device.synchronize()
        if dist_on:
            dist.barrier()
        device.synchronize()

    out = run_passes(scorer, gather, row_of, args.steps, args.warmup, args.gather, fence, gather_ring,
                     getattr(device, "new_event", None))  # fmt: skip
    dt = out["dt"]
    import socket

    site_ms = device.site_pass_ms(scorer) if scorer is not None else []
    on_stream = [a.elapsed_time(b) for a, b in out["gather_events"]]
    avg = lambda xs: round(sum(xs) / len(xs), 4) if xs else None  # noqa: E731
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", 0)), "hostname": socket.gethostname(),
            "ms_per_step_wall": round(dt / max(args.steps, 1) * 1e3, 4), "site_pass_avg_ms": avg(site_ms) or 0.0,
            "gather_avg_ms_on_stream": avg(on_stream), "gather_max_ms_on_stream": round(max(on_stream), 4) if on_stream else None,
            "gather_avg_ms_host_call": avg(out["gather_host_ms"]), "gathers_timed": len(out["gather_host_ms"]),
            "windows": scorer.n_windows if scorer is not None else 0,
            "sites": block.n_real_sites if block is not None else 0, "setup_s": round(t_setup, 2),
            # the placement search of this rank's block (sai_amd/placement.py): the pass over the pair as built and as chosen, ms
            "placement_ms": _placement_ms(block),
            **(device.identity() if hasattr(device, "identity") else {})}  # fmt: skip
    rank_figures = per_rank_figures(dist_on, cdev, mine)
    collective = collective_record(backend, rank_figures) if dist_on else None
    if collective is not None and collective["backend"] == "nccl" and not collective["distinct_devices"]:
        # every rank holds the same records: all leave, with a message from rank 0, and no line is printed
        if rank == 0:
            print(f"bench.py: {world} RCCL rank(s) but not as many distinct devices: {collective['devices']}", file=sys.stderr, flush=True)
        sys.exit(3)
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # what the job produced: on rank 0 the gathered rows of the last pass, merged over the global list
    n_sets = len(wl.specs)
    gather_check = None
    if dist_on:
        res = None
        if rank == 0:
            res = merge_rank_results(gather.decode(out["rows"]), plan_shards(win_counts, world), n_sets)
        # every rank's own results against what rank 0 received from it: CRC of the records, sizes and
        # sums of the candidate lists (outside the timed region)
        import zlib

        own = scorer.results() if scorer is not None else None
        mine = [0, 0, 0, 0, 0, 0]
        if own is not None:
            mine = [own.records.shape[1], zlib.crc32(own.records.tobytes()), int(own.cdd_u.size), int(own.cdd_u.sum()),
                    int(own.cdd_q.size), int(own.cdd_q.sum())]  # fmt: skip
        digest = torch.tensor(mine, dtype=torch.int64, device=cdev)
        digests = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(digests, digest)
        if rank == 0:
            digests = [d.cpu().tolist() for d in digests]
            off = 0
            for r, d in enumerate(digests):
                got = res.records[:, off : off + d[0]]
                assert zlib.crc32(np.ascontiguousarray(got).tobytes()) == d[1], f"gathered records of rank {r} differ from its own"
                off += d[0]
            assert off == res.records.shape[1] == total_windows, (off, res.records.shape, total_windows)
            sums = [sum(d[i] for d in digests) for i in (2, 3, 4, 5)]
            have = [int(res.cdd_u.size), int(res.cdd_u.sum()), int(res.cdd_q.size), int(res.cdd_q.sum())]
            assert sums == have, f"gathered candidate lists differ from the ranks' own: {sums} != {have}"
            gather_check = f"records CRC + candidate-list sizes and sums of {world} rank(s) match what rank 0 received"
    else:
        res = scorer.results()  # also checks the candidate buffers were large enough

    if rank == 0:
        kernel_ms = device.site_pass_ms(scorer)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = device.genotype_bytes(block, args.layout)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        stream_read = device.stream_read_probe(block)
        n_sites_rank0 = block.n_real_sites
        path_bytes = alg_bytes + 4 * n_sites_rank0 + 24 * n_sets * scorer.n_windows
        traffic, traffic_source = static_traffic(wl, args, world, n_sites_rank0)
        score_path = None
        want_sp = args.score_path == "on" or (args.score_path == "auto" and world == 1)
        if want_sp and world == 1 and len(wl.chroms) == 1 and args.layout == "int8":
            try:
                score_path = device.score_path(wl, block, lay)
            except Exception as exc:  # noqa: BLE001 - a secondary figure must not cost the headline line
                score_path = {"error": f"{type(exc).__name__}: {exc}"}
        line = {
            "metric": METRIC,
            "value": round(total_windows * args.steps / dt, 1),
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "i8",  # signed int8 dosages in HBM; sums in integer fields, frequencies and compares in f64
            "data": "synthetic",
            "config": {
                "workload": wl.description,
                "workload_id": wl.name,
                # which job this line ran, and what the other N run by default: a 1 -> N curve over ONE job uses
                # `--workload c4` at every N (on one GPU: 220 GB resident)
                "job": (f"{wl.name}: " + ("the default of --gpus 1 (the metric's configuration)" if wl.name == "c3" and world == 1 else
                                          "the default of --gpus N > 1" if wl.name == "c4" and world > 1 else "chosen with --workload")
                        + "; --gpus 1 defaults to c3, --gpus N > 1 to c4; `--workload c4` runs the N > 1 job at any N incl. 1"),  # fmt: skip
                "layout": args.layout,
                "steps_pipelined": overlap,
                "chromosomes": len(wl.chroms),
                "n_sites_per_chromosome": wl.n_sites,
                "parameter_sets": n_sets,
                "anc_allele_available": args.anc == "true",
                "missing_per_million": int(args.missing_per_million),
                "windows_total": total_windows,
                "windows_rank0": scorer.n_windows,
                "sites_rank0": n_sites_rank0,
                "pieces_rank0": len(lay.pieces),
                "sharding": (
                    f"global window list cut into {world} contiguous ranges (chunk_generator.py:130-142), each rank "
                    f"holds its sites + the win-step halo, no data-path collective; ONE RCCL gather of 24-byte records "
                    f"+ CSR candidate lists to rank 0 per {'pass' if args.gather == 'step' else 'run (rows of all passes)'}"
                    if dist_on else "none"
                ),
                "gather": args.gather if dist_on else None,
                "gather_row_bytes": gather.sizes if dist_on else None,
                "gather_check": gather_check if dist_on else None,
                "collective": collective,
                "one_gpu_base": one_gpu_base(wl, args) if world > 1 else None,
                "per_rank": rank_figures,
                "source_digest": source_digest(),
                # where the big populations of rank 0's block lie relative to each other (sai_amd/placement.py): the passes
                # over the placements it tried, in ms, and whether a population was moved; part of the set-up
                "placement": (getattr(block, "extra", None) or {}).get("placement"),
                "setup_s": round(t_setup, 2),
                "u_sum": int(res.records["u_count"].sum()),
                "q_finite": int(np.isfinite(res.records["q"]).sum()),
                "cdd_u_entries": int(res.cdd_u.size),
                "cdd_q_entries": int(res.cdd_q.size),
            },
            "roofline": {
                "kernel": "site_counts" if args.layout == "int8" else "site_counts_packed2",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(avg_ms, 4),
                # the launches one by one: a mean can hide a few slow ones (a fresh process's first passes)
                "launch_ms_min_median_max": [round(v, 4) for v in (min(site_ms), sorted(site_ms)[len(site_ms) // 2], max(site_ms))] if site_ms else None,
                "launches_over_1p1_median": int(sum(1 for v in site_ms if v > 1.1 * sorted(site_ms)[len(site_ms) // 2])) if site_ms else None,
                "rank": 0,
                "stream_read_probe_gbps": round(stream_read, 1),
                "frac_of_stream_read_probe": round(achieved / stream_read, 4),
                # genotypes + 4 B per position + 24 B per record, over the whole step
                "whole_path_bytes_per_step": path_bytes,
                "whole_path_gbps_this_rank": round(path_bytes / (dt / args.steps) / 1e9, 1),
            },
            "cpu_baseline": cpu,
            # the product entry point on the same block (score_path.value): what a caller of `score` gets per second
            "product_windows_per_s": score_path.get("value") if isinstance(score_path, dict) else None,
            "score_path": score_path,
        }
        reduced = bool(args.sites or args.chroms or args.missing_per_million)
        # never under a profiler: its preloaded library has initialised the GPU in every process of the tree, and a
        # nested rocprofv3 would exec its target from such a process (refused on this pool, for good reason)
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        if profiled and args.traffic in ("auto", "live"):
            line["roofline"]["traffic_source"] = f"{traffic_source or 'none stored'}; not measured live: this run is itself under a profiler"
        elif device.name == "hip" and world == 1 and not dist_on and (args.traffic == "live" or (args.traffic == "auto" and not reduced)):
            # the counters need their own runs (the profiler changes the clock): release this process's HBM first
            kernel = line["roofline"]["kernel"]
            del scorer, block, res
            device.release()
            try:
                live, how = measure_traffic(argv, kernel)
            except Exception as exc:  # noqa: BLE001 - the line must come out whatever the profiler does
                live, how = None, f"{type(exc).__name__}: {exc}"
            if live is not None:
                line["roofline"]["traffic"], line["roofline"]["traffic_source"] = live, how
            else:
                line["roofline"]["traffic_source"] = f"{traffic_source or 'none stored'}; live measurement unavailable ({how})"
        print(json.dumps(line), file=result_out, flush=True)
    if dist_on:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
