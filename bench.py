#!/usr/bin/env python3
"""Benchmark of the sliding-window U/Q hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): one synthetic
chromosome per GPU -- 1e7 sites, 1,000 ref / 1,000 tgt / 2 src diploids, 50 kb windows every
25 kb, U (w=0.01, x=0.5, y "=1") and Q95 -- generated in place in HBM by the counter-based
synth-v1 generator (rank r holds chromosome r+1: weak scaling, windows are independent).  One
step = one pass of the whole path over the resident block: site_counts -> site_flags ->
window_bounds -> window_stats -> copy of the per-window records to pinned host memory, plus, for
N > 1, the RCCL gather of all records to rank 0.  Rank 0 prints one JSON line.

`roofline` prices the dominant kernel (site_counts) with HIP events on the launch stream:
algorithmic bytes = n_sites x (n_ref + n_tgt + n_src) genotype bytes per launch.
`cpu_baseline` times the numpy oracle (the reference's per-window structure) on the host
cores over a bounded site prefix of the same chromosome, before the GPU is initialised.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

SEED = 20260633  # 20260630 + config number 3
WIN_LEN, WIN_STEP = 50000, 25000
U_Q_PARAMS = dict(w=0.01, x=0.5, quantile=0.95, y_list=[("=", 1.0)], anc=True)
METRIC = "windows/sec (whole node) + achieved HBM GB/s, 50kb windows over 1e7 sites"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def _cpu_chunk(bounds):
    from oracle import sai_oracle as O

    d = _CPU
    return len(
        O.run_chunk("1", {"ref": d["ref"]}, {"tgt": d["tgt"]}, {"src": d["src"]}, WIN_LEN, WIN_STEP, d["stats"],
                    d["ploidies"], True, start=bounds[0], end=bounds[1])  # fmt: skip
    )


def cpu_baseline(n_sites: int, n_ref: int, n_tgt: int, n_src: int, workers: int) -> dict:
    """windows/s of the numpy oracle on a site prefix of chromosome 1 (same seed and generator as
    the GPU run).  Runs before any HIP call so that forking the pool is safe."""
    import multiprocessing as mp
    from concurrent.futures import ThreadPoolExecutor

    from oracle import sai_oracle as O
    from sai_amd import _ffi

    lib = _ffi.load()

    def host_block(stream, n_ind):
        out = np.empty((n_sites, n_ind), dtype=np.int8)
        step = max(n_sites // (workers * 4), 1)

        def fill(s0):
            n = min(step, n_sites - s0)
            _ffi.check(lib.sai_synth_fill_host(SEED, 1, s0, n, stream, n_ind, 2, 0, out[s0:].ctypes.data_as(C.c_void_p)))

        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(fill, range(0, n_sites, step)))
        return out.astype(np.int64)  # the reference's resident layout (utils.py:410)

    gaps = np.empty(n_sites, dtype=np.int32)
    _ffi.check(lib.sai_synth_gaps_host(SEED, 1, 0, n_sites, gaps.ctypes.data_as(C.c_void_p)))
    pos = np.cumsum(gaps).astype(np.int32)
    _CPU.update(
        ref=O.Chrom(pos, host_block(0, n_ref)),
        tgt=O.Chrom(pos, host_block(1, n_tgt)),
        src=O.Chrom(pos, host_block(2, n_src)),
        stats={
            "U": {"ref": {"ref": 0.01}, "tgt": {"tgt": 0.5}, "src": {"src": ("=", 1.0)}},
            "Q": {"ref": {"ref": 0.01}, "tgt": {"tgt": 0.95}, "src": {"src": ("=", 1.0)}},
        },
        ploidies={"ref": {"ref": 2}, "tgt": {"tgt": 2}, "src": {"src": 2}},
    )
    windows = O.split_windows([int(pos[0]), int(pos[-1])], WIN_LEN, WIN_STEP)
    windows = [w for w in windows if w[1] <= int(pos[-1])]  # only windows fully inside the prefix
    chunks = O.split_window_ranges(windows, workers * 8)  # 8 chunks per worker (sai.py:91)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        done = sum(pool.map(_cpu_chunk, chunks, chunksize=1))
    dt = time.perf_counter() - t0
    _CPU.clear()
    assert done == len(windows), (done, len(windows))
    return {
        "value": round(len(windows) / dt, 2),
        "unit": "windows/s",
        "cores": workers,
        "kind": "port",
        "sample": f"first {n_sites} sites of chromosome 1 ({len(windows)} windows, {n_ref}/{n_tgt}/{n_src} diploids, "
        f"int64 matrices, U+Q95), multiprocessing.Pool({workers}) over {len(chunks)} chunks, {dt:.1f} s",
    }


# ------------------------------------------------------------------------------------------
# GPU run
# ------------------------------------------------------------------------------------------


def main() -> None:
    # stdout carries exactly one line, the JSON record: everything libraries print while the job runs
    # (RCCL's version banner at communicator creation, gloo's rank messages, ...) goes to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=float, default=1e7)
    ap.add_argument("--ref", type=int, default=1000)
    ap.add_argument("--tgt", type=int, default=1000)
    ap.add_argument("--src", type=int, default=2)
    ap.add_argument("--layout", choices=["int8", "packed2"], default="int8",
                    help="int8 = the SoA int8 block the metric is defined on (default); packed2 = the optional "
                    "2-bit layout (4x fewer genotype bytes; reported with its own algorithmic bytes)")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="pipeline every step's windows stage under the next step's site pass on a second stream "
                    "(auto = on)")
    ap.add_argument("--gather", choices=["end", "step"], default="end",
                    help="N>1: 'end' keeps every step's records on the GPU and brings them to rank 0 with ONE RCCL "
                    "gather before the closing fence (inside the timed region); 'step' gathers after every step")
    ap.add_argument("--cpu-sites", type=float, default=4e5, help="site prefix timed on the CPU (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="0 = usable cores (affinity mask capped by the cgroup quota)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if "SAI_BENCH_DEVICE" in os.environ:  # rehearsal knob: every rank on one device of a 1-GPU box
        local_rank = int(os.environ["SAI_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    n_sites = int(args.sites)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sites > 0:
        # a one-GPU box grants a 16-CPU share of the host whatever nproc says
        workers = args.cpu_workers or min(usable_cores(), 16)
        cpu = cpu_baseline(int(args.cpu_sites), args.ref, args.tgt, args.src, workers)

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    # rehearsal knob: run the N>1 branches (process group, gather, reductions) with one rank, so that the
    # RCCL calls themselves execute on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("SAI_BENCH_FORCE_DIST"))
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")

    torch.cuda.set_device(local_rank)
    backend = os.environ.get("SAI_BENCH_BACKEND", "nccl")  # "gloo" only for rehearsals on a 1-GPU box
    if dist_on:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    # one process per node builds (a no-op when the in-tree library is current); the others wait
    if local_rank == 0:
        entry.build()
    if dist_on:
        dist.barrier()
    from sai_amd import _ffi
    from sai_amd.distributed import gather_padded
    from sai_amd.engine import Engine
    from sai_amd.resident import ResidentScorer, default_windows, synth_block

    eng = Engine.get(local_rank)

    chrom = rank + 1
    block = synth_block(eng, SEED, chrom, n_sites, args.ref, args.tgt, [args.src])
    p0, p1 = int(block.pos[0]), int(block.pos[-1])
    windows = default_windows(p0, p1, WIN_LEN, WIN_STEP)
    prm = _ffi.make_params(U_Q_PARAMS["w"], U_Q_PARAMS["x"], U_Q_PARAMS["quantile"], U_Q_PARAMS["y_list"], U_Q_PARAMS["anc"])
    overlap = args.overlap != "off"
    scorer = ResidentScorer(eng, block, windows, [prm], cap_u=1 << 22, cap_q=1 << 22, layout=args.layout,
                            overlap=overlap)
    alg_bytes = block.genotype_bytes if args.layout == "int8" else block.packed2_bytes

    cdev = eng.device if backend == "nccl" else torch.device("cpu")  # where the small collectives live
    # N>1: what travels to rank 0 per step is one row = the window records + both candidate lists
    # (SURVEY.md 8e); their sizes are fixed for a resident block and come from one untimed step
    rec_bytes = scorer.bufs[0].numel()
    n_cdd_u = n_cdd_q = 0
    if dist_on:
        scorer.step()
        first = scorer.results()
        n_cdd_u, n_cdd_q = int(first.cdd_u.size), int(first.cdd_q.size)
    row_bytes = rec_bytes + 4 * (n_cdd_u + n_cdd_q)
    sizes = [row_bytes]
    if dist_on:
        t = torch.tensor(sizes, dtype=torch.int64, device=cdev)
        all_sizes = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_sizes, t)
        sizes = [int(s.item()) for s in all_sizes]

    def fill_row(row) -> None:  # on the stream that produced the records
        row[:rec_bytes].copy_(scorer.bufs[0], non_blocking=True)
        row[rec_bytes : rec_bytes + 4 * n_cdd_u].copy_(scorer.bufs[2][:n_cdd_u].view(torch.uint8), non_blocking=True)
        row[rec_bytes + 4 * n_cdd_u :].copy_(scorer.bufs[3][:n_cdd_q].view(torch.uint8), non_blocking=True)

    # 'end': the rows of every step stay on the GPU and go to rank 0 in one gather before the closing fence
    ring = None
    if dist_on:
        n_rows = max(args.steps, args.warmup, 1) if args.gather == "end" else 1
        ring = torch.empty((n_rows, row_bytes), dtype=torch.uint8, device=eng.device)

    # the scorer calls this on the window stream right after each step's windows stage (in the pipelined
    # form one step late): steps and stages come in the same order, so the rows queue up first-in first-out
    row_queue: list[int] = []

    def on_stage(_index: int) -> None:
        k = row_queue.pop(0)
        if args.gather == "end":
            fill_row(ring[k])
        else:
            fill_row(ring[0])
            gather_padded(ring[0], sizes)

    if dist_on:
        scorer.after_stage = on_stage

    def step(timed: bool, k: int) -> None:
        if dist_on:
            row_queue.append(k)
        scorer.step(time_counts=timed)

    def gather_ring(n_rows: int):
        with scorer.window_stream():
            return gather_padded(ring[:n_rows].reshape(-1), [n_rows * s for s in sizes])

    def fence() -> None:
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(False, k)
    scorer.flush()
    if dist_on and args.gather == "end":
        gather_ring(max(args.warmup, 1))  # also sets up RCCL's point-to-point channels outside the timed region
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(True, k)
    scorer.flush()  # the pipelined form holds the last step's windows stage back until asked
    if dist_on and args.gather == "end":
        gathered = gather_ring(args.steps)
        if rank == 0:
            assert len(gathered) == world and all(g.numel() == args.steps * s for g, s in zip(gathered, sizes))
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        nw = torch.tensor([len(windows)], dtype=torch.int64, device=cdev)
        dist.all_reduce(nw, op=dist.ReduceOp.SUM)
        total_windows = int(nw.item())
    else:
        total_windows = len(windows)

    res = scorer.results()  # also checks the candidate buffers were large enough
    kernel_ms = [a.elapsed_time(b) for a, b in scorer.count_events]
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    # on-box ceiling: plain 16-B-per-lane streaming read of the ref block (outside the timed region)
    stream_read = eng.probe_stream_read(block.pops[0].tiles)

    path_bytes = alg_bytes + 4 * n_sites + 24 * len(windows)
    if rank == 0:
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            rec = json.loads(tfile.read_text())
            key = f"{n_sites}x{args.ref}+{args.tgt}+{args.src}" + ("" if args.layout == "int8" else f":{args.layout}")
            traffic = rec.get(key, {}).get("site_counts_hbm_bytes_per_launch")
        line = {
            "metric": METRIC,
            "value": round(total_windows * args.steps / dt, 1),
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic chr: {n_sites:.0e} sites, {args.ref} ref/{args.tgt} tgt/{args.src} src diploids, "
                "50kb/25kb windows, U+Q95 (BASELINE.json configs[2]); one chromosome per GPU",
                "layout": args.layout,
                "steps_pipelined": overlap,
                "n_sites_per_gpu": n_sites,
                "windows_per_gpu": len(windows),
                "windows_total": total_windows,
                "sharding": f"windows sharded by chromosome, RCCL gather of records + candidate lists to rank 0 ({args.gather})" if dist_on else "none",
                "u_sum_rank0": int(res.records["u_count"].sum()),
                "q_finite_rank0": int(np.isfinite(res.records["q"]).sum()),
            },
            "roofline": {
                "kernel": "site_counts" if args.layout == "int8" else "site_counts_packed2",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(avg_ms, 4),
                "stream_read_probe_gbps": round(stream_read, 1),
                "frac_of_stream_read_probe": round(achieved / stream_read, 4),
                # genotypes + 4 B per position + 24 B per record, over the whole step
                "whole_path_bytes_per_step": path_bytes,
                "whole_path_gbps_this_rank": round(path_bytes / (dt / args.steps) / 1e9, 1),
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), file=result_out, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
