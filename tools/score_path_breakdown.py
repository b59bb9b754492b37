"""Where the milliseconds of the product's score_windows go on a resident C3 block (ResidentScorer
construction, the pass, results to the host).  GPU box only: python tools/score_path_breakdown.py"""

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
from sai_amd.engine import Engine  # noqa: E402
from sai_amd.resident import ResidentScorer  # noqa: E402
from sai_amd.sharding import build_synth_shard  # noqa: E402


def main() -> None:
    wl = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c3", 0, 0, "strong", 1)
    eng = Engine.get(0)
    block, lay, _ = build_synth_shard(eng, wl, 0, 1)
    windows = [(s, e) for _, s, e in lay.windows]
    sets = wl.params()

    import gc

    gc_log = []
    gc_t = [0.0]

    def on_gc(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], 1e3 * (time.perf_counter() - gc_t[0]), info["collected"]))

    gc.callbacks.append(on_gc)
    if len(sys.argv) > 3 and sys.argv[3] == "freeze":
        gc.collect()
        gc.freeze()
    print("tracked objects:", len(gc.get_objects()), "thresholds:", gc.get_threshold())

    spans = []

    def timed(name, fn):
        def wrapper(*a, **k):
            t = time.perf_counter()
            out = fn(*a, **k)
            dt = 1e3 * (time.perf_counter() - t)
            if dt > 1.0:
                spans.append(f"{name}:{dt:.1f}ms")
            return out

        return wrapper

    torch.Tensor.pin_memory = timed("pin_memory", torch.Tensor.pin_memory)
    torch.full = timed("torch.full", torch.full)
    torch.empty = timed("torch.empty", torch.empty)
    torch.Tensor.to = timed("Tensor.to", torch.Tensor.to)
    torch.cuda.Stream.synchronize = timed("Stream.synchronize", torch.cuda.Stream.synchronize)
    torch.cuda.Event.synchronize = timed("Event.synchronize", torch.cuda.Event.synchronize)
    torch.Tensor.copy_ = timed("copy_", torch.Tensor.copy_)
    for name in ("site_pass", "window_bounds", "window_stats_async"):
        if hasattr(eng, name):
            setattr(eng, name, timed(name, getattr(eng, name)))

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = tick()
        scorer = ResidentScorer(eng, block, windows, sets, cap_u=1 << 16, cap_q=1 << 16)
        t1 = tick()
        e0.record()
        scorer.step()
        e1.record()
        t2 = tick()
        res = scorer.results(grow=True)
        t3 = tick()
        nsnps = res.records[0]["n_sites"].astype("int32")
        t4 = tick()
        print(f"rep {rep}: ctor {1e3 * (t1 - t0):.3f}  step {1e3 * (t2 - t1):.3f}  results {1e3 * (t3 - t2):.3f}  "
              f"nsnps {1e3 * (t4 - t3):.3f}  total {1e3 * (t4 - t0):.3f} ms  step on the GPU {e0.elapsed_time(e1):.3f} ms  ({len(windows)} windows, {nsnps.sum()} site refs)")  # fmt: skip
        del scorer, res
        if spans:
            print("   slow calls:", " ".join(spans))
            spans.clear()
        if gc_log:
            print("   gc:", " ".join(f"gen{g}:{ms:.1f}ms/{n}" for g, ms, n in gc_log))
            gc_log.clear()


if __name__ == "__main__":
    main()
