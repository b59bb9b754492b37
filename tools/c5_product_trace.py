#!/usr/bin/env python3
"""Where a C5 product call (FeaturePreprocessor.score_and_write, the sweep's first parameter set: long candidate lists) spends its
time, call by call: host time inside ResidentScorer.step / list_totals / results, the merge of the held ranges, the writer.
Successive calls alternate between two levels (bench.py: score_path.ms_by_call_in_row of --workload c5); this prints what differs."""
import os, sys, tempfile, time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main() -> None:
    import torch

    import sai_amd.preprocessors.feature_preprocessor as fpm
    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.engine import Engine
    from sai_amd.generators import WindowGenerator
    from sai_amd.resident import ResidentScorer
    from sai_amd.sai import write_headers
    from sai_amd.sharding import build_synth_shard

    torch.cuda.set_device(0)
    eng = Engine.get(0)
    wl = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c5")
    block, lay, _ = build_synth_shard(eng, wl, 0, 1)
    src_names = ["src"] if len(wl.src_sizes) == 1 else [f"src{i + 1}" for i in range(len(wl.src_sizes))]
    s0 = wl.specs[0]
    ystr = {n: f"{op}{y:g}" for n, (op, y) in zip(src_names, s0["y_list"])}
    stats = StatConfig({"U": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["x"]}, "src": dict(ystr)},
                        "Q": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["quantile"]}, "src": dict(ystr)}})  # fmt: skip
    ploidies = PloidyConfig({"ref": {"ref": wl.ploidy}, "tgt": {"tgt": wl.ploidy}, "src": {n: wl.ploidy for n in src_names}})
    n = lay.n_sites[0]
    wg = WindowGenerator.from_resident(str(wl.chroms[0]), block.pos[:n].cpu().numpy(), block.pos[:n], {"ref": bench._trim(block.pops[0], n)},
                                       {"tgt": bench._trim(block.pops[1], n)}, {nm: bench._trim(p, n) for nm, p in zip(src_names, block.pops[2:])},
                                       wl.win_len, wl.win_step, ploidies)  # fmt: skip
    acc = defaultdict(float)

    def timed(owner, name, label=None):
        inner = getattr(owner, name)

        def outer(*a, **k):
            t0 = time.perf_counter()
            try:
                return inner(*a, **k)
            finally:
                acc[label or name] += (time.perf_counter() - t0) * 1e3

        setattr(owner, name, outer)

    timed(ResidentScorer, "step")
    timed(ResidentScorer, "list_totals")
    timed(ResidentScorer, "results")
    timed(fpm, "_merge_window_ranges")
    timed(fpm, "_rows_per_statistic")
    timed(fpm.FeaturePreprocessor, "_write_combo")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "scores.tsv")
        fp = fpm.FeaturePreprocessor(out, stats, anc_allele_available=s0["anc"])
        for k in range(14):
            write_headers(out, stats, ploidies)
            acc.clear()
            t0 = time.perf_counter()
            fp.score_and_write(wg)
            total = (time.perf_counter() - t0) * 1e3
            scorers = [s for key, s in wg.__dict__.get("_scorers", {}).items()]
            print(f"call {k}: {total:.3f} ms  " + "  ".join(f"{n} {v:.3f}" for n, v in acc.items()) +
                  f"  buffer set of the step {[(s._k - 1) % len(s._flags) for s in scorers]} caps {[(s.cap_u, s.cap_q) for s in scorers]}", flush=True)  # fmt: skip


if __name__ == "__main__":
    main()
