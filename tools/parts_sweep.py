#!/usr/bin/env python3
"""FeaturePreprocessor.score_and_write on a resident C3 block for several ways of cutting the region into window
ranges (PARTS / PART_FRACTIONS): ms per call, calls in a row, min / median of 16 after 3 untimed ones.  One process,
one block, the settings interleaved twice so that drift of the box shows.  Usage: python tools/parts_sweep.py"""
import os, statistics, sys, tempfile, time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

SETTINGS = [(3, None), (3, (0.42, 0.84)), (3, (0.45, 0.90)), (3, (0.47, 0.94)), (3, (0.485, 0.97)), (2, (0.90,)), (2, (0.95,)),
            (4, (0.32, 0.64, 0.95)), (3, (0.50, 0.93)), (1, None)]  # fmt: skip


def main() -> None:
    import torch

    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.engine import Engine
    from sai_amd.generators import WindowGenerator
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.sai import write_headers
    from sai_amd.sharding import build_synth_shard

    torch.cuda.set_device(0)
    eng = Engine.get(0)
    wl = bench.make_workload("c3")
    block, lay, _ = build_synth_shard(eng, wl, 0, 1)
    s0 = wl.specs[0]
    ystr = {"src": f"{s0['y_list'][0][0]}{s0['y_list'][0][1]:g}"}
    stats = StatConfig({"U": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["x"]}, "src": dict(ystr)},
                        "Q": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["quantile"]}, "src": dict(ystr)}})  # fmt: skip
    ploidies = PloidyConfig({"ref": {"ref": wl.ploidy}, "tgt": {"tgt": wl.ploidy}, "src": {"src": wl.ploidy}})
    n = lay.n_sites[0]
    pos_host = block.pos[:n].cpu().numpy()
    wg = WindowGenerator.from_resident(str(wl.chroms[0]), pos_host, block.pos[:n], {"ref": bench._trim(block.pops[0], n)},
                                       {"tgt": bench._trim(block.pops[1], n)}, {"src": bench._trim(block.pops[2], n)},
                                       wl.win_len, wl.win_step, ploidies)  # fmt: skip
    ref_bytes = None
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "scores.tsv")
        fp = FeaturePreprocessor(out, stats, anc_allele_available=s0["anc"])
        for rnd in range(2):
            for parts, fractions in SETTINGS:
                FeaturePreprocessor.PARTS, FeaturePreprocessor.PART_FRACTIONS = parts, fractions
                ms = []
                for k in range(19):
                    write_headers(out, stats, ploidies)
                    t0 = time.perf_counter()
                    fp.score_and_write(wg)
                    if k >= 3:
                        ms.append((time.perf_counter() - t0) * 1e3)
                got = b"".join(open(os.path.join(tmp, f), "rb").read() for f in ("scores.tsv", "scores.U.log", "scores.Q.log"))
                ref_bytes = ref_bytes or got
                print(f"round {rnd} parts={parts} fractions={fractions}: min {min(ms):.3f} median {statistics.median(ms):.3f} ms"
                      f"  same bytes: {got == ref_bytes}", flush=True)  # fmt: skip


if __name__ == "__main__":
    main()
