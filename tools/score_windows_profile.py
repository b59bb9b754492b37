"""cProfile of FeaturePreprocessor.score_windows + write_batches on a resident C3 block (host-side
overhead of the product path around the kernels).  GPU box only."""

import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
from sai_amd.configs import PloidyConfig, StatConfig  # noqa: E402
from sai_amd.engine import Engine  # noqa: E402
from sai_amd.generators import WindowGenerator  # noqa: E402
from sai_amd.preprocessors import FeaturePreprocessor  # noqa: E402
from sai_amd.sai import write_headers  # noqa: E402
from sai_amd.sharding import build_synth_shard  # noqa: E402


def main() -> None:
    wl = bench.make_workload("c3", int(sys.argv[1]) if len(sys.argv) > 1 else 0, 0, "strong", 1)
    eng = Engine.get(0)
    block, lay, _ = build_synth_shard(eng, wl, 0, 1)
    s0 = wl.specs[0]
    ystr = {"src": f"{s0['y_list'][0][0]}{s0['y_list'][0][1]:g}"}
    stats = StatConfig({"U": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["x"]}, "src": dict(ystr)},
                        "Q": {"ref": {"ref": s0["w"]}, "tgt": {"tgt": s0["quantile"]}, "src": dict(ystr)}})  # fmt: skip
    ploidies = PloidyConfig({"ref": {"ref": wl.ploidy}, "tgt": {"tgt": wl.ploidy}, "src": {"src": wl.ploidy}})
    n = lay.n_sites[0]
    pos_host = block.pos[:n].cpu().numpy()
    with tempfile.TemporaryDirectory() as d:
        out = str(Path(d) / "o.tsv")
        fp = FeaturePreprocessor(out, stats, anc_allele_available=s0["anc"])

        wg = WindowGenerator.from_resident(
            str(wl.chroms[0]), pos_host, block.pos[:n], {"ref": bench._trim(block.pops[0], n)},
            {"tgt": bench._trim(block.pops[1], n)}, {"src": bench._trim(block.pops[2], n)}, wl.win_len, wl.win_step, ploidies,
        )  # fmt: skip

        def once():
            t0 = time.perf_counter()
            batch = fp.score_windows(wg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            write_headers(out, stats, ploidies)
            fp.write_batches([batch])
            return t1 - t0, time.perf_counter() - t1

        for _ in range(3):
            once()
        times = [once() for _ in range(10)]
        print("score_windows ms:", " ".join(f"{1e3 * a:.2f}" for a, _ in times))
        print("write ms:        ", " ".join(f"{1e3 * b:.2f}" for _, b in times))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            once()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
