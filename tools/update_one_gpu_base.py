#!/usr/bin/env python3
"""profiles/one_gpu_base.json from a one-GPU bench line of the N > 1 job (`python bench.py --workload c4`):
the figure every `--gpus N` line names as its strong-scaling denominator, tied to the digest of the sources
it was measured on (bench.source_digest) -- a line from another tree is handed out as `stale`, not as the base.
Usage: python tools/update_one_gpu_base.py profiles/r04_bench_c4.json ["note about the box"]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = Path(sys.argv[1])
line = json.loads(src.read_text())
assert line["n_gpus"] == 1 and line["config"]["workload_id"] == "c4" and "REDUCED" not in line["config"]["workload"], "need the full-size one-GPU C4 line"
rec = {
    "value": line["value"],
    "ms_per_step": line["ms_per_step"],
    "site_counts_frac_of_peak": line["roofline"]["frac"],
    "windows": line["config"]["windows_total"],
    "steps": line["steps"],
    "source_digest": line["config"]["source_digest"],
    "source": f"{src.as_posix()} (python bench.py --workload c4 --steps {line['steps']}, one MI355X, {line['config']['windows_total']} windows, "
              f"{line['roofline']['algorithmic_bytes_per_launch'] / 1e9:.1f} GB resident)",
    "box": (sys.argv[2] if len(sys.argv) > 2 else "") + f" stream-read probe {line['roofline']['stream_read_probe_gbps']} GB/s in the same run",
}
out = ROOT / "profiles" / "one_gpu_base.json"
out.write_text(json.dumps({"c4": rec}, indent=2) + "\n")
print(json.dumps(rec, indent=1))
