#!/usr/bin/env python3
"""DESIGN.md section 5's table from the kept files of the round (profiles/r05_bench_*.json, *_kernel_durations.csv):
one tree, so the document's figures are generated, not retyped."""
import csv, json
from pathlib import Path

P = Path(__file__).resolve().parents[1] / "profiles"
rows = [("c2", "C2 1e6 × 402, 50k/10k, U"), ("c2x22", "c2x22: 22 chromosomes of C2's size, ONE block of 22 pieces"),
        ("c3", "C3 1e7 × 2002, 50k/25k, U+Q95"), ("c3_noanc", "C3, `--anc false` (mirror match + inversion, SURVEY §8d's second row)"),
        ("c4", "C4 22 × 5e6 × 2002, one GPU"), ("c5", "C5 1e7 × 2002, 2 sources, 18 sets"), ("c3_packed2", "C3, optional packed2 layout")]
print("| config | windows (sets) | ms per step | **windows/s** | `site_counts` avg launch (HIP events) | achieved | frac of 8 TB/s | PMC traffic ÷ algorithmic (in the run) | under rocprofv3: timed avg / min (alone) | CPU baseline (16 workers, median of 3; min–max) | probe of the run |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for key, label in rows:
    f = P / f"r05_bench_{key}.json"
    if not f.exists():
        continue
    d = json.loads(f.read_text())
    r, cb = d["roofline"], d.get("cpu_baseline") or {}
    dur = P / f"r05_{key}_kernel_durations.csv"
    under = "—"
    if dur.exists():
        for row in csv.DictReader(open(dur)):
            if row["kernel"] == r["kernel"]:  # the line's dominant kernel (a packed2 run also holds placement.py's int8 passes)
                under = f"{float(row['avg_us_timed_steps']) / 1e3:.3f} / {float(row['min_us']) / 1e3:.3f} ms"
                break
    traffic = f"{r['traffic'] / 1e9:.3f} GB ÷ {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB = {r['traffic'] / r['algorithmic_bytes_per_launch']:.4f}" if r.get("traffic") else "—"
    cpu = f"{cb['value']:.0f} windows/s ({cb['min']:.0f}–{cb['max']:.0f})" if cb.get("value") else "—"
    sets = d["config"]["parameter_sets"]
    n_w = f"{d['config']['windows_total']:,}".replace(",", " ")
    print(f"| {label} | {n_w}{f' (× {sets})' if sets > 1 else ''} | {d['ms_per_step']:.4g} | **{d['value'] / 1e6:.2f} M** | {r['avg_launch_ms']:.4g} ms | "
          f"{r['achieved'] / 1e3:.2f} TB/s | **{r['frac']:.3f}** | {traffic} | {under} | {cpu} | {r['stream_read_probe_gbps'] / 1e3:.2f} TB/s |")
sp = json.loads((P / "r05_bench_c3.json").read_text())
print("\nscore_path (c3):", json.dumps({k: v for k, v in sp["score_path"].items() if k != "what"}))
print("product_windows_per_s:", sp.get("product_windows_per_s"))
