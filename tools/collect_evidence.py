#!/usr/bin/env python3
"""gpurun_out/r05 + gpurun_out/prof_r05_* (tools/evidence.sh) -> profiles/r05_* and profiles/manifest.json.

Every bench line names the digest of the sources it ran on (bench.source_digest); a file measured on other sources
than this tree's is refused.  The manifest lists, per kept file, that digest, the box (its stream-read probe) and the
command -- what tests/test_profiles_cpu.py checks."""
import csv, glob, json, os, shutil, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

SRC, DST = ROOT / "gpurun_out" / "r05", ROOT / "profiles"
here = bench.source_digest()
mfile = DST / "manifest.json"
manifest = json.loads(mfile.read_text()) if mfile.exists() else {}


def keep(name: str, text: str, probe, command: str) -> None:
    (DST / name).write_text(text if text.endswith("\n") else text + "\n")
    manifest[name] = {"source_digest": here, "box_stream_read_probe_gbps": probe, "command": command}


probes = {}
for f in sorted(SRC.glob("bench_*.json")):
    raw = f.read_text().strip()
    if not raw:
        continue
    line = json.loads(raw.splitlines()[-1])
    got = line["config"].get("source_digest")
    if got != here:
        sys.exit(f"{f}: measured on sources {got}, this tree is {here}: run tools/evidence.sh on the frozen tree again")
    name = f.stem[len("bench_"):]
    probes[name] = line["roofline"].get("stream_read_probe_gbps")
    base = name[: -len("_second_box")] if name.endswith("_second_box") else name  # (tools/evidence.sh second_box: another gpurun call, another box)
    flags = {"c3": "", "c3_noanc": " --anc false", "c3_packed2": " --layout packed2", "c4": " --workload c4 --steps 20 --cpu-sites 0"}.get(base, f" --workload {base}")
    keep(f"r05_bench_{name}.json", json.dumps(line), probes[name], f"python bench.py{flags}")
box = probes.get("c3")
for src, dst, cmd, head in (
    ("eval_cost.txt", "r05_eval_cost.txt", "python tools/eval_cost.py",
     "# What the per-site decision costs: the stand-alone site_flags kernel (counts from HBM: division + the sets' evaluation) and the lone\n"
     "# fused pass on a C5 block, for 1..18 of C5's parameter sets: predicate table (site_eval.hpp) / set by set (SAI_NO_PRED_TABLE=1).\n"),
    ("plugin_rate.txt", "r05_plugin_rate.txt", "python tools/plugin_rate.py",
     "# The per-window plugin calls, host numpy in, results out (PCIe-inclusive; never the bench value).  Round 5: U and Q of one parameter\n"
     "# set are ONE device call inside FeaturePreprocessor.run, the upload is not followed by a stream synchronisation (round 4: 0.53 / 0.62 ms).\n"),
    ("widened_perf.txt", "r05_widened_perf.txt", "python tools/widened_perf.py",
     "# The widened statistics on a C3-shaped block with a 100-diploid outgroup (1e7 sites, 1000 / 1000 / 2 / 100): kernels of fd / df / Danc /\n"
     "# Dplus and DD, stand-alone and -- round 5 -- DD's per-site terms riding along the site pass (site_pass_dd).\n"),
):
    if (SRC / src).exists():
        body = "\n".join(l for l in (SRC / src).read_text().splitlines() if "amdgpu.ids" not in l and "Warning" not in l and l.strip())
        keep(dst, head + body, box, cmd)
stats = sorted(glob.glob(str(SRC / "widened_trace" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]  # (an earlier run's files may lie beside)
if stats:
    keep("r05_widened_kernel_stats.csv", open(stats[0]).read(), box, "rocprofv3 --kernel-trace --stats -- python3 tools/widened_perf.py")
if (SRC / "rehearse_n2_on_one_gpu.json").exists() and (SRC / "rehearse_n2_on_one_gpu.json").read_text().strip():
    line = json.loads((SRC / "rehearse_n2_on_one_gpu.json").read_text().strip().splitlines()[-1])
    if line["config"]["source_digest"] == here:
        keep("r05_rehearse_n2_on_one_gpu.json", json.dumps(line), line["roofline"].get("stream_read_probe_gbps"),
             "SAI_BENCH_DEVICE=0 SAI_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sites 0 (both ranks on the box's one GPU)")
mfile.write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
# counters of other trees are not this tree's traffic: dropped before this tree's are added
tfile = DST / "traffic.json"
if tfile.exists():
    old = json.loads(tfile.read_text())
    tfile.write_text(json.dumps({k: v for k, v in old.items() if v.get("source_digest") == here}, indent=1) + "\n")
NOTES = {"r05_c5_grid.txt", "r05_dd_pass.txt", "r05_score_parts.txt", "r05_placement.txt", "r05_placement_ab.txt", "r05_c5_product_trace.txt"}  # experiment logs of trees on the way (their headers say so)
manifest = {k: v for k, v in manifest.items() if v["source_digest"] == here and (DST / k).exists() and k not in NOTES}
mfile.write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
# the rocprofv3 runs: kernel trace + PMC passes -> summaries, traffic.json, manifest (tools/summarize_profile.py refuses other trees)
for tag, key, dom in (("r05_c3", "c3", "site_counts"), ("r05_c3_noanc", "c3:noanc", "site_counts"), ("r05_c3_packed2", "c3:packed2", "site_counts_packed2"),
                      ("r05_c2", "c2", "site_counts"), ("r05_c2x22", "c2x22", "site_counts"), ("r05_c5", "c5", "site_counts"), ("r05_c4", "c4", "site_counts")):
    if (ROOT / "gpurun_out" / f"prof_{tag}" / "trace.log").exists():
        res = subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_profile.py"), tag, key, "3", dom], cwd=str(ROOT), capture_output=True, text=True)
        print(tag, "ok" if res.returncode == 0 else res.stderr.strip()[-300:])
# the SQ counters of the c2x22 pass (what its waves spend their cycles on)
manifest = json.loads(mfile.read_text())
for pas in ("pmc_sq", "pmc_sq2"):
    files = sorted(glob.glob(str(ROOT / "gpurun_out" / "prof_r05_c2x22" / pas / "*" / "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
    if files:
        acc = {}
        for r in csv.DictReader(open(files[0])):
            if "site_counts" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
        # the bench's own 13 launches (set-up, 2 warm-up, 10 timed): the passes placement.py times come before them
        acc = {k: [x[1] for x in sorted(v)][-13:] for k, v in acc.items()}
        rows = "\n".join(f"{k},{len(v)},{sum(v) / len(v):.1f}" for k, v in sorted(acc.items()))
        manifest_name = f"r05_c2x22_{pas}.csv"
        (DST / manifest_name).write_text("counter,launches,avg_per_site_counts_launch\n" + rows + "\n")
        manifest[manifest_name] = {"source_digest": here, "box_stream_read_probe_gbps": None, "command": "SAI_PROFILE_SQ=1 tools/profile.sh r05_c2x22 c2x22"}
mfile.write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
if (DST / "r05_bench_c4.json").exists():
    subprocess.run([sys.executable, str(ROOT / "tools" / "update_one_gpu_base.py"), "profiles/r05_bench_c4.json"], cwd=str(ROOT))
print(sorted(manifest))
