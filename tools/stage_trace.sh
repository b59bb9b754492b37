#!/bin/bash
# Per-kernel durations of one workload's step, stage not overlapped (so every kernel runs alone) and overlapped.
# Usage: bash tools/stage_trace.sh <tag> <workload> [bench flags]   -> gpurun_out/<tag>_{alone,overlap}.txt
TAG=$1; WL=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in alone overlap; do
  FLAG="--overlap off"; [ $mode = overlap ] && FLAG=""
  OUT=$ROOT/gpurun_out/trace_${TAG}_$mode
  rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --cpu-sites 0 --score-path off --traffic off $FLAG "$@" > $OUT.log 2>&1 || exit 1
  python3 - $OUT > $ROOT/gpurun_out/${TAG}_$mode.txt <<'PY'
import csv, glob, sys, re
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
per = defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+)_kernel\b", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    per[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort(); d = [x[1] for x in v][3:] or [x[1] for x in v]
    print(f"{k:28s} calls {len(v):3d}  avg_us_timed {sum(d)/len(d)/1e3:9.2f}  min {min(d)/1e3:9.2f}  max {max(d)/1e3:9.2f}")
PY
done
