#!/bin/bash
# A/B on one box: bench.py runs of the given workloads under each value of an environment knob.
# Usage: bash tools/ab_env.sh VAR "v1 v2 ..." "wl1 wl2 ..." [extra bench flags]  -> one line per run
VAR=$1; VALS=$2; WLS=$3; shift; shift; shift
for rep in $(seq 1 ${REPS:-2}); do
for v in $VALS; do
  for wl in $WLS; do
    env $VAR=$v python bench.py --workload $wl --steps 40 --cpu-sites 0 --score-path off --traffic off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v', '$wl', 'step', d['ms_per_step'], 'site', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['u_sum'], d['config']['cdd_q_entries'], 'placement', [(q['ms'][0], q['ms_chosen'], q.get('arena')) for q in ((d['config'].get('placement') or {}).get('pairs') or [])])"
  done
done
done
