#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh) x site-pass waves per CU x workloads.
# Usage: bash tools/ab_lib.sh "variant1 variant2 ... (cur = the tree's own)" "waves..." "workloads..." [bench flags]
VARS=$1; WAVES=$2; WLS=$3; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in $(seq 1 ${REPS:-2}); do
for v in $VARS; do
  LIB=$ROOT/sai_amd/lib/ab/$v/libsaihip.so; [ "$v" = cur ] && LIB=$ROOT/sai_amd/lib/libsaihip.so
  for wv in $WAVES; do
    for wl in $WLS; do
      SAI_AMD_LIB=$LIB SAI_STREAM_WAVES_PER_CU=$wv python $ROOT/bench.py --workload $wl --steps 40 --cpu-sites 0 --score-path off --traffic off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'waves=$wv', '$wl', 'step', d['ms_per_step'], 'site', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['u_sum'], d['config']['cdd_q_entries'])"
    done
  done
done
done
