#!/bin/bash
# The round's evidence set, measured on the FROZEN tree (one tree, one evidence set).  Run on the GPU box
# through gpurun, one part per call (a call is limited to 20 minutes); tools/collect_evidence.py then condenses
# gpurun_out/ into profiles/r05_* and refuses anything that was measured on other sources than the tree's.
#   bash tools/evidence.sh benches    the un-profiled bench lines of every workload + the small tools' logs
#   bash tools/evidence.sh profiles   rocprofv3 kernel trace + the two PMC passes of c2, c2x22, c3 (+ noanc, packed2), c5
#   bash tools/evidence.sh second_box c3 / c5 / c2x22 lines from another call (another box of the pool)
#   bash tools/evidence.sh c4         the whole-genome job on one GPU: bench line, profile, N = 2 rehearsal on the one GPU
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
part=${1:-benches}
bench() { # name, flags...
  name=$1; shift
  python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -5 $OUT/bench_$name.err; return 1; }
  python3 -c "
import json; d = json.loads(open('$OUT/bench_$name.json').read()); r = d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'site', r['avg_launch_ms'], 'frac', r['frac'], 'probe', r['stream_read_probe_gbps'], 'traffic', r['traffic'], 'product', d.get('product_windows_per_s'))"
}
case $part in
benches)
  bench c3 && bench c3_noanc --anc false && bench c3_packed2 --layout packed2 && bench c2 --workload c2 && bench c2x22 --workload c2x22 && bench c5 --workload c5
  python3 tools/eval_cost.py > $OUT/eval_cost.txt 2>&1; tail -4 $OUT/eval_cost.txt
  python3 tools/plugin_rate.py > $OUT/plugin_rate.txt 2>&1; grep matrices $OUT/plugin_rate.txt
  python3 tools/widened_perf.py > $OUT/widened_perf.txt 2>&1; tail -12 $OUT/widened_perf.txt
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/widened_trace -- python3 $ROOT/tools/widened_perf.py > $OUT/widened_trace.log 2>&1 || echo "widened trace failed"
  find $OUT/widened_trace -name "*.csv" -size +40M -delete
  ;;
profiles)
  bash tools/profile.sh r05_c3 c3 && bash tools/profile.sh r05_c3_noanc c3 --anc false && bash tools/profile.sh r05_c3_packed2 c3 --layout packed2 \
    && bash tools/profile.sh r05_c2 c2 && SAI_PROFILE_SQ=1 bash tools/profile.sh r05_c2x22 c2x22 && bash tools/profile.sh r05_c5 c5
  ;;
second_box)  # the headline lines once more in another call: the pool's boxes differ by 3-4 % on the same code
  bench c3_second_box && bench c5_second_box --workload c5 && bench c2x22_second_box --workload c2x22
  ;;
c4)
  bench c4 --workload c4 --steps 20 --cpu-sites 0
  bash tools/profile.sh r05_c4 c4
  SAI_BENCH_DEVICE=0 SAI_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sites 0 > $OUT/rehearse_n2_on_one_gpu.json 2> $OUT/rehearse_n2.err || tail -5 $OUT/rehearse_n2.err
  ;;
esac
echo "evidence part $part done"
