#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + the two PMC passes of the bench command.
# Usage: bash tools/profile.sh <tag> <workload> [extra bench flags]   -> gpurun_out/prof_<tag>/...
# Counters are collected in their own passes (--pmc with --kernel-trace only), as the guide's
# HBM / rocprofv3 section prescribes; the program itself follows `--` (no env/bash hop).
set -o pipefail
TAG=${1:-r02_c3}
WL=${2:-c3}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --cpu-sites 0 --score-path off --traffic static $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 3
# keep what the summariser reads (the raw traces of a C4 run are large)
find $OUT -name "*.csv" -size +40M -delete
find $OUT -name "*.csv" | head -20
# optional fourth pass: shader-engine counters of the dominant kernel (what the wavefronts spend their
# cycles on) -- SAI_PROFILE_SQ=1 bash tools/profile.sh ...
if [ -n "$SAI_PROFILE_SQ" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || echo "SQ counter pass failed (see pmc_sq.log)"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 || echo "SQ counter pass 2 failed (see pmc_sq2.log)"
fi
find $OUT -name "*.csv" -size +40M -delete
