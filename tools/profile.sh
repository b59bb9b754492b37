#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + the two PMC passes of the bench command.
# Usage: bash tools/profile.sh <tag> [extra bench flags]   -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r01}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-sites 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 3
find $OUT -name "*.csv" | head -20
