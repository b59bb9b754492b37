#!/bin/bash
# stand-alone (not overlapped) durations of the stage kernels for library variants: tools/stage_alone.sh "v1 v2" workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $1; do
  LIB=$ROOT/sai_amd/lib/ab/$v/libsaihip.so; [ "$v" = cur ] && LIB=$ROOT/sai_amd/lib/libsaihip.so
  export SAI_AMD_LIB=$LIB
  bash $ROOT/tools/stage_trace.sh var_$v $2 > /dev/null 2>&1
  echo "== $v"; grep "window_stats\|window_lists" $ROOT/gpurun_out/var_${v}_alone.txt
done
