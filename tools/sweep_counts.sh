#!/bin/bash
# tuning sweep of the site_counts variants on the C3 block (run on the GPU box)
for v in ${VARIANTS:-0 8 9 2}; do
  for g in ${GRIDS:-64 128}; do
    echo "variant $v grid_mult $g multi=$M: $(SAI_COUNTS_VARIANT=$v SAI_COUNTS_GRID=$g python tools/quick_perf.py 1e7 2>&1 | grep -E 'site_counts|probe' | tr '\n' ' ')"
  done
done
echo "general kernel: $(SAI_COUNTS_MULTI=1 python tools/quick_perf.py 1e7 2>&1 | grep -E 'site_counts')"
