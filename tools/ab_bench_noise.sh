#!/bin/bash
# A/B on pure-noise frames (bench.py --scene noise): the worst case of the homogeneity vote, where waves of the packed-layout select kernel take the exact float form
for v in "$@"; do
  if [ "$v" = base ]; then unset PYSP_HIP_LIB; else export PYSP_HIP_LIB=$(pwd)/tools/scratch/$v.so; fi
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --workload ahd24 --scene noise --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'noise', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
  done
done
