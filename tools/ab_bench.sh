#!/bin/bash
# A/B of library variants on the GPU box: [WORKLOAD=eag24raw] bash tools/ab_bench.sh name1 name2 ...   ("base" = the in-tree library)
for v in "$@"; do
  if [ "$v" = base ]; then unset PYSP_HIP_LIB; else export PYSP_HIP_LIB=$(pwd)/tools/scratch/$v.so; fi
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --no-cpu-baseline --workload ${WORKLOAD:-ahd24} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
  done
done
