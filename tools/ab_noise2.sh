#!/bin/bash
# scene and noise timings of library variants and of the two Lab layouts of the in-tree library
run() { python bench.py --steps 100 --warmup 30 --workload ahd24 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['ms_per_step'], d['roofline']['all_kernels_ms'])"; }
for rep in 1 2; do
  unset PYSP_HIP_LIB
  TAG="packed scene"; run
  TAG="packed noise"; run --scene noise
  TAG="planes scene"; run --lab-layout planes
  TAG="planes noise"; run --lab-layout planes --scene noise
  export PYSP_HIP_LIB=$(pwd)/tools/scratch/fastfloat.so
  TAG="fastfloat(78 VGPR) scene"; run
  TAG="fastfloat(78 VGPR) noise"; run --scene noise
done
unset PYSP_HIP_LIB
