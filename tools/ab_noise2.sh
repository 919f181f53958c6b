#!/bin/bash
# scene and noise timings of the Lab layouts of the in-tree library (automatic policy included) and of a library variant
run() { python bench.py --steps ${STEPS:-400} --warmup 30 --workload ahd24 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['ms_per_step'], d['roofline']['all_kernels_ms'])"; }
for rep in 1 2; do
  unset PYSP_HIP_LIB
  TAG="packed scene"; run --lab-layout packed
  TAG="packed noise"; run --lab-layout packed --scene noise
  TAG="planes scene"; run --lab-layout planes
  TAG="planes noise"; run --lab-layout planes --scene noise
  TAG="auto scene"; run --lab-layout auto
  TAG="auto noise"; run --lab-layout auto --scene noise
done
unset PYSP_HIP_LIB
