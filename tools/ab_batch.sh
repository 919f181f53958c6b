#!/bin/bash
# A/B of the role-interleaved batch path on the GPU box: bash tools/ab_batch.sh name1 name2 ...   ("base" = the in-tree library)
# per library: the batch workload with the interleaved launches and, same library, without PYSP_ROLE_INTERLEAVE=1 (n frame-by-frame passes in one call, the default)
for v in "$@"; do
  if [ "$v" = base ]; then unset PYSP_HIP_LIB; else export PYSP_HIP_LIB=$(pwd)/tools/scratch/$v.so; fi
  for mode in interleaved sequential; do
    if [ $mode = interleaved ]; then export PYSP_ROLE_INTERLEAVE=1; else unset PYSP_ROLE_INTERLEAVE; fi
    for rep in 1 2; do
      python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --workload ahd24b --frames ${FRAMES:-8} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); n=d['config']['frames_per_rank_per_step'] if 'frames_per_rank_per_step' in d['config'] else ${FRAMES:-8}; print('$v', '$mode', 'ms/frame', round(d['ms_per_step']/n,4), 'MP/s', d['value'], d['roofline']['all_kernels_ms'])"
    done
  done
done
unset PYSP_ROLE_INTERLEAVE
