#!/bin/bash
# A/B of bench.py --streams N on one box (consecutive independent frames on N HIP streams): bash tools/ab_streams.sh 1 2 3 1 2 3
for n in "$@"; do
  for rep in 1 2; do
    python bench.py --steps 300 --warmup 50 --no-cpu-baseline --streams $n --workload ${WORKLOAD:-ahd24} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $n', d['ms_per_step'], d['roofline']['all_kernels_ms'], d.get('verify',{}) and d['verify'].get('bit_exact'))"
  done
done
