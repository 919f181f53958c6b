#!/bin/bash
cp pysp_amd/csrc/libpysp_hip.so /tmp/orig.so
for v in a b d; do
  cp tools/scratch/lib_$v.so pysp_amd/csrc/libpysp_hip.so
  echo "variant $v: $(python bench.py --workload eag24raw --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["all_kernels_ms"])')"
done
cp /tmp/orig.so pysp_amd/csrc/libpysp_hip.so
