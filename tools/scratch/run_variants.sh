#!/bin/bash
cp pysp_amd/csrc/libpysp_hip.so /tmp/orig.so
for v in a b c d e; do
  cp tools/scratch/lib_$v.so pysp_amd/csrc/libpysp_hip.so
  echo "variant $v: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["all_kernels_ms"])')"
done
cp tools/scratch/lib_a.so pysp_amd/csrc/libpysp_hip.so
timeout -k 10 200 python -m pytest tests -m gpu -x -q --timeout 150 -p no:cacheprovider 2>&1 | tail -2
cp /tmp/orig.so pysp_amd/csrc/libpysp_hip.so
