#!/bin/bash
# A/B of the streaming select kernel's schedule knobs on the GPU box: bash tools/ab_stream.sh   (prints ms per step and per kernel)
run() { python bench.py --no-cpu-baseline --steps 100 --warmup 20 --lab-layout packed "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['all_kernels_ms'])"; }
echo "tile: $(run --select-form tile)"
for m in 0 1 3 7; do echo "chunk-per-workgroup maxm=$m: $(PYSP_STREAM_MAXM=$m run --select-form stream)"; done
for d in 1 4; do echo "chunk-per-workgroup div=$d: $(PYSP_STREAM_DIV=$d run --select-form stream)"; done
for m in 0 7; do echo "persistent maxm=$m: $(PYSP_STREAM_PERSIST=1 PYSP_STREAM_MAXM=$m run --select-form stream)"; done
echo "tile: $(run --select-form tile)"
