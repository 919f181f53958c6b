#!/bin/bash
# A/B of the streaming select kernel (pysp_ctx_set_select_form) against the tile kernel on the GPU box, with its schedule knobs:   bash tools/ab_stream.sh
#   PYSP_STREAM_MAXM  longest chunk = 14 + 16 m quad rows (default 7; 0 = every chunk a single head pass: the tile kernel's work partition through the streaming code)
#   PYSP_STREAM_DIV   divisor of the guided schedule (default 2: chunk length = rows left / (2 x resident workgroups of the queue); 1 = longer first chunks)
# Prints ms per step and per kernel (bench.py, 100 steps, Lab layout pinned to the packed cells the streaming form needs).  Source of profiles/r5_ab_select_stream.log.
run() { python bench.py --no-cpu-baseline --steps 100 --warmup 20 --streams 1 --lab-layout packed "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['all_kernels_ms'])"; }
echo "tile: $(run --select-form tile)"
for m in 0 1 3 7; do echo "stream maxm=$m: $(PYSP_STREAM_MAXM=$m run --select-form stream)"; done
for d in 1 4; do echo "stream div=$d: $(PYSP_STREAM_DIV=$d run --select-form stream)"; done
echo "tile: $(run --select-form tile)"
