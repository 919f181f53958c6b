#!/bin/bash
# The CPU suite and 400 random calls against an AddressSanitizer + UBSan build of the oracle (test infrastructure; CPU only -- the GPU pool has no sanitizer).
#   bash tests/oracle_sanitized.sh            builds into a scratch directory, nothing in the tree changes
# A shadow root links tests/, pysp_amd/ ... and holds a COPY of oracle/ with the instrumented libraries; libasan / libubsan are preloaded into python.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
shadow=$(mktemp -d)
trap 'rm -rf "$shadow"' EXIT
cp -r "$root/oracle" "$shadow/oracle"; rm -rf "$shadow/oracle/_ref" "$shadow"/oracle/*.so
for d in tests pysp_amd include tools bench.py __graft_entry__.py; do ln -s "$root/$d" "$shadow/$d"; done
flags="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math"
gcc $flags -o "$shadow/oracle/liboracle.so" "$shadow/oracle/pysp_oracle.c" -lm
gcc $flags -mfma -mavx2 -o "$shadow/oracle/liboracle_fma.so" "$shadow/oracle/pysp_oracle.c" -lm
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
cd "$shadow"
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider --rootdir="$shadow"
python - <<PY
import sys; sys.path.insert(0, "$shadow")
import numpy as np
from oracle import oracle as orc
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.synth import default_wb
assert "$shadow" in orc.lib()._name
M0 = final_matrix(default_wb().get_matrix())
rng = np.random.default_rng(7)
for it in range(400):
    H, W = 2 * int(rng.integers(1, 40)), 2 * int(rng.integers(1, 50))
    bay = rng.normal(0.3, 0.5, (H, W)).astype(np.float32)
    wb = (1.0 / rng.uniform(0.3, 1.0, 3)).astype(np.float32)
    orc.set_lab_mode(int(rng.integers(0, 2)))
    c = it % 8
    if c == 0: orc.demosaic_ahd(bay, wb, M0, bool(it & 8), int(rng.integers(0, 4)))
    elif c == 1: orc.pipeline_srgb(bay, wb, M0, int(rng.integers(0, 3)), bool(it & 8), int(rng.integers(0, 3)), bool(it & 16))
    elif c == 2: orc.demosaic_eag(bay, wb)
    elif c == 3: orc.demosaic_draft(bay, wb)
    elif c == 4:
        K = int(rng.integers(2, 20))
        orc.fuse_raw([rng.random((H, W), dtype=np.float32) for _ in range(K)], [10.0 + k for k in range(K)], wb)
    elif c == 5:
        orc.remap_lanczos4(rng.random((H, W), dtype=np.float32), rng.uniform(-5, W + 5, (H, W)).astype(np.float32), rng.uniform(-5, H + 5, (H, W)).astype(np.float32))
    elif c == 6: orc.warp_table(1.0, 0.01, 0.002, 0.001, 1e-3, -1e-3, W, H, 0.4, 0.6, 1.0)
    else: orc.bayer_normalize((rng.random((H, W)) * 16383).astype(np.uint16), rng.uniform(0, 600, 4).astype(np.float32), rng.uniform(8000, 16383, 4).astype(np.float32))
print("400 random oracle calls under ASan + UBSan: no report")
PY
