#!/bin/bash
# Collects hardware counters for one bench.py workload, one rocprofv3 --pmc pass per counter group (a pass
# holds only counters that fit the hardware together).  A pass combines --pmc with --kernel-trace only (for the kernel
# names), never with the sys/runtime/hip/hsa/memory-copy/marker trace domains.
# Usage on the GPU box (through gpurun):  bash tools/pmc_collect.sh <tag> [bench.py args...]
# Result: gpurun_out/pmc_<tag>/<group>/..._counter_collection.csv, summarised by tools/pmc_summary.py.
set -eo pipefail
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_$tag/g$i
  mkdir -p "$out"
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o p -- \
      python3 "$root/bench.py" --steps 8 --warmup 1 --streams 1 --settle 0 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1)
  echo "pass $i done: $g"
  i=$((i+1))
done
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/pmc_$tag" > "$root/gpurun_out/pmc_${tag}_summary.csv"
# which binary the counters describe: bench.py compares this with the library it loads (roofline.traffic_stale)
sha256sum "${PYSP_HIP_LIB:-$root/pysp_amd/csrc/libpysp_hip.so}" | cut -d" " -f1 > "$root/gpurun_out/pmc_${tag}_lib.sha256"
