#!/bin/bash
# Memory-side counters of k_eag beside the zero-compute mover with the same access pattern: L2 (TCC) busy cycles, the
# stalls of its memory-side (EA) write and read queues, request counts.  One rocprofv3 --pmc pass per counter group (TCC: four slots), --kernel-trace
# only beside it, the program itself after "--".   bash tools/pmc_memside.sh <tag>    ->  gpurun_out/pmc_memside_<tag>.csv
set -eo pipefail
tag=${1:-r3}
root=$(pwd)
export TMPDIR=/tmp
groups=(
  "TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE"
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_WRREQ_sum"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_RDREQ_sum"
)
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_memside_$tag/eag_g$i; mkdir -p "$out"
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o p -- \
      python3 "$root/bench.py" --workload eag24ccm --steps 8 --warmup 1 --streams 1 --settle 0 --no-cpu-baseline > "$out/bench.log" 2>&1) || echo "pass $i (k_eag) failed: $g"
  out=$root/gpurun_out/pmc_memside_$tag/probe_g$i; mkdir -p "$out"
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o p -- "$root/tools/ubench_stream.bin" > "$out/probe.log" 2>&1) || echo "pass $i (probe) failed: $g"
  i=$((i+1))
done
python3 - "$root/gpurun_out/pmc_memside_$tag" > "$root/gpurun_out/pmc_memside_$tag.csv" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not name.startswith(("k_eag", "k_mix13t<true>", "k_mix13<1>", "k_copy")):
                continue
            acc[(name, row["Counter_Name"])][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
print("kernel,counter,mean_per_launch,launches")
for (k, c), d in sorted(acc.items()):
    print(f'"{k}",{c},{sum(d.values()) / len(d):.6g},{len(d)}')
PY
cat "$root/gpurun_out/pmc_memside_$tag.csv"
