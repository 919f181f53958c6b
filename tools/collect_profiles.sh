#!/bin/bash
# One GPU-box call that produces every measurement artefact of a round (copied into profiles/ afterwards):
#   bash tools/collect_profiles.sh r3      (through gpurun; results under gpurun_out/<tag>_final/)
# rocprofv3 runs the program itself (python3 bench.py ...), never a shell or env wrapper; --pmc passes are separate from the stats pass.
set -o pipefail
tag=${1:-r5}
root=$(pwd)
out=$root/gpurun_out/${tag}_final
mkdir -p "$out"
export TMPDIR=/tmp
one() { python3 "$root/bench.py" "$@" 2>>"$out/bench.err" | grep '^{' ; }
part=${2:-all}           # a: bench lines + kernel stats + workloads; b: hardware counters + host-path timings + microbenchmarks (two gpurun calls of <= 20 min each)
if [ "$part" != b ]; then
# 1. the driver's own command shape, then the default run
one --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_20_5.json"
one > "$out/bench.json"
echo "bench done"
# 2. kernel trace + stats of the default command
# (one stream: rocprofv3's per-kernel durations must not be stretched by a kernel of the neighbouring frame -- the timed region itself runs on two by default)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 "$root/bench.py" --no-cpu-baseline --streams 1 > "$out/stats.log" 2>&1)
find "$out/stats" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
echo "stats done"
# 3. every other workload (one JSON line each)
: > "$out/workloads.jsonl"
# (the lines of the BASELINE configs and of the uint16 headline carry their own verify block: no --no-cpu-baseline for those -- VERDICT r3 item 6b)
# (round 5: EVERY workload line carries its own verify block -- VERDICT r4 item 3b: configs 4 and 5 too, the warp under the classified bar of oracle/checks.py)
for w in ahd24b ahd24u16 eag24ccm draft12 cfg3 eag24 eag24ccmu16 eag24raw draft12ccm draft12raw fuse45 warp100 cfg5; do one --workload $w >> "$out/workloads.jsonl"; done
one --streams 1 >> "$out/workloads.jsonl"                        # the headline on ONE stream (rounds 1-4's timed region)
one --select-form stream --lab-layout packed >> "$out/workloads.jsonl"      # the streaming form of the select kernel (round 5: bit-exact, not faster)
one --lab-mode closed_form >> "$out/workloads.jsonl"
# the Lab layouts of the select kernel (round 4): round 3's float planes on the scene; pure noise with the automatic policy, the packed form and the planes form
one --lab-layout planes >> "$out/workloads.jsonl"
for l in auto packed planes; do one --scene noise --lab-layout $l >> "$out/workloads.jsonl"; done
one --gpus 2 --backend gloo --workload cfg5 --steps 5 --warmup 2 >> "$out/workloads.jsonl"
one --gpus 2 --backend gloo --workload cfg3 --steps 20 --warmup 3 >> "$out/workloads.jsonl"
echo "workloads done"
fi
if [ "$part" = a ]; then echo "part a done"; exit 0; fi
# 4. hardware counters of the default workload and of the EAG / Draft / warp kernels
# (the Lab layout is pinned: under the automatic policy which select instance runs depends on the call history, and the summaries would mix two kernels -- ADVICE r4)
bash "$root/tools/pmc_collect.sh" ${tag}_ahd24 --lab-layout packed > "$out/pmc_ahd24.log" 2>&1
bash "$root/tools/pmc_collect.sh" ${tag}_eag24ccm --workload eag24ccm > "$out/pmc_eag.log" 2>&1
bash "$root/tools/pmc_collect.sh" ${tag}_draft12 --workload draft12 > "$out/pmc_draft.log" 2>&1
bash "$root/tools/pmc_collect.sh" ${tag}_warp100 --workload warp100 > "$out/pmc_warp.log" 2>&1
for k in ahd24 eag24ccm draft12 warp100; do cp "$root/gpurun_out/pmc_${tag}_${k}_summary.csv" "$out/pmc_${k}_summary.csv"; cp "$root/gpurun_out/pmc_${tag}_${k}_lib.sha256" "$out/pmc_${k}_lib.sha256"; done
echo "pmc done"
# 5. PCIe-inclusive timings of the drop-in API, whole configs 4 / 5 on one GPU
python3 "$root/tools/dropin_time.py" > "$out/dropin_time.log" 2>&1
python3 "$root/tools/dropin_probe.py" > "$out/dropin_probe.log" 2>&1
python3 "$root/tools/config_time.py" > "$out/config_time.log" 2>&1
python3 "$root/tests/ref_native_time.py" gpu > "$out/native_units_gpu.log" 2>&1
# 6. the microbenchmarks behind the issue-cost model and the stream ceilings (binaries built by `make -C tools` / hipcc before the call)
[ -x "$root/tools/ubench_valu3.bin" ] && "$root/tools/ubench_valu3.bin" > "$out/ubench_valu.log" 2>&1
[ -x "$root/tools/ubench_valu4.bin" ] && timeout -k 10 150 "$root/tools/ubench_valu4.bin" > "$out/ubench_pairs.log" 2>&1
[ -x "$root/tools/ubench_stream.bin" ] && "$root/tools/ubench_stream.bin" > "$out/ubench_stream.log" 2>&1
echo "all done"
