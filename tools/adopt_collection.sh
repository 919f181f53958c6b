#!/bin/bash
# Copies the artefacts of the last `tools/collect_profiles.sh <tag>` call (gpurun_out/<tag>_final/) into profiles/ under the round's names and regenerates
# profiles/traffic.json:   bash tools/adopt_collection.sh r3
set -e
tag=${1:-r5}
F=gpurun_out/${tag}_final
cp $F/bench.json profiles/${tag}_bench.json
cp $F/bench_driver_20_5.json profiles/${tag}_bench_driver_20_5.json
cp $F/kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $F/workloads.jsonl profiles/${tag}_workloads.jsonl
for k in ahd24 eag24ccm draft12 warp100; do cp $F/pmc_${k}_summary.csv profiles/${tag}_pmc_${k}_summary.csv; cp $F/pmc_${k}_lib.sha256 profiles/${tag}_pmc_${k}_lib.sha256; done
for f in dropin_time dropin_probe config_time native_units_gpu ubench_valu ubench_pairs ubench_stream gputest; do [ -f $F/$f.log ] && cp $F/$f.log profiles/${tag}_$f.log; done
python3 tools/make_traffic.py $tag > /dev/null
python3 tools/make_isa_mix.py > /dev/null
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench", "bench_driver_20_5"):
    d = json.loads(open(f"profiles/{tag}_{f}.json").read())
    print(f, d["ms_per_step"], d["value"], d["roofline"]["all_kernels_ms"], d["roofline"]["frac"], d["verify"]["bit_exact"], d["cpu_baseline"]["value"])
for l in open(f"profiles/{tag}_workloads.jsonl"):
    d = json.loads(l)
    print(d["n_gpus"], d["config"]["workload"][:64], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"]["all_kernels_ms"])
t = json.load(open("profiles/traffic.json"))
for w, v in t.items():
    if w[0] != "_":
        for k, e in v.items():
            print(w, k, e["hbm_bytes"], round(e["valu_insts"] * 64 / e["px"], 1), e["lib_sha256"][:12])
PY
