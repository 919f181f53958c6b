#!/usr/bin/env python3
"""profiles/<tag>_pmc_<workload>_summary.csv (tools/pmc_collect.sh + tools/pmc_summary.py) -> profiles/traffic.json, the per-kernel
HBM bytes and wave-level VALU instruction counts bench.py reports as roofline.traffic / roofline.valu:  python tools/make_traffic.py r2"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
PX = {"ahd24": 4000 * 6000, "eag24ccm": 4000 * 6000, "draft12": 3000 * 4000, "warp100": 8736 * 11648}
out = {"_format": "per workload: kernel -> {hbm_bytes, valu_insts (wave-level VALU instructions), px (pixels per launch), source}; means per launch from "
                  "tools/pmc_collect.sh (rocprofv3 --pmc, one counter group per pass, --kernel-trace only beside it); hbm_bytes = FETCH_SIZE*1024*2 + "
                  "WRITE_SIZE*1024 (FETCH_SIZE doubled per the gfx950 calibration in tools/ubench_fetch.hip / MI355X_MICROARCH.md section HBM)"}
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_pmc_*_summary.csv"))):
    wl = re.match(rf"{tag}_pmc_(.*)_summary.csv", os.path.basename(path)).group(1)
    sha_path = path.replace("_summary.csv", "_lib.sha256")
    sha = open(sha_path).read().strip() if os.path.exists(sha_path) else None
    rows = {}
    for r in csv.DictReader(open(path)):
        k = r["kernel"].split("<")[0].split("(")[0].strip()
        if k.startswith("k_"): rows.setdefault(k, {})[r["counter"]] = float(r["mean_per_launch"])
    out[wl] = {}
    for k, c in sorted(rows.items()):
        if "FETCH_SIZE" not in c or "SQ_INSTS_VALU" not in c: continue
        out[wl][k] = {"hbm_bytes": int(round(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)), "valu_insts": c["SQ_INSTS_VALU"], "px": PX[wl],
                      "source": os.path.relpath(path, ROOT), "lib_sha256": sha}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({w: {k: (v["hbm_bytes"], round(v["valu_insts"] * 64 / v["px"], 1)) for k, v in d.items()} for w, d in out.items() if w != "_format"}, indent=1))
