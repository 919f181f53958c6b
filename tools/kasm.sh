#!/bin/bash
# Device assembly of one kernel source with extra flags, and the register / LDS lines of the kernels whose mangled name matches a pattern:
#   bash tools/kasm.sh k_ahd.hip 'k_ahd_selectILb0ELb0ELb0ELi1ELb0' [-DFOO ...]     (assembly left in /tmp/kasm_<source>.s)
src=$1; pat=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/kasm_$(basename $src .hip).s
/opt/rocm/bin/hipcc "$@" -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -S --cuda-device-only -o $out $root/pysp_amd/csrc/$src 2>/dev/null || exit 1
python3 - "$out" "$pat" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.rfind("amdhsa.kernels"):]
for blk in meta.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if re.search(sys.argv[2], name):
        g = lambda k: int(re.search(k + r":\s+(\d+)", blk).group(1))
        print(name, "vgpr", g(r"\.vgpr_count"), "sgpr", g(r"\.sgpr_count"), "lds", g(r"\.group_segment_fixed_size"), "scratch", g(r"\.private_segment_fixed_size"))
PY
