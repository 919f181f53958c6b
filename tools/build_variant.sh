#!/bin/bash
# Builds a variant of libpysp_hip.so with extra compiler flags into tools/scratch/<name>.so (kernel A/B experiments):
#   bash tools/build_variant.sh nosb -DAHD_NO_SB      then on the GPU box:  PYSP_HIP_LIB=tools/scratch/nosb.so python bench.py
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cp "$root"/pysp_amd/csrc/*.hip "$root"/pysp_amd/csrc/*.h "$root"/pysp_amd/csrc/*.inc "$root"/pysp_amd/csrc/*.cpp "$root"/pysp_amd/csrc/Makefile "$tmp"/
mkdir -p "$tmp/../../include" 2>/dev/null || true
sed -i "s#\.\./\.\./include/pysp_hip.h#$root/include/pysp_hip.h#g" "$tmp"/Makefile "$tmp"/api.cpp
extra=(); vars=()
for a in "$@"; do if [[ "$a" =~ ^[A-Z]+= ]]; then vars+=("$a"); else extra+=("$a"); fi; done   # VAR=value goes to make, the rest to the compiler
make -C "$tmp" EXTRA="${extra[*]}" "${vars[@]}" -j4 >/dev/null
mkdir -p "$root/tools/scratch"
cp "$tmp/libpysp_hip.so" "$root/tools/scratch/$name.so"
rm -rf "$tmp"
echo "built tools/scratch/$name.so"
