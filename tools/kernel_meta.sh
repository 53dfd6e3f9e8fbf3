#!/bin/bash
# Registers, spills, LDS and scratch of every kernel in a built libhh_world.so (the gfx950 code object's metadata notes).
#   bash tools/kernel_meta.sh [lib.so] [name filter]         (runs in the build container: no GPU needed)
# With DIS=1 also writes the disassembly to $OUT/k.s
set -e
LIB=${1:-/root/repo/hhmarl_2d_amd/lib/libhh_world.so}
FILTER=${2:-.}
OUT=${OUT:-/tmp/hh_kernel_meta}
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"
"$LLVM/llvm-objcopy" --dump-section .hip_fatbin="$OUT/fat.bin" "$LIB"
"$LLVM/clang-offload-bundler" --unbundle --type=o --input="$OUT/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$OUT/k.co"
"$LLVM/llvm-readelf" --notes "$OUT/k.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
flt = re.compile(sys.argv[1])
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if not flt.search(name):
        continue
    agpr = blk.split()[0]
    cols = [("vgpr", "vgpr_count"), ("sgpr", "sgpr_count"), ("vspill", "vgpr_spill_count"), ("sspill", "sgpr_spill_count"),
            ("lds", "group_segment_fixed_size"), ("scratch", "private_segment_fixed_size")]
    print("%-110s agpr %4s " % (name[:110], agpr) + " ".join("%s %5s" % (a, g(b)) for a, b in cols))
' "$FILTER"
if [ -n "$DIS" ]; then "$LLVM/llvm-objdump" -d "$OUT/k.co" > "$OUT/k.s"; echo "disassembly: $OUT/k.s"; fi
