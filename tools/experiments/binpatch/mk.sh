#!/bin/bash
# mk.sh <variant>: $WORK/dev_<variant>.s (a patched copy of the -save-temps device assembly of ir_fused.hip) ->
#   tools/experiments/ablib/libcasync_<variant>.so, linked with the tree's other objects.  README.md beside this file.
#   BUILD = the directory of the `hipcc ... -save-temps` build (holds ir_fused-host-x86_64-unknown-linux-gnu.s)
#   WORK  = where the dev_*.s files are and the intermediates go (default /tmp/binpatch)
set -e
v=$1
: "${BUILD:?the directory of the -save-temps build}"
export WORK=${WORK:-/tmp/binpatch} BUILD
REPO=$(cd "$(dirname "$0")/../../.." && pwd)
LL=/opt/rocm/lib/llvm/bin
cd "$WORK"
$LL/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name ir_fused.hip -target-cpu gfx950 -mrelocation-model pic -o dev_$v.o dev_$v.s
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -plugin-opt=O3 \
  --whole-archive -o dev_$v.out dev_$v.o --no-whole-archive
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null \
  -input=dev_$v.out -output=dev_$v.hipfb
python3 - "$v" <<'PY'
# the host assembly carries the fat binary as one .asciz string: swap it for an .incbin of the new one
import os, sys
v, work, build = sys.argv[1], os.environ["WORK"], os.environ["BUILD"]
src = open(f"{build}/ir_fused-host-x86_64-unknown-linux-gnu.s").read().split("\n")
out, i, size = [], 0, os.path.getsize(f"{work}/dev_{v}.hipfb")
while i < len(src):
    line = src[i]
    if line.startswith('\t.asciz\t"__CLANG_OFFLOAD_BUNDLE__'):
        out.append(f'\t.incbin\t"{work}/dev_{v}.hipfb"')
        i += 1
        name = src[i].split()[1].rstrip(",")
        out.append(f"\t.size\t{name}, {size}")
        i += 1
        continue
    out.append(line)
    i += 1
open(f"{work}/host_{v}.s", "w").write("\n".join(out))
PY
$LL/clang -cc1as -triple x86_64-unknown-linux-gnu -filetype obj -main-file-name ir_fused.hip -target-cpu x86-64 -mrelocation-model pic -o ir_fused_$v.o host_$v.s
mkdir -p "$REPO/tools/experiments/ablib"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$REPO/tools/experiments/ablib/libcasync_$v.so" "$WORK/ir_fused_$v.o" \
  $(ls "$REPO"/calipsync_amd/lib/obj/*.o | grep -v ir_fused.o) 2>&1 | grep -v "argument unused" || true
echo built $v
