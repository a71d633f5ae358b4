#!/usr/bin/env python3
"""Register / scratch / occupancy table of every gfx950 kernel in the built objects (calipsync_amd/lib/obj/*.o), read
from the code objects' AMDGPU metadata notes -- no GPU needed.

    python tools/kernel_resources.py            # table on stdout
    python tools/kernel_resources.py --json     # {kernel: {"vgprs": total, "agprs": n, "scratch": bytes, "waves": per SIMD}}

A gfx950 SIMD has 512 registers per lane for its waves (arch + accumulation registers, allocated in blocks of 8), so
waves per SIMD = 512 // roundup(vgpr_count, 8), at most 8.  tests/test_kernel_resources.py pins the occupancy class of
the kernels the bench line rests on: an epilogue feature added to a shared template once took the dominant bf16 GEMM
from two waves per SIMD to one (-13 % at B=512) without failing a single numeric test."""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ_DIR = os.path.join(ROOT, "calipsync_amd", "lib", "obj")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_available() -> bool:
    return all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))


def demangle(names):
    try:
        # binutils' c++filt does not know DF16b (std::bfloat16_t / __bf16): hand it the vendor-type spelling instead
        out = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16b", "u6__bf16") for n in names), capture_output=True,
                             text=True, check=True).stdout.split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)
    short = []
    for d in out[:len(names)]:
        d = d.replace("(anonymous namespace)::", "")
        d = re.sub(r"^void ", "", d)
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):            # cut the argument list: the first '(' outside the template brackets
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        short.append(d[:cut])
    return short


def object_kernels(obj_path: str) -> dict:
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj_path],
                           capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):      # an object without device code (ir_stream.o, product build)
            return {}
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={TARGET}", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
    kernels, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if line.lstrip().startswith("- .") and cur.get("name"):
            kernels[cur["name"]] = cur
            cur = {}
        if key in ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            cur[key] = val if key == "name" else int(val)
    if cur.get("name"):
        kernels[cur["name"]] = cur
    return {k: v for k, v in kernels.items() if "vgpr_count" in v}


def table(obj_dir: str = OBJ_DIR) -> dict:
    out = {}
    for fn in sorted(os.listdir(obj_dir)):
        if not fn.endswith(".o"):
            continue
        ks = object_kernels(os.path.join(obj_dir, fn))
        for mangled, short in zip(ks, demangle(list(ks))):
            k = ks[mangled]
            total = k["vgpr_count"]
            out[short] = {"object": fn, "vgprs": total, "agprs": k.get("agpr_count", 0),
                          "scratch": k.get("private_segment_fixed_size", 0),
                          "static_lds": k.get("group_segment_fixed_size", 0),
                          "waves": min(8, 512 // max(8, (total + 7) // 8 * 8))}
    return out


def main():
    if not tools_available():
        sys.exit("llvm-objcopy / clang-offload-bundler / llvm-readelf not found under " + LLVM_BIN)
    t = table()
    if "--json" in sys.argv:
        print(json.dumps(t, indent=1, sort_keys=True))
        return
    print(f"{'kernel':86s} {'object':14s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>8s} {'waves/SIMD':>10s}")
    for name, k in sorted(t.items(), key=lambda kv: (kv[1]["object"], kv[0])):
        print(f"{name[:86]:86s} {k['object']:14s} {k['vgprs']:5d} {k['agprs']:5d} {k['scratch']:8d} {k['waves']:10d}")


if __name__ == "__main__":
    main()
