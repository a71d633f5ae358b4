#!/usr/bin/env python3
"""The gfx950 packed-fp32 erratum found in round 6, as a check on the built code objects (no GPU needed).

    python tools/isa_pk_opsel.py [obj ...]

A packed fp32 instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) whose LOW result takes the HIGH half of its second source
-- `op_sel:[x,1]` / `op_sel:[x,1,x]` -- reads that half as ZERO on lanes 48..63 whenever another wave of the SIMD issues
`v_mfma_f32_16x16x32_bf16` at the same time: the FMA returns its addend, the multiply 0, the add its first source.  200 of 200
launches of tools/experiments/ubench/pk_hazard.hip (forms 19, 20, 23, 24) beside a register-only loop of that matrix instruction,
never beside fp32 matrix instructions, vector work or nothing; `op_sel` on the FIRST source and `op_sel_hi` (the high result taking a
low half: the usual scalar broadcast) never fail.  hipcc emits the form when a scalar factor happens to live in the high half of a
register pair -- in round 6 the bilinear weights of the fused Up block, which made an fp32 model return wrong patches beside a bf16
model (profiles/r6_two_models.txt section 9).  This tool lists every such instruction per kernel (op_sel on the THIRD source of an FMA,
form 25 of the microbenchmark, never failed and is not listed); tests/test_kernel_resources.py keeps the shipped library free of them.
"""
from __future__ import annotations

import glob
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import isa_loop_stats as isa  # noqa: E402
import kernel_resources as kr  # noqa: E402

PACKED = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b(.*?)(?://|$)")


def scan(asm: str):
    """-> {kernel: [instruction text]} of packed fp32 instructions with op_sel set for the second source"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            continue
        mm = PACKED.search(line)
        if cur is None or not mm:
            continue
        sel = re.search(r"op_sel:\[([01,]+)\]", mm.group(2))
        if sel and sel.group(1).split(",")[1:2] == ["1"]:
            out.setdefault(cur, []).append((mm.group(1) + mm.group(2)).strip())
    return out


def scan_objects(objs=None):
    res = {}
    for obj in objs or sorted(glob.glob(os.path.join(kr.OBJ_DIR, "*.o"))):
        try:
            asm = isa.disassemble(obj)
        except Exception:   # an object without device code
            continue
        for k, v in scan(asm).items():
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
            res[os.path.basename(obj) + "  " + (name[:name.index("(")] if "(" in name else name)] = v
    return res


if __name__ == "__main__":
    hits = scan_objects(sys.argv[1:] or None)
    for k, v in hits.items():
        print(f"{k}: {len(v)}")
        for ins in v[:8]:
            print("     ", ins)
    print(f"{sum(len(v) for v in hits.values())} packed fp32 instruction(s) whose low result reads the high half of src1")
    sys.exit(1 if hits else 0)
