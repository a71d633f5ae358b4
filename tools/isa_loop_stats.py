#!/usr/bin/env python3
"""Instruction mix of a kernel's loops, from the gfx950 code object (no GPU needed).

    python tools/isa_loop_stats.py ir_fused.o 'ir_fused_kernel<float, 32, 128, 32, 1, 16, 2>' [--dump]

Disassembles the kernel, finds every backward branch (a loop), and prints per loop body the count of MFMA /
VALU / LDS / VMEM / SALU / wait instructions and the MFMA issue cycles (pass counts: 16x16x4 f32 = 8 passes = 32
cycles, 32x32x2 f32 = 16 passes = 64 cycles, bf16 16x16x32 = 8 passes... taken from the opcode), so "how many
vector instructions share the issue slots with the MFMAs of one chunk" is read off the binary, not inferred."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import kernel_resources as kr  # noqa: E402

MFMA_CYCLES = {"v_mfma_f32_16x16x4_f32": 32, "v_mfma_f32_32x32x2_f32": 64, "v_mfma_f32_16x16x32_bf16": 32,
               "v_mfma_f32_32x32x16_bf16": 64, "v_mfma_f32_16x16x16_bf16": 16, "v_mfma_f32_32x32x8_bf16": 32}


def disassemble(obj_path: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(kr.LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj_path], check=True)
        subprocess.run([os.path.join(kr.LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={kr.TARGET}", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([os.path.join(kr.LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", co], check=True,
                              capture_output=True, text=True).stdout


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def kernel_body(asm: str, want: str):
    cur, out = None, {}
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None and line.strip():
            out[cur].append(line)
    norm = lambda s: re.sub(r"\s+", "", s.replace("(anonymous namespace)::", "").replace("void ", ""))
    for name, body in out.items():
        if norm(name).startswith(norm(want)):
            return name, body
    raise SystemExit(f"kernel {want!r} not found; have e.g. {[n for n in out if 'kernel' in n][:5]}")


def main():
    obj = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(kr.OBJ_DIR, sys.argv[1])
    name, body = kernel_body(disassemble(obj), sys.argv[2])
    ins = []        # (addr-label or None, op, text)
    labels = {}
    for line in body:
        m = re.match(r"^<(L\d+|[.\w$]+)>:$", line.strip())
        if m:
            labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^\s*(\S+)\s*(.*?)\s*(?://\s*([0-9A-Fa-f]+):.*)?$", line)
        if not m:
            continue
        ins.append((m.group(1), m.group(2), m.group(3)))
    # address of each instruction -> index
    addr = {int(a, 16): i for i, (_, _, a) in enumerate(ins) if a}
    loops = []
    for i, (op, args, a) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = None
            m2 = re.match(r"(\d+)", args)
            if m2 and a:      # relative branch: target = next instruction + 4 * simm16
                off = int(m2.group(1))
                if off >= 0x8000:
                    off -= 0x10000
                tgt = addr.get(int(a, 16) + 4 + 4 * off)
            if tgt is not None and tgt <= i:
                loops.append((tgt, i))
    print(f"{name}: {len(ins)} instructions, {len(loops)} loops")
    for tgt, end in loops:
        cnt, cyc = {}, 0
        for op, args, _ in ins[tgt:end + 1]:
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
            if c == "mfma":
                cyc += MFMA_CYCLES.get(op, 32)
        inner = [l for l in loops if l != (tgt, end) and l[0] >= tgt and l[1] <= end]
        print(f"  loop [{tgt}..{end}] {end - tgt + 1:5d} instr: " + "  ".join(f"{k} {v}" for k, v in sorted(cnt.items())) +
              f"  | MFMA issue cycles {cyc}" + (f"  (contains {len(inner)} inner loops)" if inner else ""))
    if "--dump" in sys.argv and loops:
        tgt, end = max(loops, key=lambda l: l[1] - l[0])
        for op, args, a in ins[tgt:end + 1]:
            print(f"    {op} {args}")


if __name__ == "__main__":
    main()
