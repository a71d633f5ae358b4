"""Static resource budget of the built gfx950 kernels (no GPU): nothing spills, and the kernels the bench lines rest on
keep the waves-per-SIMD class they were tuned at.  Numeric tests cannot see this: a feature added to the shared GEMM
epilogue once raised the register count of EVERY instance (bf16 128x128: 2 -> 1 waves/SIMD, -13 % at B=512; the
dominant fp32 64x64 ring kernel: 48 B of scratch) with every parity test green."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

from calipsync_amd import build  # noqa: E402

pytestmark = pytest.mark.skipif(not kernel_resources.tools_available(), reason="llvm binutils of the ROCm image not found")

# kernel -> lowest waves per SIMD it may be built for (= co-resident workgroups of 4 waves per CU, registers permitting)
MIN_WAVES = {
    "pw_gemm_glds_kernel<float, 64, 64, 2, 2, 2, false>": 5,     # dominant fp32 kernel: five 32-KB rings share a CU
    "pw_gemm_glds_kernel<float, 64, 64, 2, 2, 2, true>": 4,      # implicit-GEMM 3x3 conv
    "pw_gemm_glds_kernel<float, 128, 64, 2, 2, 2, false>": 3,
    "pw_gemm_glds_kernel<float, 64, 32, 2, 1, 4, false>": 3,     # small-M tile, K split in the workgroup (48 KB of LDS: three per CU)
    "pw_gemm_ups_kernel<float, 64, 64, 2, 2, 2>": 4,
    "pw_gemm_kernel<float, 64, 64, 2, 2>": 4,
    "pw_gemm_kernel<float, 128, 32, 4, 1>": 4,
    "pw_gemm_kernel<__bf16, 128, 128, 2, 2>": 2,                 # dominant bf16 kernel (B=512)
    "pw_gemm_kernel<__bf16, 128, 64, 2, 2>": 3,
    "pw_gemm_glds_kernel<__bf16, 64, 64, 2, 2, 2, false>": 4,
    "ir_fused_kernel<float, 32, 64, 32, 1, 16, 0>": 4,           # up3.1 / up4.1
    "ir_fused_kernel<float, 64, 128, 32, 1, 16, 0>": 3,
    "ir_fused_kernel<float, 32, 128, 32, 1, 16, 2>": 3,          # up4.0, upsample commuted
    "ir_fused_kernel<float, 64, 256, 32, 1, 16, 2>": 2,          # up3.0, upsample commuted
    "ir_fused_kernel<float, 64, 128, 64, 1, 16, 0>": 3,          # down1.1
    "ir_fused_kernel<float, 32, 64, 64, 2, 16, 0>": 3,           # down1.0
    "ir_fused_bf16_kernel<64, 128, 32, 1, true, true>": 3,       # (round 6: depthwise on the matrix pipe)
    "ir_fused_bf16_kernel<32, 64, 32, 1, false, true>": 4,
    "ir_fused_bf16_kernel<64, 128, 32, 1, true, false>": 3,      # (round 4: D in registers, stacked pixels share tap rows)
    "ir_fused_bf16_kernel<32, 64, 32, 1, false, false>": 4,
    "pw_dw_kernel<10, 2, 32, 16, 1, 2>": 5,                         # (LDS allows four workgroups per CU; round 4: 9 tap weights + 3 column offsets)
    "pw_dw_kernel<20, 1, 32, 16, 1, 2>": 5,
    "pw_dw_strip_kernel<40, 8, 1, 32, 16>": 4,                   # (57 KB of LDS: two workgroups per CU either way)
    "pw_dw_strip_kernel<40, 4, 2, 32, 16>": 4,
    "pw_dw_bf16_kernel<10, 2, 128, 1>": 2,                       # bf16 expand + depthwise (round 5): 2 workgroups per CU by LDS and registers
    "pw_dw_bf16_kernel<20, 1, 64, 1>": 2,
    "pw_dw_bf16_strip_kernel<40, 8, 1, 64>": 2,
    "pw_dw_bf16_strip_kernel<40, 4, 2, 64>": 2,
    "inc_kernel<float>": 4,
    "outc_kernel<float>": 4,
}


@pytest.fixture(scope="module")
def table():
    build.build()                      # no-op when the library is up to date
    if not os.path.isdir(kernel_resources.OBJ_DIR) or not any(f.endswith(".o") for f in os.listdir(kernel_resources.OBJ_DIR)):
        build.build(force=True)        # a library shipped without its objects: compile them
    return kernel_resources.table()


def test_every_kernel_is_listed(table):
    assert len(table) >= 60
    missing = [k for k in MIN_WAVES if k not in table]
    assert not missing, f"kernels renamed or dropped (update MIN_WAVES): {missing}"


def test_no_kernel_spills(table):
    spilling = {k: v["scratch"] for k, v in table.items() if v["scratch"]}
    assert not spilling, f"kernels with scratch (register spills), bytes per lane: {spilling}"


def test_tuned_kernels_keep_their_occupancy(table):
    low = {k: (table[k]["vgprs"], table[k]["waves"], want) for k, want in MIN_WAVES.items() if table[k]["waves"] < want}
    assert not low, f"(registers, waves/SIMD now, waves/SIMD pinned): {low}"



def test_no_packed_fp32_instruction_takes_its_low_result_from_the_high_half_of_src1(table):
    """The gfx950 erratum of round 6 (tools/isa_pk_opsel.py, tools/experiments/ubench/pk_hazard.hip forms 19-24): v_pk_fma_f32 /
    v_pk_mul_f32 / v_pk_add_f32 with op_sel set for the second source read that half as zero on lanes 48..63 beside another wave's
    v_mfma_f32_16x16x32_bf16.  hipcc emits the form on its own (a scalar factor that lives in the high half of a register pair); when
    it does, take the factor out of the pair at the source level (an empty asm on the scalar, as ir_common.h fma4_scalar and common.h
    ups_lerp do) and look at the disassembly again."""
    import isa_pk_opsel
    hits = isa_pk_opsel.scan_objects()
    assert not hits, hits


def test_the_build_refuses_the_erratum_form(tmp_path):
    """calipsync_amd.build.erratum_instructions() is what build() runs over every object before it links: it must SEE the form
    (a two-line kernel with the instruction written out) and find none in the library's own objects."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "t.hip"
    src.write_text('#include <hip/hip_runtime.h>\ntypedef float f2 __attribute__((ext_vector_type(2)));\n'
                   '__global__ void bad(f2* p) { f2 a = p[threadIdx.x], b = p[threadIdx.x + 64], d;\n'
                   '  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); p[threadIdx.x] = d; }\n'
                   '__global__ void fine(f2* p) { f2 a = p[threadIdx.x], b = p[threadIdx.x + 64], d;\n'
                   '  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); p[threadIdx.x] = d; }\n')
    obj = tmp_path / "t.o"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], check=True, capture_output=True)
    hits = build.erratum_instructions(str(obj))
    assert len(hits) == 1 and hits[0][0].startswith("bad") and "op_sel:[0,1]" in hits[0][1], hits
    own = [h for f in os.listdir(kernel_resources.OBJ_DIR) if f.endswith(".o") for h in build.erratum_instructions(os.path.join(kernel_resources.OBJ_DIR, f))]
    assert not own, own
