"""The committed evidence under profiles/ is self-consistent: the bench line names a dominant kernel that the rocprofv3
kernel stats, the PMC traffic pass and the MFMA-busy pass of the same round all contain under the same name."""
import csv
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "profiles")
sys.path.insert(0, os.path.join(REPO, "tools"))
from kernel_names import short  # noqa: E402


def test_kernel_names_are_demangled():
    assert short("_ZN12_GLOBAL__N_114pw_gemm_kernelIDF16bLi128ELi128ELi2ELi2EEEvPKT_iS3_PS1_iiiiii12GemmEpilogue") == \
        "pw_gemm_kernel<__bf16, 128, 128, 2, 2>"
    assert short("void (anonymous namespace)::pw_gemm_glds_kernel<float, 64, 64, 2, 2, 2, false>(float const*, int)") == \
        "pw_gemm_glds_kernel<float, 64, 64, 2, 2, 2, false>"
    assert short("_ZN12_GLOBAL__N_120ir_fused_bf16_kernelILi64ELi128ELi32ELi1ELb1EEEvPKDF16b") == \
        "ir_fused_bf16_kernel<64, 128, 32, 1, true>"


@pytest.mark.parametrize("bench,traffic,busy,stats", [
    ("r2_bench.json", "r2_pmc_traffic.json", "r2_mfma_busy.json", "r2_f32_b64_kernel_stats_replay.csv"),
    ("r2_bench_bf16_b512.json", "r2_pmc_traffic_bf16_b512.json", "r2_mfma_busy_bf16_b512.json",
     "r2_bf16_b512_kernel_stats_replay.csv"),
])
def test_round_profiles_agree(bench, traffic, busy, stats):
    line = json.load(open(os.path.join(PROF, bench)))
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_busy"):
        assert roof.get(key) is not None, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    dom = roof["kernel"]
    t = json.load(open(os.path.join(PROF, traffic)))["kernels"]
    b = json.load(open(os.path.join(PROF, busy)))["kernels"]
    assert not [k for k in list(t) + list(b) if k.startswith("_Z")], "mangled kernel names in the PMC summaries"
    assert dom in t and dom in b
    assert roof["traffic"] == t[dom]["hbm_bytes_per_launch"] and roof["mfma_busy"] == b[dom]["mfma_busy_of_kernel_time"]
    names = {short(r["Name"]) for r in csv.DictReader(open(os.path.join(PROF, stats)))}
    assert dom in names
    # the average duration rocprofv3 reports for that kernel agrees with the live HIP-event figure of the bench line
    avg = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(PROF, stats)))}[dom]
    assert abs(avg / 1e6 - roof["avg_launch_ms"]) / roof["avg_launch_ms"] < 0.35
    assert line["cpu_baseline"]["kind"] in ("port", "reference") if "cpu_baseline" in line else True
