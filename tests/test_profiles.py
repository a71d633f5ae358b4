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
def test_round2_profiles_agree(bench, traffic, busy, stats):
    """Round 2's set, kept for the record (its rocprofv3 average contained the warm-up's two-lane launches, hence the
    loose 35 % bound)."""
    line = json.load(open(os.path.join(PROF, bench)))
    roof = line["roofline"]
    dom = roof["kernel"]
    t = json.load(open(os.path.join(PROF, traffic)))["kernels"]
    b = json.load(open(os.path.join(PROF, busy)))["kernels"]
    assert dom in t and dom in b
    assert roof["traffic"] == t[dom]["hbm_bytes_per_launch"] and roof["mfma_busy"] == b[dom]["mfma_busy_of_kernel_time"]
    avg = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(PROF, stats)))}[dom]
    assert abs(avg / 1e6 - roof["avg_launch_ms"]) / roof["avg_launch_ms"] < 0.35


R3 = [("r3_bench.json", "r3_pmc_traffic.json", "r3_mfma_busy.json", "r3_f32_b64_kernel_stats_replay"),
      ("r3_bench_bf16_b512.json", "r3_pmc_traffic_bf16_b512.json", "r3_mfma_busy_bf16_b512.json",
       "r3_bf16_b512_kernel_stats_replay")]


@pytest.mark.parametrize("bench,traffic,busy,stats", R3)
def test_round3_profiles_reproduce_the_line(bench, traffic, busy, stats):
    """Round 3: every summary carries the sha256 of the kernel sources it was collected from, the bench
    line quotes counters only under the same hash, the rocprofv3 statistics contain ONLY serialised replays, and
    recomputing the roofline fraction from the CSV's AverageNs gives the line's `frac` within 10 %."""
    for f in (bench, traffic, busy, stats + ".csv", stats + ".meta.json"):
        if not os.path.exists(os.path.join(PROF, f)):
            pytest.skip(f"profiles/{f} not collected yet")
    line = json.load(open(os.path.join(PROF, bench)))
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_busy", "avg_kernel_us_rocprof", "source_hash"):
        assert roof.get(key) is not None, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    h = roof["source_hash"]
    t = json.load(open(os.path.join(PROF, traffic)))
    b = json.load(open(os.path.join(PROF, busy)))
    meta = json.load(open(os.path.join(PROF, stats + ".meta.json")))
    assert t["source_hash"] == b["source_hash"] == meta["source_hash"] == h
    dom = roof["kernel"]
    assert not [k for k in list(t["kernels"]) + list(b["kernels"]) if k.startswith("_Z")], "mangled kernel names"
    assert roof["traffic"] == t["kernels"][dom]["hbm_bytes_per_launch"]
    assert roof["mfma_busy"] == b["kernels"][dom]["mfma_busy_of_kernel_time"]
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(PROF, stats + ".csv")))}[dom]
    assert abs(avg_ns / 1e3 - roof["avg_kernel_us_rocprof"]) < 0.02
    # HIP events vs rocprofv3 on the same (serialised) launches: 10 %, and the fraction recomputed from the CSV
    assert abs(avg_ns / 1e6 - roof["avg_launch_ms"]) / roof["avg_launch_ms"] < 0.10
    per_launch = roof["algorithmic_gflop_per_launch"] * 1e9 if roof["bound"] == "mfma" else roof["algorithmic_bytes_per_launch"]
    frac = per_launch / (avg_ns * 1e-9) / (roof["peak"] * (1e12 if roof["bound"] == "mfma" else 1e9))
    assert abs(frac - roof["frac"]) / roof["frac"] < 0.10, (frac, roof["frac"])
    if "cpu_baseline" in line:
        assert line["cpu_baseline"]["kind"] in ("port", "reference")


R4 = [("r4_bench.json", "r4_pmc_traffic.json", "r4_mfma_busy.json", "r4_f32_b64_kernel_stats_replay"),
      ("r4_bench_bf16_b512.json", "r4_pmc_traffic_bf16_b512.json", "r4_mfma_busy_bf16_b512.json",
       "r4_bf16_b512_kernel_stats_replay")]


@pytest.mark.parametrize("bench,traffic,busy,stats", R4)
def test_round4_profiles_reproduce_the_line(bench, traffic, busy, stats):
    """Round 4: as round 3 (one source hash over the set, replay-only statistics, HIP events within 10 % of rocprofv3), and
    the MFMA-busy figure of the line is normalised by the TRACE pass's duration, so it cannot sit below
    achieved / peak; the PMC pass's own normalisation and its slowdown are printed next to it."""
    for f in (bench, traffic, busy, stats + ".csv", stats + ".meta.json"):
        if not os.path.exists(os.path.join(PROF, f)):
            pytest.skip(f"profiles/{f} not collected yet")
    line = json.load(open(os.path.join(PROF, bench)))
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_busy", "mfma_busy_pmc_pass", "pmc_pass_slowdown",
                "avg_kernel_us_rocprof", "avg_launch_ms_raw", "source_hash"):
        assert roof.get(key) is not None, key
    h = roof["source_hash"]
    t = json.load(open(os.path.join(PROF, traffic)))
    b = json.load(open(os.path.join(PROF, busy)))
    meta = json.load(open(os.path.join(PROF, stats + ".meta.json")))
    assert t["source_hash"] == b["source_hash"] == meta["source_hash"] == h
    dom = roof["kernel"]
    assert roof["traffic"] == t["kernels"][dom]["hbm_bytes_per_launch"]
    assert roof["mfma_busy"] == b["kernels"][dom]["mfma_busy_of_trace_time"]
    assert roof["mfma_busy_pmc_pass"] == b["kernels"][dom]["mfma_busy_of_kernel_time"]
    if roof["bound"] == "mfma":
        assert roof["mfma_busy"] >= roof["frac_rocprof"] - 0.03, (roof["mfma_busy"], roof["frac_rocprof"])
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(PROF, stats + ".csv")))}[dom]
    assert abs(avg_ns / 1e6 - roof["avg_launch_ms"]) / roof["avg_launch_ms"] < 0.10
    assert roof["avg_launch_ms_raw"] >= roof["avg_launch_ms"]
    per_launch = roof["algorithmic_gflop_per_launch"] * 1e9 if roof["bound"] == "mfma" else roof["algorithmic_bytes_per_launch"]
    frac = per_launch / (avg_ns * 1e-9) / (roof["peak"] * (1e12 if roof["bound"] == "mfma" else 1e9))
    assert abs(frac - roof["frac"]) / roof["frac"] < 0.10, (frac, roof["frac"])


R5 = [("r5_bench.json", "r5_pmc_traffic.json", "r5_mfma_busy.json", "r5_f32_b64_kernel_stats_replay"),
      ("r5_bench_bf16_b512.json", "r5_pmc_traffic_bf16_b512.json", "r5_mfma_busy_bf16_b512.json",
       "r5_bf16_b512_kernel_stats_replay")]


@pytest.mark.parametrize("bench,traffic,busy,stats", R5)
def test_round5_profiles_reproduce_the_line(bench, traffic, busy, stats):
    """Round 5: the same contract as round 4 -- ONE collection (`tools/collect_profiles.sh r5 f32|bf16`), one source hash
    over the set, replay-only statistics, HIP events within 10 % of rocprofv3, MFMA-busy normalised by the trace pass."""
    test_round4_profiles_reproduce_the_line(bench, traffic, busy, stats)


def test_round5_line_carries_the_single_gpu_anchor_of_configs3():
    """the N = 1 line carries the 512-frame fp32 shard of BASELINE configs[3] with its own whole-net
    fractions and the whole 4096-frame job walked through one GPU in 512-frame chunks (the point an N > 1
    `config.strong_scaling` is divided by), next to the blocks round 3 introduced."""
    path = os.path.join(PROF, "r5_bench.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r5_bench.json not collected yet")
    line = json.load(open(path))
    assert line["dtype"] == "f32" and "batch=64" in line["config"]["workload"] and line["n_gpus"] == 1
    sec = line["secondary"]
    assert set(sec) >= {"b8_fp32", "e2e", "bf16_b512", "fp32_b512", "strong_n1"}
    f512, strong = sec["fp32_b512"], sec["strong_n1"]
    assert f512["batch"] == 512 and f512["dtype"] == "f32" and 0 < f512["whole_net"]["mfma_frac_executed"] <= f512["whole_net"]["mfma_frac"] < 1
    assert strong["global_batch"] == 4096 and strong["frames_per_gpu"] == 4096 and strong["chunk"] == 512 and strong["chunks_per_step"] == 8
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 1
    # eight 512-frame chunks through one arena cost what eight 512-frame forwards cost (no per-chunk penalty): within 5 %
    assert abs(strong["value"] - f512["value"]) / f512["value"] < 0.05
    assert "NOT the parity path" in sec["bf16_b512"]["dtype"] and sec["bf16_b512"]["value"] > sec["fp32_b512"]["value"]


def test_round3_line_carries_the_secondary_block():
    """the driver's plain `bench.py` run also reports configs[2], the frame loop and the reference's own
    B=8 shape -- next to the headline, never instead of it."""
    path = os.path.join(PROF, "r3_bench.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r3_bench.json not collected yet")
    line = json.load(open(path))
    assert line["dtype"] == "f32" and "batch=64" in line["config"]["workload"]
    sec = line["secondary"]
    assert set(sec) >= {"b8_fp32", "e2e", "bf16_b512"}
    assert "NOT the parity path" in sec["bf16_b512"]["dtype"] and sec["bf16_b512"]["roofline"]["frac"] > 0
    assert sec["b8_fp32"]["batch"] == 8 and sec["e2e"]["frames_per_s"] > 0 and sec["e2e"]["frames_per_s_with_masks"] > 0


R6 = [("r6_bench.json", "r6_pmc_traffic.json", "r6_mfma_busy.json", "r6_f32_b64_kernel_stats_replay"),
      ("r6_bench_bf16_b512.json", "r6_pmc_traffic_bf16_b512.json", "r6_mfma_busy_bf16_b512.json",
       "r6_bf16_b512_kernel_stats_replay")]


@pytest.mark.parametrize("bench,traffic,busy,stats", R6)
def test_round6_profiles_reproduce_the_line(bench, traffic, busy, stats):
    """Round 6: the round-4 contract on ONE collection (`tools/collect_profiles.sh r6 f32|bf16`)."""
    test_round4_profiles_reproduce_the_line(bench, traffic, busy, stats)


@pytest.mark.parametrize("bench,traffic", [(R6[0][0], R6[0][1]), (R6[1][0], R6[1][1])])
def test_round6_lines_report_the_bytes_the_engine_moves(bench, traffic):
    """VERDICT r5 #2 / SURVEY 8(d): the roofline at the granularity the kernels implement.  `whole_net.hbm_bytes_moved_per_frame`
    = sum over the step's kernels of the PMC pass's HBM bytes per launch x launches / frames; a fused engine moves FEWER bytes
    than the canonical un-fused network, so `hbm_frac_moved <= hbm_frac_canonical`; and the dominant kernel's share is named
    for what it is (`share_of_replay`)."""
    for f in (bench, traffic):
        if not os.path.exists(os.path.join(PROF, f)):
            pytest.skip(f"profiles/{f} not collected yet")
    line = json.load(open(os.path.join(PROF, bench)))
    wn = line["whole_net"]
    assert wn["hbm_bytes_moved_per_frame"] and 0 < wn["hbm_frac_moved"] <= wn["hbm_frac_canonical"] < 1
    assert wn["hbm_bytes_moved_per_frame"] < wn["canonical_mb_per_frame"] * 1e6
    assert "share_of_replay" in line["roofline"] and "share_of_step" not in line["roofline"]
    # recompute: per_gpu frames/s x bytes / 8 TB/s
    assert abs(line["value"] / line["n_gpus"] * wn["hbm_bytes_moved_per_frame"] / 8e12 - wn["hbm_frac_moved"]) < 2e-3
    if "secondary" in line:
        w16 = line["secondary"]["bf16_b512"]["whole_net"]
        assert 0 < w16["hbm_frac_moved"] <= w16["hbm_frac_canonical"] and w16["mfma_frac_executed"] > 0
        assert "share_of_replay" in line["secondary"]["bf16_b512"]["roofline"]
