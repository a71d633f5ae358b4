#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/collect_pmc_busy.sh -> per-kernel MFMA-busy / stall / LDS figures.

    python tools/pmc_busy_to_json.py <pmc_busy_dir> <out.json> [<kernel_stats_replay.csv>]

Per kernel (average over its launches in `bench.py --replay-only`, i.e. the timed run's own launches,
serialised):
  kernel_cycles            GRBM_GUI_ACTIVE / 8          (the counter sums the 8 XCDs)
  cu_busy                  SQ_BUSY_CU_CYCLES / (256 CUs x kernel_cycles): share of the launch a CU has a wave
  mfma_busy_of_kernel_time SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel_cycles)
  mfma_busy_of_cu_busy     SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)
  wait_any / wait_inst_any / wait_inst_lds   shares of SQ_WAVE_CYCLES (parked on s_waitcnt or a barrier /
                           issue-stalled / issue-stalled on LDS)
  lds_bank_conflict        SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  issue shares (round 4)   SQ_INSTS_VALU (vector instructions incl. MFMA), SQ_INSTS_MFMA / SQ_INSTS_VALU_MFMA_* where the
                           counter exists, SQ_INSTS_LDS, SQ_INSTS_SALU per launch, and valu_per_mfma = (VALU - MFMA) / MFMA
With the rocprofv3 --kernel-trace --stats summary of the same replay (third argument): a PMC pass runs
the kernels slower than the trace pass does, so a busy fraction normalised by the PMC pass's own cycles can sit BELOW
achieved / peak.  Per kernel:
  trace_us                 AverageNs of the trace pass
  pmc_pass_slowdown        kernel_cycles / (trace_us x 2.4 GHz)          (also contains the chip's clock give-back)
  mfma_busy_of_trace_time  SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x trace_us x 2.4 GHz): the matrix pipes' busy cycles over
                           the time the kernel takes when it is not being counted, at the clock the peak is quoted at --
                           by construction >= achieved / peak of the same launch
Units per MI355X_MICROARCH.md (rocprofv3 PMC slots; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_names import short  # noqa: E402
from calipsync_amd.build import source_hash  # noqa: E402  (sha256 of the kernel sources the counters were collected from)


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if name.startswith(("at::", "__amd")) or "elementwise" in name:
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_n": max(len(v) for v in cs.values())} for k, cs in agg.items()}


CLOCK_GHZ = 2.4     # the clock the peaks of bench.py are quoted at (MI355X_MICROARCH.md)


def trace_avg_us(path):
    if not path or not os.path.exists(path):
        return {}
    return {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(path))}


def main():
    src, out_path = sys.argv[1], sys.argv[2]
    trace = trace_avg_us(sys.argv[3] if len(sys.argv) > 3 else None)
    out = {}
    for k, c in sorted(load(src).items()):
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        row = {"launches_sampled": int(c["_n"]), "kernel_cycles": round(cyc)}
        if k in trace and cyc:
            t_cyc = trace[k] * 1e3 * CLOCK_GHZ
            row["trace_us"] = round(trace[k], 2)
            row["pmc_pass_slowdown"] = round(cyc / t_cyc, 3)
            row["mfma_busy_of_trace_time"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * t_cyc), 4)
        insts = {n: c[n] for n in c if n.startswith("SQ_INSTS_")}
        if insts:
            row["insts_per_launch"] = {n[9:].lower(): round(v) for n, v in sorted(insts.items())}
            mfma = insts.get("SQ_INSTS_MFMA") or insts.get("SQ_INSTS_VALU_MFMA_MOPS_F32") or 0.0
            if mfma and insts.get("SQ_INSTS_VALU"):
                row["valu_per_mfma"] = round((insts["SQ_INSTS_VALU"] - mfma) / mfma, 3) if insts["SQ_INSTS_VALU"] > mfma \
                    else round(insts["SQ_INSTS_VALU"] / mfma, 3)
        if cyc:
            row["cu_busy"] = round(c.get("SQ_BUSY_CU_CYCLES", 0.0) / (256 * cyc), 4)
            row["mfma_busy_of_kernel_time"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc), 4)
        if c.get("SQ_BUSY_CU_CYCLES"):
            row["mfma_busy_of_cu_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * c["SQ_BUSY_CU_CYCLES"]), 4)
        if c.get("SQ_WAVE_CYCLES"):
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                row[name[3:].lower()] = round(c.get(name, 0.0) / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = row
    json.dump({"source_hash": source_hash(), "source": f"profiles/{os.path.basename(out_path)}: rocprofv3 --pmc (SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
                         "GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS | SQ_LDS_BANK_CONFLICT "
                         "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS, one group per pass) of `bench.py --replay-only` "
                         "(tools/collect_pmc_busy.sh)",
               "kernels": out}, open(out_path, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:58s} n={v['launches_sampled']:4d} cyc={v['kernel_cycles']:8d} cu_busy={v.get('cu_busy', 0):.2f} "
              f"mfma/kernel={v.get('mfma_busy_of_kernel_time', 0):.2f} mfma/cu_busy={v.get('mfma_busy_of_cu_busy', 0):.2f} "
              f"wait={v.get('wait_any', 0):.2f} bank={v.get('lds_bank_conflict', 0):.2f}")


if __name__ == "__main__":
    main()
