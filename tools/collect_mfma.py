#!/usr/bin/env python3
"""Reduce one rocprofv3 SQ/GRBM PMC pass over bench.py to profiles/mfma_util.json (per-kernel MFMA busy fraction).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing
    python tools/collect_mfma.py gpurun_out/pmc_mfma

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 256 CUs * 4 SIMDs).  rocprofv3 reports SQ_BUSY_CYCLES
summed over the 32 shader engines (4 per XCD): SQ_BUSY_CYCLES / 32 / duration comes out at 1.7-2.1 GHz for every long
kernel, the (power-managed) shader clock, so it is the kernel's length in shader cycles.  GRBM_GUI_ACTIVE (summed over
the 8 XCDs) is kept for reference only: it carries ~25 k cycles of per-dispatch overhead under --pmc and would put the
clock above 3 GHz.  ROCm 7.2 ships no gfx950 derived-counter section (guides/MI355X_MICROARCH.md "rocprofv3 PMC
slots"), hence the hand-made MfmaUtil.  SQ_VALU_MFMA_BUSY_CYCLES comes out at 14.67 per 16x16x32 bf16 MFMA (16
would be the issue rate at peak), so 100 % here is ~9 % above the flop peak at the same clock.  Kernels are serialised
under --pmc, so durations are a little longer than in the graph replay.  The wave-cycle
buckets are quad-cycles and disjoint: wait_any (parked on s_waitcnt / barrier) + wait_inst (issue stall) +
active ~= wave cycles.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_traffic import short

SIMDS = 256 * 4


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in launches[k]:
            acc[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        launches[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in acc.items():
        n = len(launches[k])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": n,
             "duration_us_per_launch": c["_ns"] / n / 1e3,
             "gui_active_cycles_per_launch": gui / n,
             "sq_busy_cycles_per_launch": busy / n,
             "clock_ghz": busy / c["_ns"] if c["_ns"] else None,
             "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n,
             "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (busy * SIMDS) if busy else None}
        if wave:
            e["wave_cycles_frac"] = {"wait_any": c.get("SQ_WAIT_ANY", 0.0) / wave,
                                     "wait_inst": c.get("SQ_WAIT_INST_ANY", 0.0) / wave,
                                     "active": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave}
            if c.get("SQ_ACTIVE_INST_LDS"):
                e["lds_bank_conflict_per_lds_cycle"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"]
        out[k] = e
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "mfma_util.json"), "w") as fo:
        json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE (own pass) over bench.py --steps 5 --warmup 2; "
                             "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 SEs * 1024 SIMDs); clock_ghz = SQ_BUSY_CYCLES / 32 / duration",
                   "kernels": out}, fo, indent=1, sort_keys=True)
    tot = sum(v["duration_us_per_launch"] * v["launches"] for v in out.values())
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["duration_us_per_launch"] * kv[1]["launches"])[:16]:
        mb = v["mfma_busy"]
        print(f"{k[:58]:58s} n={v['launches']:5d} share={v['duration_us_per_launch'] * v['launches'] / tot:5.1%} "
              f"us={v['duration_us_per_launch']:7.1f} clk={v['clock_ghz'] or 0:4.2f} mfma_busy={(mb if mb is not None else 0):6.1%}")


if __name__ == "__main__":
    main()
