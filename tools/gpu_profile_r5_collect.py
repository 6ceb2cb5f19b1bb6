#!/usr/bin/env python3
"""Collects what tools/gpu_profile_r5.sh left in gpurun_out/prof_r5 into the tracked profiles/r5_* files.

    python tools/gpu_profile_r5_collect.py gpurun_out/prof_r5 [--summary]     (--summary: only print, copy nothing)

r5_pmc_mfma.json: per kernel symbol (template arguments kept), per-launch averages of the matrix-core counters of the three
workloads (C3 step, per-operator table, C5) — SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs), SQ_BUSY_CYCLES, and the
MOPS counters by type (one MOPS = 512 flop, MI355X_MICROARCH.md) — with the launch's duration under the counters, and derived:
executed TFLOP/s by type and the matrix pipe's busy share = MFMA busy cycles / (1024 SIMDs x duration x 2.4 GHz)."""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

SIMDS, CLOCK_GHZ = 1024, 2.4


def sym(name: str) -> str:
    m = re.match(r"(?:void )?((?:k_|ln_k)[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.match(r"_Z\d+((?:k_|ln_k)[A-Za-z0-9_]+?)I(.*?)E[Ev]", name)  # mangled template instance (the f16 kernels are listed that way)
    return (m.group(1) + "<" + m.group(2) + ">") if m else ""


def counters(path):
    acc, dur, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(float), defaultdict(lambda: defaultdict(int))
    if not os.path.exists(path):
        return {}
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = sym(row["Kernel_Name"])
            if not k:
                continue
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            cnt[k][c] += 1
            key = (row.get("Dispatch_Id"), k)
            if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add(key)
                dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                cnt[k]["_dur"] += 1
    out = {}
    for k in acc:
        out[k] = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        if cnt[k]["_dur"]:
            out[k]["avg_us_under_counters"] = round(dur[k] / cnt[k]["_dur"], 2)
        out[k]["launches"] = max(cnt[k][c] for c in acc[k])
    return out


def mfma_table(out_dir):
    table = {}
    for wl in ("c3", "ops", "c5"):
        busy = counters(os.path.join(out_dir, f"mfma_busy_{wl}_counter_collection.csv"))
        mops = counters(os.path.join(out_dir, f"mfma_mops_{wl}_counter_collection.csv"))
        for k in sorted(set(busy) | set(mops)):
            e = {}
            e.update({c: round(v, 1) for c, v in busy.get(k, {}).items() if c.startswith("SQ_")})
            e.update({c: round(v, 1) for c, v in mops.get(k, {}).items() if c.startswith("SQ_")})
            us = busy.get(k, {}).get("avg_us_under_counters") or mops.get(k, {}).get("avg_us_under_counters")
            if not any(e.get(c, 0) for c in e if "MFMA" in c):
                continue
            e["avg_us_under_counters"] = us
            e["launches"] = busy.get(k, {}).get("launches") or mops.get(k, {}).get("launches")
            if us:
                if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                    e["mfma_busy_share"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * us * 1e-6 * CLOCK_GHZ * 1e9), 4)
                for c in [c for c in e if c.startswith("SQ_INSTS_VALU_MFMA_MOPS_")]:
                    e["executed_tflops_" + c.rsplit("_", 1)[1].lower()] = round(e[c] * 512 / (us * 1e-6) / 1e12, 2)
            table[f"{wl}:{k}"] = e
    return table


def lds_table(out_dir):
    """Per dense kernel: LDS instructions, bank-conflict cycles, LDS issue stalls, LDS-array cycles, VALU / MFMA instruction counts,
    wave cycles and issue stalls (per-launch averages of the lds_* passes), and the chip clock of the clock_ops pass."""
    table = {}
    for wl in ("ops", "c5"):
        t = counters(os.path.join(out_dir, f"lds_{wl}_counter_collection.csv"))
        for k, e in t.items():
            if not e.get("SQ_INSTS_MFMA"):
                continue
            row = {c: round(v, 1) for c, v in e.items() if c.startswith("SQ_")}
            row["avg_us_under_counters"], row["launches"] = e.get("avg_us_under_counters"), e.get("launches")
            if row.get("SQ_LDS_IDX_ACTIVE"):
                row["bank_conflict_share_of_lds_cycles"] = round(row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"], 4)
            if row.get("SQ_WAVE_CYCLES"):
                row["lds_issue_stall_share_of_wave_cycles"] = round(row.get("SQ_WAIT_INST_LDS", 0.0) / row["SQ_WAVE_CYCLES"], 4)
                row["issue_stall_share_of_wave_cycles"] = round(row.get("SQ_WAIT_INST_ANY", 0.0) / row["SQ_WAVE_CYCLES"], 4)
            table[f"{wl}:{k}"] = row
    clk = counters(os.path.join(out_dir, "clock_ops_counter_collection.csv"))
    for k, e in clk.items():
        key = f"ops:{k}"
        if key in table and e.get("GRBM_GUI_ACTIVE") and e.get("avg_us_under_counters"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            table[key]["clock_ghz_under_counters"] = round(e["GRBM_GUI_ACTIVE"] / 8 / (e["avg_us_under_counters"] * 1e3), 3)
    return table


def main():
    out_dir = sys.argv[1]
    summary = "--summary" in sys.argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    table = mfma_table(out_dir)
    if summary:
        for name in ("c3_in_flight", "c3_one_in_flight", "ops", "lnn_unet", "c5"):
            p = os.path.join(out_dir, f"{name}_kernel_stats.csv")
            if os.path.exists(p):
                rows = list(csv.DictReader(open(p)))
                print(f"== {name}")
                for r in rows[:14]:
                    print(f'  {r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Percentage"]}%')
        for k, e in table.items():
            print(k[:60].ljust(60), {c: e[c] for c in e if not c.startswith("SQ_")})
        return
    copies = {"c3_in_flight_kernel_stats.csv": "r5_kernel_stats.csv", "c3_one_in_flight_kernel_stats.csv": "r5_kernel_stats_one_in_flight.csv",
              "ops_kernel_stats.csv": "r5_ops_kernel_stats.csv", "lnn_unet_kernel_stats.csv": "r5_lnn_unet_kernel_stats.csv",
              "c5_kernel_stats.csv": "r5_kernel_stats_C5_one_in_flight.csv", "c3_in_flight_bench_line.json": "r5_bench_line_of_kernel_stats.json",
              "c3_one_in_flight_bench_line.json": "r5_bench_line_of_kernel_stats_one_in_flight.json", "ops_bench_line.json": "r5_ops_table_of_kernel_stats.json",
              "pmc_fetch_counter_collection.csv": "r5_pmc_fetch_size_counter_collection.csv",
              "pmc_write_counter_collection.csv": "r5_pmc_write_size_counter_collection.csv", "pmc_traffic.json": "r5_pmc_traffic.json",
              "c3_in_flight_bench_details.json": "r5_bench_details_of_kernel_stats.json", "ops_bench_details.json": "r5_ops_table_of_kernel_stats_details.json",
              "bench_C3_line.json": "r5_bench_C3_default.json", "bench_C3_details.json": "r5_bench_C3_default_details.json",
              "bench_C2_line.json": "r5_bench_C2.json", "bench_C4_line.json": "r5_bench_C4.json", "bench_C5_line.json": "r5_bench_C5.json",
              "bench_C5_details.json": "r5_bench_C5_details.json",
              "bench_driver_line.json": "r5_bench_C3_driver_style_K20.json", "bench_driver_details.json": "r5_bench_C3_driver_style_K20_details.json",
              "conv_time.txt": "r5_conv_time.txt", "gf_time.txt": "r5_grad_filter_time.txt"}
    for src, dst in copies.items():
        p = os.path.join(out_dir, src)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(prof, dst))
            print("copied", dst)
        else:
            print("MISSING", src)
    with open(os.path.join(prof, "r5_pmc_mfma.json"), "w") as f:
        json.dump({"_note": "tools/gpu_profile_r5.sh: rocprofv3 --pmc passes (busy cycles; MOPS by type), --kernel-trace only, per-launch averages; "
                            "keys = workload:kernel symbol (c3 = bench.py C3 step, ops = tools/ops_roofline.py, c5 = bench.py --workload C5); "
                            "mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz); executed_tflops_* = MOPS x 512 / duration",
                   **table}, f, indent=1)
    print("wrote r5_pmc_mfma.json with", len(table), "entries")
    lds = lds_table(out_dir)
    with open(os.path.join(prof, "r5_pmc_lds.json"), "w") as f:
        json.dump({"_note": "tools/gpu_profile_r5.sh: rocprofv3 --pmc passes over the dense kernels (ops = tools/ops_roofline.py, c5 = bench.py --workload C5), "
                            "--kernel-trace only, per-launch averages.  SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves; "
                            "SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles of conflicts", **lds}, f, indent=1)
    print("wrote r5_pmc_lds.json with", len(lds), "entries")


if __name__ == "__main__":
    main()
