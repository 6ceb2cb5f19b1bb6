#!/usr/bin/env python3
"""Per-launch averages of the counters tools/pmc_conv.sh collected, per convolution kernel symbol; derived shares:
matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE-or-duration cycles), issue counts per wave."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

out_dir, dst = sys.argv[1], sys.argv[2]
table = defaultdict(dict)
for name in ("busy", "insts", "mops"):
    path = os.path.join(out_dir, name + ".csv")
    if not os.path.exists(path):
        continue
    acc, cnt, dur, seen = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), defaultdict(list), set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "conv" not in k and "grad_filter" not in k:
                continue
            k = re.sub(r"^void ", "", k).split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
            key = (row.get("Dispatch_Id"), k)
            if key not in seen and row.get("Start_Timestamp"):
                seen.add(key)
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k in acc:
        for c in acc[k]:
            table[k][c] = acc[k][c] / cnt[k][c]
        table[k]["us_under_" + name] = round(sum(dur[k]) / max(len(dur[k]), 1), 2)
        table[k]["launches"] = len(dur[k])
for k, t in table.items():
    us = t.get("us_under_busy")
    if us and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        cyc = t.get("GRBM_GUI_ACTIVE") or us * 1e-6 * 2.4e9
        t["mfma_busy_share_of_1024_simds"] = round(t["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        t["clock_assumed"] = "GRBM_GUI_ACTIVE" if t.get("GRBM_GUI_ACTIVE") else "2.4 GHz x duration"
    if "SQ_WAVE_CYCLES" in t:  # quad-cycles summed over waves
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if c in t:
                t[c + "_share_of_wave_cycles"] = round(t[c] / t["SQ_WAVE_CYCLES"], 4)
with open(dst, "w") as f:
    json.dump(table, f, indent=1, sort_keys=True)
for k, t in sorted(table.items()):
    print(k)
    for c in sorted(t):
        print(f"    {c:45s} {t[c]}")
