#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into per-kernel HBM traffic per launch.
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
gfx950 correction: FETCH_SIZE counts each 128-byte request as 64 B -> x2; both counters are in KB."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.match(r"(?:void )?(k_[a-z0-9_]+)", name)
    return m.group(1).replace("k_conv_mfma_full", "k_conv_mfma") if m else ""  # both conv variants launch as k_conv_mfma


def per_kernel(path: str, counter: str):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k and row["Counter_Name"] == counter:
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 4, workload C3); values are "
                    "per-launch averages in KB; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts 128-B "
                    "requests as 64 B (MI355X_MICROARCH.md, HBM section), WRITE_SIZE taken as is (atomics are counted as writes)."}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"fetch_size_kb_raw": round(fk, 1), "write_size_kb": round(wk, 1), "traffic_bytes": int((2 * fk + wk) * 1024)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
