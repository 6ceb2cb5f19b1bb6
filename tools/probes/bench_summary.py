#!/usr/bin/env python3
"""Prints the per-kernel figures of a bench_details.json (or a DETAILS line on stdin) in a few lines."""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "bench_details.json"
d = json.load(open(src))
print("value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "one-pass", (d.get("latency") or {}).get("us_per_scan_median"),
      "slot_order", (d.get("config") or {}).get("slot_order"))
for ent in [d.get("roofline")] + list(d.get("roofline_others") or []):
    if ent:
        print(f"  {ent['kernel']:45s} alone {ent['avg_us']:7.2f} us  in flight {ent.get('avg_us_in_flight')}")
st = d.get("stages") or {}
for k in ("splat_plus_slice", "splat_plus_slice_in_flight"):
    if isinstance(st.get(k), dict):
        print(" ", k, st[k].get("us_per_scan"), st[k].get("frac_of_hbm_peak"), st[k].get("error"))
