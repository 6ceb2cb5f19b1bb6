#!/usr/bin/env python3
"""Copies the summaries tools/gpu_profile_r6.sh left in gpurun_out/prof_r6/ into profiles/ under their round-6 names."""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_r6"), os.path.join(ROOT, "profiles")
MAP = {
    "bench_C3_line.json": "r6_bench_C3.json", "bench_C3_details.json": "r6_bench_C3_default_details.json",
    "bench_C2_line.json": "r6_bench_C2.json", "bench_C2_details.json": "r6_bench_C2_details.json",
    "bench_C4_line.json": "r6_bench_C4.json", "bench_C5_line.json": "r6_bench_C5.json",
    "bench_driver_line.json": "r6_bench_C3_driver_style_K20.json", "bench_driver_details.json": "r6_bench_C3_driver_style_K20_details.json",
    "c3_hash_in_flight_kernel_stats.csv": "r6_kernel_stats.csv", "c3_hash_in_flight_bench_line.json": "r6_bench_line_of_kernel_stats.json",
    "c3_hash_one_in_flight_kernel_stats.csv": "r6_kernel_stats_one_in_flight.csv",
    "c3_hash_one_in_flight_bench_line.json": "r6_bench_line_of_kernel_stats_one_in_flight.json",
    "c3_space_in_flight_kernel_stats.csv": "r6_kernel_stats_space_order.csv",
    "c3_space_in_flight_bench_line.json": "r6_bench_line_of_kernel_stats_space_order.json",
    "c3_space_one_in_flight_kernel_stats.csv": "r6_kernel_stats_space_order_one_in_flight.csv",
    "c2_one_in_flight_kernel_stats.csv": "r6_kernel_stats_C2_one_in_flight.csv",
    "c4_one_in_flight_kernel_stats.csv": "r6_kernel_stats_C4_one_in_flight.csv",
    "c5_one_in_flight_kernel_stats.csv": "r6_kernel_stats_C5_one_in_flight.csv",
    "lnn_unet_kernel_stats.csv": "r6_lnn_unet_kernel_stats.csv", "lnn_scannet_kernel_stats.csv": "r6_lnn_scannet_kernel_stats.csv",
    "lnn_shapenet_kernel_stats.csv": "r6_lnn_shapenet_kernel_stats.csv", "lnn_graph_steps.txt": "r6_lnn_graph_steps.txt",
    "pmc_fetch_hash_counter_collection.csv": "r6_pmc_fetch_size_counter_collection.csv",
    "pmc_write_hash_counter_collection.csv": "r6_pmc_write_size_counter_collection.csv",
    "pmc_fetch_space_counter_collection.csv": "r6_pmc_fetch_size_space_order_counter_collection.csv",
    "pmc_write_space_counter_collection.csv": "r6_pmc_write_size_space_order_counter_collection.csv",
    "pmc_traffic_hash.json": "r6_pmc_traffic.json", "pmc_traffic_space.json": "r6_pmc_traffic_space_order.json",
    "conv_time.txt": "r6_conv_time.txt", "conv_time_level2.txt": "r6_conv_time_level2.txt",
    "pmc_lds_c3.txt": "r6_pmc_lds_c3.txt", "pmc_lds_kitti.txt": "r6_pmc_lds_kitti.txt",
}
missing = []
for s, d in MAP.items():
    p = os.path.join(SRC, s)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        missing.append(s)
        continue
    shutil.copyfile(p, os.path.join(DST, d))
print("copied", len(MAP) - len(missing), "files; missing:", missing)
sys.exit(1 if missing else 0)
