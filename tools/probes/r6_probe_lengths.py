#!/usr/bin/env python3
"""Occupancy statistics of the table under both slot orders: load factor per kd leaf run and the expected number of probes of an
unsuccessful retrieval (distance from a random slot to the next empty one) — what the neighbour traversal pays for absent neighbours."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lattice_net_amd as L
from lattice_net_amd import lattice as LT, synthetic
dev = torch.device("cuda", 0)
n, sigma, cap = 120000, 0.9, int(os.environ.get('CAP', '100000'))
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
cal = torch.from_numpy(synthetic.lidar_cloud(n, 77)).to(dev)
for order in ("hash", "space"):
    LT.set_slot_order(order)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
    lat.begin_splat(); idx, w = lat.just_create_verts(cal, True); lat.nr_lattice_vertices()
    lat.calibrate_regions(idx)
    lat.begin_splat(); idx, w = lat.just_create_verts(pos, True); m = lat.nr_lattice_vertices()
    st = lat.m_hash_table._storage
    ent = st.entries[:st.hashed()].cpu().numpy()
    occ = ent >= 0
    # distance to the next empty slot (wrapping over the whole table: an upper bound on the in-bucket wrap)
    nxt = np.zeros(len(occ), np.int64); d = 0
    for i in range(2 * len(occ) - 1, -1, -1):
        j = i % len(occ)
        d = 0 if not occ[j] else d + 1
        if i < len(occ): nxt[j] = d
    print(f"== {order}: slots {len(occ)} vertices {m} load {occ.mean():.3f}  unsuccessful probes mean {1 + nxt.mean():.2f} p99 {1 + np.percentile(nxt, 99):.0f} max {1 + nxt.max()}")
    if st.slot_map is not None:
        sm = st.slot_map.cpu().numpy()
        for r in range(8):
            a, b = sm[8 + r], sm[9 + r]
            print(f"   leaf {r}: slots {b - a:6d} bucket {sm[17 + r]:4d} load {occ[a:b].mean():.3f} unsuccessful {1 + nxt[a:b].mean():.2f} rows {N[r] if False else ''}")
        print("   row_regions", st.row_regions.cpu().numpy()[:9])
