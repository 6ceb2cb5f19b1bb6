#!/usr/bin/env python3
"""Per-shape timing of the streaming linear (+LeakyReLU) kernels of csrc/ln_mlp.hip: forward, backward (x and w parts via
the library's own per-kernel event hooks).  Usage: python tools/bench_linear.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L
from lattice_net_amd.lattice_modules import LinearLeakyReluFunction, LinearMfmaFunction, _MFMA_LINEAR_WIDTHS
lib = L.load_library()
dev = torch.device("cuda", 0)
shapes = [(480000, 4, 16, 0.2), (480000, 16, 32, 0.2), (480000, 9, 1, -1.0), (46538, 96, 96, -1.0), (46538, 96, 48, -1.0), (46538, 48, 8, -1.0),
          (11407, 128, 32, -1.0), (11407, 32, 128, -1.0)]
for rows, cin, cout, slope in shapes:
    x = torch.randn((rows, cin), device=dev, requires_grad=True)
    w = torch.randn((cout, cin), device=dev, requires_grad=True)
    b = torch.randn((cout,), device=dev, requires_grad=True)
    g = torch.randn((rows, cout), device=dev)
    def step():
        x.grad = w.grad = b.grad = None
        LinearLeakyReluFunction.apply(x, w, b, slope).backward(g)
    for _ in range(3):
        step()
    out = []
    for k in ("k_linear_act_forward", "k_linear_act_backward_x", "k_linear_act_backward_w", "k_linear_reduce_slabs"):
        lib.ln_profile_begin(k.encode(), 64)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0.0), C.c_int(0)
        lib.ln_profile_end(C.byref(ms), C.byref(cnt))
        out.append(f"{k[13:]} {ms.value / max(cnt.value, 1) * 1e3:6.1f}us x{cnt.value // 5}")
    print(f"rows={rows:7d} {cin:3d}->{cout:3d}: " + "  ".join(out), flush=True)
    if cin in _MFMA_LINEAR_WIDTHS and cout in _MFMA_LINEAR_WIDTHS:  # the same layer (without bias) as an extent-1 convolution
        def step2():
            x.grad = w.grad = None
            LinearMfmaFunction.apply(x, w).backward(g)
        for _ in range(3):
            step2()
        out = []
        for k in ("k_conv_mfma", "k_grad_filter_mfma", "k_reduce_slabs"):
            lib.ln_profile_begin(k.encode(), 64)
            for _ in range(5):
                step2()
            torch.cuda.synchronize()
            ms, cnt = C.c_double(0.0), C.c_int(0)
            lib.ln_profile_end(C.byref(ms), C.byref(cnt))
            out.append(f"{k} {ms.value / max(cnt.value, 1) * 1e3:6.1f}us x{cnt.value // 5}")
        print(" " * 24 + "as 1x1 convolution: " + "  ".join(out), flush=True)
