"""Does the NUMBERING of the lattice rows matter to the gather-bound kernels?  The convolution is called through the C ABI on the
C3 lattice's neighbour list under three numberings of the same rows: slot order (shipped), a spatial sort of the keys, a random
permutation.  Same arithmetic, same bytes; only which rows sit next to each other in memory changes."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import lattice_net_amd as L
from lattice_net_amd import synthetic, _lib
from ops_roofline import _profile
dev = torch.device("cuda", 0)
lib = L.load_library()
pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
lat.begin_splat()
dl, _, _, _ = lat.distribute(pos, torch.zeros((120000, 1), device=dev))
m = dl.nr_lattice_vertices()
nbr = dl.neighbours(dl, 1, False).long()
keys = dl.hash_table().m_keys_tensor[:m].long()                      # [m, 3] lattice coordinates
k = keys - keys.min(0).values
# spatial order: Morton code of the (coarsened by 4) key, then the key itself
def part(x):
    x = x & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x
mort = part(k[:, 0] >> 2) | (part(k[:, 1] >> 2) << 1) | (part(k[:, 2] >> 2) << 2)
spatial = torch.argsort(mort, stable=True)
# spatial AND XCD-aware: workgroup t (64 rows) runs on XCD t % 8 (round-robin dispatch), so hand XCD k the k-th eighth of the
# Morton order: chunk c of the sorted rows becomes tile (c % (T/8)) * 8 + c / (T/8)
T8 = (m // 64) // 8
c = torch.arange(T8 * 8, device=dev)
tile_of_chunk = (c % T8) * 8 + c // T8
xcd = torch.arange(m, device=dev)
src_rows = (c[:, None] * 64 + torch.arange(64, device=dev)[None, :])            # Morton positions of chunk c
dst_rows = (tile_of_chunk[:, None] * 64 + torch.arange(64, device=dev)[None, :])  # where they go
perm_pos = torch.arange(m, device=dev)
perm_pos[dst_rows.reshape(-1)] = src_rows.reshape(-1)
orders = {"slot order": torch.arange(m, device=dev), "spatial (Morton of key / 4)": spatial,
          "spatial, an eighth per XCD": spatial[perm_pos], "random": torch.randperm(m, device=dev)}
torch.manual_seed(0)
for v, f in ((32, 32), (64, 64), (128, 128)):
    vals0 = torch.randn((m, v), device=dev)
    bank = torch.randn((9 * v, f), device=dev) * 0.05
    ref = None
    for name, perm in orders.items():
        inv = torch.empty_like(perm); inv[perm] = torch.arange(m, device=dev)
        nb = nbr[perm]
        nb2 = torch.where(nb >= 0, inv[nb.clamp(min=0)], nb).int().contiguous()
        vals = vals0[perm].contiguous()
        out = torch.empty((m, f), device=dev)
        wsb = int(lib.ln_conv_forward_workspace_bytes(m, 9, v, f))
        ws = torch.empty((max(wsb, 256),), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            _lib.check(lib.ln_conv_forward_ws(_lib.ptr(nb2), _lib.ptr(vals), _lib.ptr(bank), m, 9, v, f, 0, _lib.ptr(out), _lib.ptr(ws), ws.numel(), C.c_void_p(st)), "conv")
        kk = _profile(lib, run, 20)
        y = torch.empty_like(out); y[perm] = out
        if ref is None: ref = y
        print(f"V {v} F {f}  {name:28s} {sum(x['us_per_call'] for x in kk):7.1f} us  " + ", ".join(f"{x['kernel']} {x['avg_us']:.1f}" for x in kk) + f"   max diff vs slot order {float((y - ref).abs().max()):.1e}")
