"""Per-launch times of the max-centring kernels of the DeformSlice head at the SemanticKITTI shape ([120 k, 4, 9])."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd.lattice_blocks import max_centre_rows  # noqa: E402
lib = L.load_library(); dev = torch.device("cuda", 0)
x = torch.randn((120000, 4, 9), device=dev, requires_grad=True)
gamma = torch.rand(9, device=dev).requires_grad_(True); beta = torch.rand(9, device=dev).requires_grad_(True)
g = torch.randn((120000, 4, 9), device=dev)
def step():
    x.grad = gamma.grad = beta.grad = None
    max_centre_rows(x, gamma, beta).backward(g)
for _ in range(200): step()
torch.cuda.synchronize()
for name in (b"k_max_centre_forward", b"k_max_centre_backward", b"k_max_centre_sum"):
    lib.ln_profile_begin(name, 64)
    for _ in range(20): step()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0), C.c_int(0); lib.ln_profile_end(C.byref(ms), C.byref(cnt))
    print(f"{name.decode():24s} {ms.value / max(cnt.value, 1) * 1e3:7.1f} us")
