import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSX_DEBUG"] = str(int(os.environ.get("RSX_DEBUG", "0"), 0) | 0x100)
import torch, radix_sort_amd as rs
from radix_sort_amd import _lib
ctx = rs.default_context(0)
L = _lib.load()
L.rsx_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
d = rs.PRIMITIVES["u32"]; n = 1 << 28
x = torch.empty(n * 4, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
out = (ctypes.c_ulonglong * 128)()
for it in range(3):
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
    torch.cuda.synchronize()
    L.rsx_debug_counters(ctx._h, out, 1)
    rs.radix_sort(x, digits=d, tmp=tmp)
    L.rsx_debug_counters(ctx._h, out, 1)
    t, hops, spins, depth, mx = out[0], out[1], out[2], out[3], out[4]
    print(f"static-mode launches {out[5]} of 4, placement-verified {out[6]};", end=" ")
    print(f"tiles {t} hops/tile {hops/t:.2f} stall-spins/tile {spins/t:.2f} depth(tiles)/tile {depth/t:.2f} max hops {mx}")
