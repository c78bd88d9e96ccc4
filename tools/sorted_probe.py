import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
d = rs.PRIMITIVES["u32"]; n = 1 << 28
x = torch.arange(n, dtype=torch.int32, device="cuda")
dst = torch.empty_like(x)
for digit in range(4):
    for it in range(3):
        ctx.profile(True)
        ctx.partition_device(x.data_ptr(), dst.data_ptr(), n, d, digit)
        torch.cuda.synchronize()
        p = ctx.profile_read()
    print(f"sorted input, pass on digit {digit}: sweep {p['sweep'][0]/p['sweep'][1]:.4f} ms  hist {p['hist'][0]/p['hist'][1]:.4f} ms")
