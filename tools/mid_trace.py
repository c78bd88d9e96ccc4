"""Kernel timeline of a few middle-size sorts (run under rocprofv3 --kernel-trace): python tools/mid_trace.py key log2n [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
key, lg = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
d = rs.PRIMITIVES[key]; n = 1 << lg
x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
for it in range(reps):
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
    torch.cuda.synchronize()
    rs.radix_sort(x, digits=d, tmp=tmp)
    torch.cuda.synchronize()
ctx.check()
