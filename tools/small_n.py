"""Sorts n = 2^LG u32 (or KEY) REPS times back to back: python tools/small_n.py LG [KEY] [REPS]  (for rocprofv3 --kernel-trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
lg = int(sys.argv[1]); key = sys.argv[2] if len(sys.argv) > 2 else "u32"; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ctx = rs.default_context(0)
d = rs.PRIMITIVES[key]; n = 1 << lg
xs = [torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda") for _ in range(reps)]
tmp = torch.empty_like(xs[0])
for i, x in enumerate(xs):
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, i)
rs.radix_sort(xs[0], digits=d, tmp=tmp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for x in xs:
    rs.radix_sort(x, digits=d, tmp=tmp)
torch.cuda.synchronize()
print(f"{key} n=2^{lg}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per sort (back to back, host clock)")
ctx.check()
