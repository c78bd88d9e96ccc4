import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, radix_sort_amd as rs
ctx = rs.default_context(0)
d = rs.PRIMITIVES["u16"]
for lg in (24, 26, 28):
    n = 1 << lg
    x = torch.empty(n * 2, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    for gen in (rs.GEN_UNIFORM, rs.GEN_ZIPF, rs.GEN_CONSTANT):
        ctx.generate_device(x.data_ptr(), n, d, gen, 3, 1.0 if gen == rs.GEN_ZIPF else 7.0)
        ctx.verify_device(x.data_ptr(), n, d, out.data_ptr()); torch.cuda.synchronize(); before = out[1].item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rs.radix_sort(x, digits=d, tmp=tmp, ctx=ctx); e1.record(); torch.cuda.synchronize(); ctx.check()
        ctx.verify_device(x.data_ptr(), n, d, out.data_ptr()); torch.cuda.synchronize()
        print(f"2^{lg} gen {gen}: {e0.elapsed_time(e1)*1e3:9.1f} us  descents {out[0].item()}  multiset {'ok' if out[1].item() == before else 'CHANGED'}", flush=True)
