"""Device-resident sort time by input size (u32 and u64, uniform keys): python tools/size_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
for key in ("u32", "u64"):
    d = rs.PRIMITIVES[key]
    for lg in (10, 14, 16, 18, 20, 22, 24, 26, 28):
        n = 1 << lg
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
        reps = 20 if lg <= 22 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for it in range(reps + 2):
            ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
            e0.record(); rs.radix_sort(x, digits=d, tmp=tmp); e1.record(); torch.cuda.synchronize()
            if it >= 2: tot += e0.elapsed_time(e1)
        ms = tot / reps
        print(f"{key} n=2^{lg:<2d} {ms*1e3:9.1f} us  {n/ms/1e6:8.2f} Gkeys/s", flush=True)
    ctx.check()
