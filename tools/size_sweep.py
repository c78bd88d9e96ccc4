"""Device-resident sort time by input size (u32 and u64, uniform keys): python tools/size_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
for key in ("u32", "u64"):
    d = rs.PRIMITIVES[key]
    for n in (1 << 10, 1 << 12, rs.PRIMITIVES[key].elem_bytes == 4 and 14336 or 7168, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26, 1 << 28):
        lg = n.bit_length() - 1
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
        reps = 20 if lg <= 22 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for it in range(reps + 2):
            ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
            e0.record(); rs.radix_sort(x, digits=d, tmp=tmp); e1.record(); torch.cuda.synchronize()
            if it >= 2: tot += e0.elapsed_time(e1)
        ms = tot / reps
        print(f"{key} n={n:<10d} {ms*1e3:9.1f} us  {n/ms/1e6:8.2f} Gkeys/s", flush=True)
    ctx.check()
