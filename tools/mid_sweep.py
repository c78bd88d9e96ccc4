"""Device-resident sort time of the middle sizes (between one tile and 2^24): python tools/mid_sweep.py [max_regions]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
if len(sys.argv) > 1 and int(sys.argv[1]): ctx.set_option(rs.OPT_MAX_REGIONS, int(sys.argv[1]))
if os.environ.get("DYN"): ctx.set_option(rs.OPT_TILE_SCHEDULE, 1)
if os.environ.get("NOSKIP"): ctx.set_option(rs.OPT_BUCKET_SKIP, 0)
import bench
for key in os.environ.get("KEYS", "u32,u64").split(";"):
    d = bench.digits_for(rs, key)
    out = []
    for lg in (14, 16, 18, 20, 22, 24):
        if lg > 20 and d.elem_bytes >= 16: continue
        n = 1 << lg
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for it in range(reps + 3):
            ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
            e0.record(); rs.radix_sort(x, digits=d, tmp=tmp); e1.record(); torch.cuda.synchronize()
            if it >= 3: tot += e0.elapsed_time(e1)
        out.append(f"2^{lg}: {tot / reps * 1e3:6.1f}")
    ctx.check()
    print(f"{os.environ.get('RSX_LIBRARY', 'main')[-12:]:12s} regions={sys.argv[1] if len(sys.argv) > 1 else 0} {key}  " + "  ".join(out) + "  us", flush=True)
