"""Wide elements by tile size / number of regions: RSX_LIBRARY=... ES=24 KEYB=8 python tools/wide24.py [max_regions ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
for reg in [int(a) for a in sys.argv[1:]] or [0]:
    ctx = rs.Context(0)
    if reg: ctx.set_option(rs.OPT_MAX_REGIONS, reg)
    ES = int(os.environ.get('ES', '24')); KB = int(os.environ.get('KEYB', '8'))
    d = rs.RadixDigits(ES, 0, KB, 0)
    n = (1 << 31) // ES
    x = torch.empty(n * ES, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for it in range(7):
        ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
        e0.record(); rs.radix_sort(x, digits=d, tmp=tmp, ctx=ctx); e1.record(); torch.cuda.synchronize()
        if it >= 2: tot += e0.elapsed_time(e1)
    ms = tot / 5
    print(f"regions<={reg:2d}  {ms:8.3f} ms  {n/ms/1e6:8.2f} Gkeys/s  whole {KB * 2 * n * ES / ms / 1e6 / 80:5.1f}% of 8 TB/s", flush=True)
    ctx.check(); ctx.close(); del x, tmp
