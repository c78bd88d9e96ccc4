"""python tools/wide_trace.py type log2n  (under rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs, bench
t, lg = sys.argv[1], int(sys.argv[2])
d = bench.digits_for(rs, t); n = 1 << lg
ctx = rs.Context(0); ctx.set_option(rs.OPT_WIDE_SORT, int(os.environ.get("MODE", "3")))
x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
for it in range(4):
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
    ctx.sort_device(x.data_ptr(), tmp.data_ptr(), n, d); torch.cuda.synchronize(); ctx.check()
