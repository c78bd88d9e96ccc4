import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
from radix_sort_amd import _lib
ctx = rs.default_context(0)
L = _lib.load()
L.rsx_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
key = sys.argv[1] if len(sys.argv) > 1 else "u32"
d = rs.PRIMITIVES[key]; n = 1 << int(os.environ.get('LG', '28'))
x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
out = (ctypes.c_ulonglong * 128)()
names = ["ticket+barrier", "load+match", "rank+barrier", "count/scan/offsets", "lds-scatter+barrier", "lookback+barrier", "writeout+barrier"]
for it in range(2):
    ctx.generate_device(x.data_ptr(), n, d, getattr(rs, os.environ.get('GEN', 'GEN_UNIFORM')), it, float(os.environ.get('GENP', '0')))
    torch.cuda.synchronize()
    L.rsx_debug_counters(ctx._h, out, 1)
    rs.radix_sort(x, digits=d, tmp=tmp)
    L.rsx_debug_counters(ctx._h, out, 1)
    kpt = int(os.environ.get('KPT', {4: 28, 8: 12, 16: 5}.get(d.elem_bytes, 28)))
    ntile = d.key_bytes * n / (512 * kpt)
    for w in range(8):
        o = out[w*8:(w+1)*8]
        tot = sum(o[k] for k in range(7))
        tot += o[7]
        print(f"wave {w}: load-wait {o[7]/ntile:6.0f} | " + " | ".join(f"{names[k][:14]} {o[k]/ntile:6.0f}" for k in range(7)), f"| total {tot/ntile:.0f}")
    wgs = max(1, out[67])
    print(f"per workgroup: roll call {out[64]/wgs:6.0f} | cursors {out[65]/wgs:6.0f} | epilogue (flush) {out[66]/wgs:6.0f} | workgroups/launch {wgs/d.key_bytes:.0f} | tiles {ntile/d.key_bytes:.0f}")
    lt = max(1, out[71])
    print(f"look-back of digit 0 per tile: {out[68]/lt:.2f} words looked at, {out[69]/lt:.2f} of them empty, {out[70]/lt:.0f} cycles in the walk")
