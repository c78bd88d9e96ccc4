import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, radix_sort_amd as rs
ctx = rs.default_context(0)
d = rs.PRIMITIVES["u16"]
n = (1 << 23) + 1237
g = torch.Generator(device="cpu"); g.manual_seed(1)
x = torch.randint(0, 65536, (n,), dtype=torch.int32, generator=g).to(torch.int16).cuda()
ref = np.sort(x.cpu().numpy().view(np.uint16))
tmp = torch.empty_like(x)
rs.radix_sort(x, digits=d, tmp=tmp, ctx=ctx); torch.cuda.synchronize(); ctx.check()
got = x.cpu().numpy().view(np.uint16)
print("equal:", np.array_equal(got, ref), flush=True)
if not np.array_equal(got, ref):
    bad = np.nonzero(got != ref)[0]
    print("first bad", bad[:10], got[bad[:10]], ref[bad[:10]], "nbad", len(bad), flush=True)
    parts = (n * 2 - 65536 * 8 - 256 * 8) // 131072
    t = tmp.cpu().numpy().view(np.uint8)
    tot = t[parts * 131072: parts * 131072 + 65536 * 8].view(np.uint64)
    print("parts", parts, "sum tot", tot.sum(), "n", n, "hist ok", np.array_equal(tot, np.bincount(ref, minlength=65536).astype(np.uint64)), flush=True)
