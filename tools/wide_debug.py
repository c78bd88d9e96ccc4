"""Reproduce one forced-hybrid case against the oracle: python tools/wide_debug.py type n dist seed skip group [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, radix_sort_amd as rs, util
from oracle import oracle as orc
t, n, dist, seed, skip, group = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
mode = int(sys.argv[7]) if len(sys.argv) > 7 else 2
es, ko, kb, kind = util.TYPES[t]
d = rs.RadixDigits(es, ko, kb, kind) if hasattr(rs, "RadixDigits") else None
import bench
d = bench.digits_for(rs, t)
c = rs.Context(0); c.set_option(rs.OPT_WIDE_SORT, mode); c.set_option(rs.OPT_BUCKET_SKIP, skip); c.set_option(rs.OPT_BUCKET_GROUP, group)
raw = util.make_input(t, n, dist, seed=seed)
x = torch.from_numpy(raw.copy()).cuda()
rs.radix_sort(x, digits=d, ctx=c); c.check()
got = x.cpu().numpy().reshape(n, es); exp = orc.sort_parallel(raw, orc.Layout(*util.TYPES[t]), 8).reshape(n, es)
bad = np.flatnonzero((got != exp).any(axis=1))
print("info", hex(c.get_info(rs.INFO_LAST_PASSES)), "mismatching rows", len(bad), "of", n)
for i in bad[:6]:
    print(i, got[i].tobytes().hex(), exp[i].tobytes().hex())
# is the output at least a sorted permutation by key?
k = got[:, ko:ko + kb][:, ::-1].copy()
print("multiset equal", np.array_equal(np.sort(got.view(np.uint8).reshape(n, es), axis=0), np.sort(exp.view(np.uint8).reshape(n, es), axis=0)))
