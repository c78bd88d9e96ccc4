import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, util
import radix_sort_amd as rs
from oracle import oracle
t, n = "u32", 3000001
d = rs.RadixDigits(*util.TYPES[t]); lay = oracle.Layout(*util.TYPES[t])
raw = util.make_input(t, n, "uniform", seed=31)
for name, opts in (("default", []), ("ballots", [(rs.OPT_RANKING, 1)]), ("ballots+dynamic", [(rs.OPT_RANKING, 1), (rs.OPT_TILE_SCHEDULE, 1)]), ("ballots+1region", [(rs.OPT_RANKING, 1), (rs.OPT_MAX_REGIONS, 1)])):
    c = rs.Context(0)
    for o, v in opts: c.set_option(o, v)
    for rep in range(3):
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c); c.check()
        got = x.cpu().numpy().view("<u4"); exp = oracle.sort_parallel(raw, lay, 8).view("<u4")
        bad = np.nonzero(got != exp)[0]
        print(name, "sort rep", rep, "mismatches", bad.size, "first", bad[:3], flush=True)
    src = torch.from_numpy(raw.copy()).cuda(); dst = torch.empty_like(src)
    for digit in range(4):
        e, h = oracle.partition_pass(raw, lay, digit)
        c.partition_device(src.data_ptr(), dst.data_ptr(), n, d, digit); c.check()
        g = dst.cpu().numpy().view("<u4"); bad = np.nonzero(g != e.view("<u4"))[0]
        print(name, "partition digit", digit, "mismatches", bad.size, bad[:3], flush=True)
    c.close()
