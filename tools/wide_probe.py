"""Wide-key hybrid (RSX_OPT_WIDE_SORT) against the LSD passes: python tools/wide_probe.py [type ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
import bench
types = sys.argv[1:] or ["u64"]
MODES = [int(v) for v in os.environ.get("MODES", "0,1").split(",")]
TRI = int(os.environ.get("TRI", "0"))
DUP = int(os.environ.get("DUP", "0"))
BITS = int(os.environ.get("BITS", "0")); BASE = int(os.environ.get("BASE", "0"), 0)
LGS = [int(v) for v in os.environ.get("LGS", "24,26,28,29,30").split(",")]
gen = {"uniform": rs.GEN_UNIFORM, "zipf": rs.GEN_ZIPF}[os.environ.get("GEN", "uniform")]
for t in types:
    d = bench.digits_for(rs, t)
    for lg in LGS:
        n = 1 << lg
        if n * d.elem_bytes > (12 << 30): continue
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
        out = torch.zeros(3, dtype=torch.int64, device="cuda")
        res = []
        for mode in MODES:
            ctx = rs.Context(0); ctx.set_option(rs.OPT_WIDE_SORT, mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for it in range(6):
                ctx.generate_device(x.data_ptr(), n, d, gen, it, 1.0)
                if TRI and d.elem_bytes == 8 and d.key_bytes == 8:  # the sum of two uniform halves: a triangular density (peak 2x the mean)
                    ctx.generate_device(tmp.data_ptr(), n, d, gen, 1000 + it, 1.0)
                    a, b = x.view(torch.int64), tmp.view(torch.int64)
                    a.bitwise_right_shift_(2).bitwise_and_((1 << 61) - 1); b.bitwise_right_shift_(2).bitwise_and_((1 << 61) - 1); a.add_(b)
                if DUP and d.elem_bytes == 8 and d.key_bytes == 8:  # 2^DUP distinct keys, spread out (a multiplicative hash of the top bits)
                    x.view(torch.int64).bitwise_right_shift_(64 - DUP).bitwise_and_((1 << DUP) - 1).mul_(-7046029254386353131)
                if BITS and d.elem_bytes == 8 and d.key_bytes == 8:  # keys of a narrow range: base + uniform below 2^BITS
                    x.view(torch.int64).bitwise_and_((1 << BITS) - 1).bitwise_or_(BASE)
                ctx.verify_device(x.data_ptr(), n, d, out.data_ptr()); torch.cuda.synchronize(); before = out[1].item()
                e0.record(); ctx.sort_device(x.data_ptr(), tmp.data_ptr(), n, d); e1.record(); torch.cuda.synchronize(); ctx.check()
                ctx.verify_device(x.data_ptr(), n, d, out.data_ptr()); torch.cuda.synchronize()
                assert out[0].item() == 0 and out[2].item() == 0 and out[1].item() == before, (t, lg, mode, out.tolist())
                if it >= 2: tot += e0.elapsed_time(e1)
            res.append(tot / 4)
            ctx.close()
        print(f"{t} 2^{lg}: " + "  ".join(f"mode {m}: {r:8.3f} ms ({n/r/1e6:.1f} Gkeys/s)" for m, r in zip(MODES, res)) + (f"  x{res[0]/res[-1]:.2f}" if len(res) > 1 else ""), flush=True)
        del x, tmp
