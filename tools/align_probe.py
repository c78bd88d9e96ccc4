"""How much do unaligned digit runs cost the sweep?  One pass (rsx_partition_device, digit 0) over
2^28 u32 keys whose low byte is (a) uniform random, (b) arranged so that every tile holds exactly 32
keys of each digit value in input order groups -- every run then starts on a 128-byte line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
d = rs.PRIMITIVES["u32"]; n = 1 << 28
g = torch.Generator(device="cuda"); g.manual_seed(1)
hi = torch.randint(0, 1 << 24, (n,), dtype=torch.int32, device="cuda", generator=g) << 8
i = torch.arange(n, dtype=torch.int32, device="cuda")
cases = {
    "uniform random low byte": hi | torch.randint(0, 256, (n,), dtype=torch.int32, device="cuda", generator=g),
    "32 of each value per tile, grouped": hi | ((i >> 5) & 255),
    "32 of each value per tile, interleaved": hi | (i & 255),
    "64-byte aligned runs (16-groups)": hi | ((i >> 4) & 255),
}
dst = torch.empty(n, dtype=torch.int32, device="cuda")
for name, x in cases.items():
    for it in range(3):
        ctx.profile(True)
        ctx.partition_device(x.data_ptr(), dst.data_ptr(), n, d, 0)
        torch.cuda.synchronize()
        p = ctx.profile_read()
    ms = p["sweep"][0] / p["sweep"][1]
    print(f"{name:42s} sweep {ms:.4f} ms = {2 * n * 4 / ms / 1e6:7.1f} GB/s")
