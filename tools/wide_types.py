"""Sort rate of the less common element sizes (12, 16 with a 16-byte key, 24, 32 bytes; 1 and 2 bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, radix_sort_amd as rs
ctx = rs.default_context(0)
cases = {"u8": (1, 0, 1, 0), "u16": (2, 0, 2, 0), "(u32,[u8;8])": (12, 0, 4, 0), "u128": (16, 0, 16, 0),
         "(u64,[u64;2])": (24, 0, 8, 0), "(u128,u128)": (32, 0, 16, 0)}
for name, lay in cases.items():
    d = rs.RadixDigits(*lay)
    n = (1 << 31) // max(8, d.elem_bytes)  # 2 GiB of 8+ byte elements, 256M narrow ones
    x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda"); tmp = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for it in range(5):
        ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, it)
        e0.record(); rs.radix_sort(x, digits=d, tmp=tmp); e1.record(); torch.cuda.synchronize()
        if it >= 2: tot += e0.elapsed_time(e1)
    ms = tot / 3
    gbps = d.key_bytes * 2 * n * d.elem_bytes / ms / 1e6
    print(f"{name:16s} n={n:>10d} {ms:8.3f} ms  {n/ms/1e6:8.2f} Gkeys/s  whole {gbps/80:5.1f}% of 8 TB/s", flush=True)
    ctx.check(); del x, tmp
