"""PCIe-inclusive rate of the literal `&mut [T]` drop-in (rsx_sort_host): H2D + sort + D2H from
pageable host memory.  Reported in DESIGN.md; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radix_sort_amd as rs
for logn in [int(x) for x in sys.argv[1:]] or (24, 28):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 2**32, size=n, dtype=np.uint32)
    rs.radix_sort(a.copy())  # warm up (staging buffers)
    ts = []
    for _ in range(7 if logn <= 22 else 3):
        b = a.copy(); t = time.perf_counter(); rs.radix_sort(b); ts.append(time.perf_counter() - t)
    assert (np.diff(b.astype(np.int64)) >= 0).all()
    print(f"rsx_sort_host u32 n=2^{logn}: {min(ts)*1e3:.1f} ms  {n/min(ts)/1e9:.2f} Gkeys/s  ({2*4*n/min(ts)/1e9:.1f} GB/s over PCIe both ways)")
