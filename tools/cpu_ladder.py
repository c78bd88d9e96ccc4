#!/usr/bin/env python3
"""the reference's CPU optimisation ladder (radix_sort0..5, mod.rs:178-571) timed on this
host as ablation baselines, protocol of main.rs:26-44 (mean of `runs` sorts of fresh data, the sort call
only).  usage: python tools/cpu_ladder.py [log2 n] [type] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = sys.argv[2] if len(sys.argv) > 2 else "(u32,u32)"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
es, kb = {"u32": (4, 4), "u64": (8, 8), "(u32,u32)": (8, 4), "(u64,u64)": (16, 8)}[t]
lay = oracle.Layout(es, 0, kb, 0)
n, cores = 1 << logn, os.cpu_count() or 1
names = ["radix_sort0 single thread", "radix_sort1 thread per digit", "radix_sort2 chunk per thread", "radix_sort3 + page-touched scratch",
         "radix_sort4 + work pool (2 chunks/worker)", "radix_sort5 + 96-element write buffers (production)"]
rng = np.random.default_rng(0)
print(f"{n} {t} uniform keys, {cores} threads, mean of {runs} runs")
for v, name in enumerate(names):
    tot = 0.0
    for _ in range(runs):
        raw = rng.integers(0, 256, size=n * es, dtype=np.uint8)
        if es != kb:
            raw.reshape(n, es)[:, kb:] = 0
        t0 = time.perf_counter()
        oracle.sort_variant_inplace(raw, lay, cores, v)
        tot += time.perf_counter() - t0
    print(f"  {name:52s} {tot / runs:8.4f} s  {n / (tot / runs) / 1e9:7.3f} Gkeys/s", flush=True)
