#!/usr/bin/env python3
"""Chain-length sensitivity: sweep time against the number of look-back chains (RSX_OPT_MAX_REGIONS).
usage: python tools/regions_probe.py [workload] [regions ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2-256m-u32"
regs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32]
for r in regs:
    ctx = rs.Context(0)
    ctx.set_option(rs.OPT_MAX_REGIONS, r)
    res = bench.run_single(rs, torch, ctx, wl, 5, 2)
    print(f"{wl} regions<={r:2d}: {res['ms_per_sort']:8.3f} ms  {res['gkeys_per_s']:7.2f} Gkeys/s  sweep {res.get('sweep_ms_per_launch',0):.4f} ms = {res.get('sweep_gbps',0):7.1f} GB/s", flush=True)
    ctx.close()
