#!/usr/bin/env python3
"""Per-pass sweep times of one sort (HIP events around every launch): python tools/passes.py [workload ...]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
from radix_sort_amd import _lib
import bench
L = _lib.load()
L.rsx_debug_sweep_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
ctx = rs.default_context(0)
if os.environ.get('HOT_LANES'): ctx.set_option(rs.OPT_HOT_LANES, int(os.environ['HOT_LANES']))
for wl in sys.argv[1:] or ["c2-256m-u32"]:
    t, logn, gen, param, _ = bench.WORKLOADS[wl]
    d = bench.digits_for(rs, t)
    r = bench.run_single(rs, torch, ctx, wl, 5, 2)
    buf = (ctypes.c_float * 4096)()
    k = L.rsx_debug_sweep_times(ctx._h, buf, 4096)
    D = d.key_bytes
    per = [sum(buf[i] for i in range(p, k, D)) / max(1, len(range(p, k, D))) for p in range(D)]
    gb = 2 * (1 << logn) * d.elem_bytes / 1e6
    print(f"{wl:26s} hist {r.get('hist_ms_per_launch',0):.4f} | " + " ".join(f"p{p}:{per[p]:.4f}({gb/per[p]:.0f})" for p in range(D)), flush=True)
