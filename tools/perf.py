#!/usr/bin/env python3
"""Quick device timing of selected workloads: python tools/perf.py [workload ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
import bench
ctx = rs.default_context(0)
if os.environ.get('MAXR'): ctx.set_option(rs.OPT_MAX_REGIONS, int(os.environ['MAXR']))
wls = sys.argv[1:] or ["c2-256m-u32", "target-1b-u32", "zipf-256m-u64"]
for wl in wls:
    # RSX_DEBUG only means something to a -DRSX_TUNING build (timing ablations with wrong output by design)
    r = bench.run_single(rs, torch, ctx, wl, int(os.environ.get('PERF_STEPS', '5')), 2, check=not os.environ.get("RSX_DEBUG", "0").strip("0x"))
    print(f"{wl:28s} {r['ms_per_sort']:8.3f} ms  {r['gkeys_per_s']:7.2f} Gkeys/s  whole {r['frac_of_hbm_peak']*100:5.1f}%  "
          f"sweep {r.get('sweep_ms_per_launch',0):.4f} ms = {r.get('sweep_gbps',0):7.1f} GB/s ({r.get('sweep_gbps',0)/80:.1f}%)  hist {r.get('hist_ms_per_launch',0):.4f} ms", flush=True)
