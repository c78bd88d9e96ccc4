#!/usr/bin/env python3
"""Rebuilds profiles/pmc_traffic.json from profiles/<tag>_<workload>_pmc.txt (the summaries tools/profile_all.sh
writes): HBM bytes per rsx_sweep_kernel launch = 2 * FETCH_SIZE + WRITE_SIZE (KB of 1024 B, separate --pmc passes;
FETCH_SIZE doubled per MI355X_MICROARCH.md and checked on rsx_hist_kernel, which reads exactly n*s bytes).
usage: python tools/pmc_traffic.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, radix_sort_amd as rs
tag = sys.argv[1]
out = {"_comment": "HBM traffic of the dominant kernel (rsx_sweep_kernel) per launch, from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                   "(separate passes, tools/profile_all.sh; rebuilt by tools/pmc_traffic.py), averaged over all sweep launches of the profiled "
                   "run (passes with and without the next-pass count weighted by their launch counts). FETCH_SIZE is doubled (gfx950 reports "
                   "half the bytes of a wide streaming read, MI355X_MICROARCH.md HBM section; calibrated here on rsx_hist_kernel, which reads "
                   "exactly n*s bytes and reports n*s/2). Counter unit: KB (1024 B)."}
for wl, (t, logn, *_r) in bench.WORKLOADS.items():
    f = os.path.join(ROOT, "profiles", f"{tag}_{wl}_pmc.txt")
    if not os.path.exists(f):
        continue
    tot = {("sweep", "FETCH_SIZE"): [0.0, 0], ("sweep", "WRITE_SIZE"): [0.0, 0], ("hist", "FETCH_SIZE"): [0.0, 0]}
    per_kernel = {}  # kernel -> {counter: per-dispatch KB}: the paths without a sweep (counting, middle-size split)
    disp = {}        # kernel -> dispatches
    sums = {}        # kernel -> {counter: total KB over all its instantiations}
    for line in open(f):
        m = re.match(r".*rsx_(sweep|hist)_kernel<.*>\s+(\w+)\s+total\s+(\S+)\s+per-dispatch\s+\S+\s+dispatches\s+(\d+)", line)
        if m and (m.group(1), m.group(2)) in tot:
            tot[(m.group(1), m.group(2))][0] += float(m.group(3))
            tot[(m.group(1), m.group(2))][1] += int(m.group(4))
        m = re.match(r".*(rsx_\w+_kernel)(?:<[^>]*>)?\s+(FETCH_SIZE|WRITE_SIZE)\s+total\s+(\S+)\s+per-dispatch\s+(\S+)\s+dispatches\s+(\d+)", line)
        if m:
            per_kernel.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
            disp[m.group(1)] = int(m.group(5))
            # several instantiations of one kernel (the hybrid's bucket kernel: every form is enqueued, one runs): totals
            sums.setdefault(m.group(1), {}).setdefault(m.group(2), 0.0)
            sums[m.group(1)][m.group(2)] += float(m.group(3))
    d = bench.digits_for(rs, t)
    n = 1 << logn
    if not tot[("sweep", "FETCH_SIZE")][1]:  # no sweep in this path: HBM bytes of the whole sort, all its kernels
        whole = int(round(sum(2 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0) for k in per_kernel.values()) * 1024))
        alg = d.key_bytes * 2 * n * d.elem_bytes
        out[wl] = {"round": tag, "path_without_sweep_kernel": True, "kernels": sorted(per_kernel),
                   "traffic_bytes_per_sort": whole, "algorithmic_bytes_per_sort": alg,
                   "traffic_over_algorithmic": round(whole / alg, 4), "source": f"profiles/{tag}_{wl}_pmc.txt"}
        print(wl, "whole sort", out[wl]["traffic_over_algorithmic"])
        continue
    hybrid = "rsx_bucket16_kernel" in per_kernel
    if hybrid:  # the wide-key hybrid: per sort 2 sweeps run, the launches of the refused sequence return at once (no traffic)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            tot[("sweep", c)][1] = 2 * disp["rsx_count16top_kernel"]
    fetch = tot[("sweep", "FETCH_SIZE")][0] / tot[("sweep", "FETCH_SIZE")][1]
    write = tot[("sweep", "WRITE_SIZE")][0] / tot[("sweep", "WRITE_SIZE")][1]
    hist = tot[("hist", "FETCH_SIZE")][0] / max(1, tot[("hist", "FETCH_SIZE")][1])
    traffic = int(round((2 * fetch + write) * 1024))
    alg = 2 * n * d.elem_bytes
    out[wl] = {"round": tag, "fetch_size_kb_raw_per_launch": int(round(fetch)), "write_size_kb_per_launch": int(round(write)),
               "hist_kernel_fetch_size_kb_raw": int(round(hist)), "hist_kernel_bytes_read": n * d.elem_bytes,
               "hist_calibration_2x_fetch_over_bytes": round(2 * hist * 1024 / (n * d.elem_bytes), 5),
               "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
               "traffic_over_algorithmic": round(traffic / alg, 4), "source": f"profiles/{tag}_{wl}_pmc.txt"}
    if hybrid:
        out[wl].pop("hist_kernel_fetch_size_kb_raw"); out[wl].pop("hist_calibration_2x_fetch_over_bytes"); out[wl].pop("hist_kernel_bytes_read")
        out[wl]["path"] = "wide-key hybrid: sweep figures are per REAL sweep launch (2 per sort)"
        for kn in ("rsx_bucket16_kernel", "rsx_count16top_kernel"):
            k = {c: v / max(1, disp.get("rsx_count16top_kernel", 1)) for c, v in sums.get(kn, {}).items()}  # per SORT: one real launch each
            out[wl][kn.replace("rsx_", "").replace("_kernel", "")] = {
                "traffic_bytes_per_launch": int(round((2 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024)),
                "array_bytes": n * d.elem_bytes}
        print(wl, "hybrid", out[wl]["traffic_over_algorithmic"], out[wl]["bucket16"], out[wl]["count16top"])
        continue
    print(wl, out[wl]["traffic_over_algorithmic"], out[wl]["hist_calibration_2x_fetch_over_bytes"])
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=2)
