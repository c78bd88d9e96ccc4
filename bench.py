#!/usr/bin/env python3
"""bench.py -- Gkeys/s of the MI355X LSD radix sort on BASELINE.json's configurations.

A "step" is one full sort (all D passes of the hot path) of one batch of synthetic
input already resident in HBM.  K steps sort K different pre-generated batches, so
every timed sort sees unsorted data and generation stays outside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or typed plainly: the parent
  process then starts exactly that launcher as a child BEFORE it touches the GPU, relays rank 0's JSON line and
  exits with the launcher's return code.

Prints ONE JSON line on rank 0 (contract in the round prompt): `value` = Gkeys/s of the
whole job; `roofline` = dominant kernel (rsx_sweep_kernel, one launch = one pass,
algorithmic bytes 2*n*s per launch) timed with HIP events on its own launch stream over
the timed region; `cpu_baseline` = the oracle (thread-parallel C restatement of the
reference, oracle/rsx_oracle.c) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# name -> (type, log2 n per GPU, generator, param, description)
WORKLOADS = {
    "c1-1m-u32": ("u32", 20, "uniform", 0.0, "1M u32 uniform keys (BASELINE.json configs[0]'s size, here on the GPU: the middle-size path)"),
    "c2-256m-u32": ("u32", 28, "uniform", 0.0, "256M u32 uniform keys, 8-bit radix (BASELINE.json configs[1])"),
    "target-1b-u32": ("u32", 30, "uniform", 0.0, "1B u32 uniform keys (north-star target)"),
    "c3-1b-u64": ("u64", 30, "uniform", 0.0, "1B u64 uniform keys, 8 passes (configs[2])"),
    "c4-slice-512m-u32": ("u32", 29, "uniform", 0.0, "2^29 u32 per GPU (configs[3] slice)"),
    "c5-slice-128m-pairs-zipf": ("(u64,u64)", 27, "zipf", 1.0, "2^27 (u64,u64) Zipf pairs per GPU (configs[4] slice)"),
    "zipf-256m-u32": ("u32", 28, "zipf", 1.0, "256M u32 Zipf(s=1) keys"),
    "step16-256m-u32": ("u32", 28, "step", 16.0, "256M u32 step-uniform(16) keys"),
    "zipf-256m-u64": ("u64", 28, "zipf", 1.0, "256M u64 Zipf(s=1) keys"),
    "pairs-256m-u32u32": ("(u32,u32)", 28, "uniform", 0.0, "256M (u32,u32) pairs (reference bench type, main.rs:112)"),
    "f32-256m": ("f32", 28, "uniform", 0.0, "256M f32 keys (uniform bit patterns: NaNs, infinities, both signs)"),
    "i32-256m": ("i32", 28, "uniform", 0.0, "256M i32 uniform keys"),
    "f64-128m": ("f64", 27, "uniform", 0.0, "128M f64 keys (uniform bit patterns)"),
    "sorted-256m-u32": ("u32", 28, "sorted", 0.0, "256M u32 keys already in order (key = index)"),
    "reversed-256m-u32": ("u32", 28, "reversed", 0.0, "256M u32 keys in reverse order"),
    "u64-128m": ("u64", 27, "uniform", 0.0, "128M u64 uniform keys"),
    "u64-256m": ("u64", 28, "uniform", 0.0, "256M u64 uniform keys"),
    "u64-512m": ("u64", 29, "uniform", 0.0, "512M u64 uniform keys"),
    "u16-256m": ("u16", 28, "uniform", 0.0, "256M u16 uniform keys (2 passes)"),
    "u8-256m": ("u8", 28, "uniform", 0.0, "256M u8 uniform keys (counting path)"),
    "pairs-128m-u64u64": ("(u64,u64)", 27, "uniform", 0.0, "128M (u64,u64) pairs (reference bench type, main.rs:123)"),
    "u128-128m": ("u128", 27, "uniform", 0.0, "128M u128 uniform keys (16 digits)"),
    "u64-40bit-256m": ("u64", 28, "uniform", 0.0, "256M u64 keys below 2^40 (uniform): the hybrid's window follows the keys' range"),
    "u64-range-1b": ("u64", 30, "uniform", 0.0, "1B u64 keys of ONE of 32 value ranges (5 fixed top bits): what a rank sorts after a multi-GPU exchange"),
}
# workloads whose uniform keys are cut to a range after generation: name -> (bits, base)
KEY_RANGE = {"u64-40bit-256m": (40, 0), "u64-range-1b": (59, 0x5 << 59)}
HEADLINE = "c3-1b-u64"  # the largest single-GPU configuration in BASELINE.json's configs (configs[2])
EXTRA_DEFAULT = ["target-1b-u32", "c2-256m-u32", "c4-slice-512m-u32", "zipf-256m-u32", "step16-256m-u32", "zipf-256m-u64", "c5-slice-128m-pairs-zipf",
                 "pairs-128m-u64u64", "u128-128m", "u64-40bit-256m", "c1-1m-u32", "u16-256m"]


PATH_NAMES = ["general passes", "one-launch sort", "middle-size bucket split", "one-byte counting", "two-byte counting",
              "wide-key hybrid (top 16 bits by two sweeps, the other digits in LDS)"] + ["?"] * 10


def digits_for(rs, t):
    if t.startswith("("):
        k, p = t[1:-1].split(",")
        return rs.tuple_of(k, int(p[1:]) // 8)
    return rs.PRIMITIVES[t]


def gen_id(rs, name):
    return {"uniform": rs.GEN_UNIFORM, "zipf": rs.GEN_ZIPF, "step": rs.GEN_STEP, "sorted": rs.GEN_SORTED,
            "reversed": rs.GEN_REVERSED}[name]


def run_single(rs, torch, ctx, wl, steps, warmup, seed0=0x5EED0000, profile=True, check=True):
    """Times `steps` sorts of `steps` different batches on the current device. Returns dict."""
    t, logn, gen, param, _ = WORKLOADS[wl]
    d = digits_for(rs, t)
    n = 1 << logn
    nbytes = n * d.elem_bytes
    free, _total = torch.cuda.mem_get_info()
    pool = max(1, min(steps, int((free * 0.8 - nbytes) // nbytes)))
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(pool)]
    tmp = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ctx.reserve(n, d)
    ctx.set_option(rs.OPT_WIDE_SORT, 1)  # (the default; setting it makes the context forget what an earlier workload's keys looked like)
    stream = torch.cuda.current_stream().cuda_stream

    def fill(i, b):
        ctx.generate_device(b.data_ptr(), n, d, gen_id(rs, gen), seed0 + i, param, 0, stream)
        if wl in KEY_RANGE:  # (8-byte keys only)
            bits, base = KEY_RANGE[wl]
            b.view(torch.int64).bitwise_and_((1 << bits) - 1).bitwise_or_(base)

    for i in range(warmup):
        fill(1000 + i, bufs[0])
        ctx.sort_device(bufs[0].data_ptr(), tmp.data_ptr(), n, d, stream)
    done = 0
    total_ms = 0.0
    prof_tot = {"sweep": [0.0, 0], "hist": [0.0, 0], "scan": [0.0, 0], "other": [0.0, 0]}
    while done < steps:  # pool-sized rounds (pool == steps unless memory is short)
        k = min(pool, steps - done)
        for i in range(k):
            fill(done + i, bufs[i])
        torch.cuda.synchronize()
        if profile:
            ctx.profile(True)
        t0 = time.perf_counter()
        for i in range(k):
            ctx.sort_device(bufs[i].data_ptr(), tmp.data_ptr(), n, d, stream)
        torch.cuda.synchronize()
        total_ms += (time.perf_counter() - t0) * 1e3
        if profile:
            pr = ctx.profile_read()
            ctx.profile(False)
            for kname in prof_tot:
                prof_tot[kname][0] += pr[kname][0]
                prof_tot[kname][1] += pr[kname][1]
        done += k
    if check:
        ctx.check()
    lp = ctx.get_info(rs.INFO_LAST_PASSES)  # tile schedule of the last timed sort's passes
    # untimed sanity: last batch is sorted and is a permutation of its input
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    ctx.verify_device(bufs[k - 1].data_ptr(), n, d, out.data_ptr(), stream)
    v = out.cpu().tolist()
    assert not check or (v[0] == 0 and v[2] == 0), f"bench output not sorted/stable: {v}"
    del bufs, tmp
    torch.cuda.empty_cache()
    ms = total_ms / steps
    D = d.key_bytes
    res = {
        "workload": wl, "type": t, "n": n, "elem_bytes": d.elem_bytes, "passes": D,
        "ms_per_sort": ms, "gkeys_per_s": n / ms / 1e6,
        "algorithmic_gbps": D * 2 * n * d.elem_bytes / ms / 1e6,
        "frac_of_hbm_peak": D * 2 * n * d.elem_bytes / ms / 1e6 / HBM_PEAK_GBPS,
    }
    path = (lp >> 24) & 0xF
    res["paths"] = {"rank_atomic": ctx.get_info(rs.INFO_RANK_ATOMIC), "l2_local": ctx.get_info(rs.INFO_L2_LOCAL),
                    "static_tiles": f"{(lp >> 8) & 0xFF}/{lp & 0xFF}", "placement_verified": f"{(lp >> 16) & 0xFF}/{lp & 0xFF}",
                    "path": PATH_NAMES[path]}
    if profile and prof_tot["sweep"][1]:
        # wide-key hybrid: both kernel sequences are enqueued and the device runs one (the other's launches return at
        # once, ~5 us each, and are inside these sums): per sort 2 real sweeps, 1 bucket kernel, 1 count + 1 marginal
        real = 2 * steps if path == 5 else D * steps if path == 0 and prof_tot["sweep"][1] > D * steps else prof_tot["sweep"][1]
        sw_ms = prof_tot["sweep"][0] / real
        res["sweep_ms_per_launch"] = sw_ms
        res["sweep_launches"] = real
        res["sweep_gbps"] = 2 * n * d.elem_bytes / sw_ms / 1e6
        if prof_tot["hist"][1]:
            res["hist_ms_per_launch"] = prof_tot["hist"][0] / (steps if path == 5 else prof_tot["hist"][1])
        if path == 5 and prof_tot["other"][1]:
            bk_ms = prof_tot["other"][0] / steps
            res["bucket_ms_per_launch"] = bk_ms
            res["bucket_launches"] = steps
            res["bucket_passes"] = D - 2
            res["bucket_algorithmic_gbps"] = (D - 2) * 2 * n * d.elem_bytes / bk_ms / 1e6
            res["bucket_hbm_gbps"] = 2 * n * d.elem_bytes / bk_ms / 1e6
            res["count16_ms_per_sort"] = prof_tot["hist"][0] / steps
            res["scan16_ms_per_sort"] = prof_tot["scan"][0] / steps
    return res


def _host_batch(np, n, es, kb, salt):
    """n elements of uniform keys (payload 0, like KeyUniform, distr.rs:42-52) at memory speed: a 2^22-element
    random block repeated under a different odd multiplier per repetition (a bijection of the 64-bit words: the keys
    stay uniform and every repetition differs)."""
    from concurrent.futures import ThreadPoolExecutor
    words = n * es // 8
    blk = 1 << 22
    base = np.random.default_rng(salt).integers(0, 1 << 63, size=min(blk, words), dtype=np.uint64)
    out = np.empty(words, dtype=np.uint64)

    def fill(i):  # numpy releases the GIL inside the multiply: the blocks are filled (and first-touched) in parallel
        off = i * blk
        m = min(blk, words - off)
        np.multiply(base[:m], np.uint64(2 * (salt * 4099 + i) + 0x9E3779B97F4A7C15), out=out[off:off + m])
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as pool:
        list(pool.map(fill, range((words + blk - 1) // blk)))
    raw = out.view(np.uint8)
    if es != kb:
        raw.reshape(n, es)[:, kb:] = 0
    return raw


def cpu_baseline(t_name, n_full, budget_seconds=30.0):
    """The oracle (port of mod.rs:61-176) on this host's cores; protocol of main.rs:26-44: mean of 5 runs on fresh
    data, timed region = the sort call incl. temp alloc + page touch.  Runs at the configuration's own n when five
    sorts of it fit `budget_seconds` (rate calibrated on 2^26 keys), else on the largest power of two that does."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    es, kb = {"u32": (4, 4), "u64": (8, 8), "(u64,u64)": (16, 8), "(u32,u32)": (8, 4)}[t_name]
    lay = oracle.Layout(es, 0, kb, 0)
    cores = os.cpu_count() or 1

    def one(n, salt):
        raw = _host_batch(np, n, es, kb, salt)
        t0 = time.perf_counter()
        oracle.sort_parallel_inplace(raw, lay, cores)
        return time.perf_counter() - t0

    n_cal = min(n_full, 1 << 26)
    one(1 << 22, 99)  # thread pools, page cache
    rate = n_cal / one(n_cal, 100)  # keys/s
    n = int(min(n_full, max(1 << 22, rate * budget_seconds / 5)))
    n = 1 << (n.bit_length() - 1)
    runs = [one(n, 1 + i) for i in range(5)]
    mean = sum(runs) / len(runs)
    own = ", the configuration's own n" if n == n_full else ""
    return {"value": n / mean / 1e9, "unit": "Gkeys/s", "cores": cores, "kind": "port",
            "sample": f"{n} {t_name} uniform keys (2^{n.bit_length()-1}{own}), mean of 5 runs, {cores} threads, "
                      f"timed like main.rs:32-34 (temp alloc + page touch inside)"}


def cpu_ladder(logn=24, t_name="(u32,u32)", runs=3):
    """the reference's optimisation ladder radix_sort0..5 (mod.rs:178-571, restated in the oracle) timed on
    THIS host's cores, protocol of main.rs:26-44.  Bounded: 2^24 pairs, 3 runs per rung."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    es, kb = {"u32": (4, 4), "u64": (8, 8), "(u32,u32)": (8, 4), "(u64,u64)": (16, 8)}[t_name]
    lay = oracle.Layout(es, 0, kb, 0)
    n, cores = 1 << logn, os.cpu_count() or 1
    names = ["radix_sort0 (single thread)", "radix_sort1 (thread per digit)", "radix_sort2 (chunk per thread)",
             "radix_sort3 (+ page-touched scratch)", "radix_sort4 (+ work pool)", "radix_sort5 (+ 96-element write buffers)"]
    rungs = {}
    for v, name in enumerate(names):
        tot = 0.0
        for r in range(runs):
            raw = _host_batch(np, n, es, kb, 7 + r)
            t0 = time.perf_counter()
            oracle.sort_variant_inplace(raw, lay, cores, v)
            tot += time.perf_counter() - t0
        rungs[name] = {"seconds": tot / runs, "gkeys_per_s": n / (tot / runs) / 1e9}
    return {"sample": f"{n} {t_name} uniform keys, mean of {runs} runs", "cores": cores, "kind": "port", "rungs": rungs}


def roofline_of(res, workload, n, d):
    """The `roofline` object of the JSON line: the dominant kernel of the sort that ran.  General passes: rsx_sweep_kernel
    (one launch = one digit pass, 2*n*s algorithmic bytes).  Wide-key hybrid: rsx_bucket16_kernel -- one launch does
    D-2 of the algorithm's digit passes, so its algorithmic bytes are (D-2)*2*n*s (2*n*s per pass);
    those passes run in LDS, the kernel's own HBM traffic is 2*n*s, and both rates are given."""
    sweep = {
        "bound": "hbm", "kernel": "rsx_sweep_kernel (one launch = one digit pass)",
        "achieved": res.get("sweep_gbps"), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": (res.get("sweep_gbps") or 0.0) / HBM_PEAK_GBPS,
        "algorithmic_bytes_per_launch": 2 * n * d.elem_bytes,
        "avg_launch_ms": res.get("sweep_ms_per_launch"), "launches_timed": res.get("sweep_launches"),
        "traffic": pmc_traffic(workload),
        "traffic_note": "HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, separate passes), profiles/pmc_traffic.json",
    }
    if "bucket_ms_per_launch" not in res:
        return sweep
    D = d.key_bytes
    return {
        "bound": "hbm", "kernel": f"rsx_bucket16_kernel (one launch = {D - 2} digit passes of every 16-bit bucket, in LDS)",
        "achieved": res["bucket_algorithmic_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": res["bucket_algorithmic_gbps"] / HBM_PEAK_GBPS,
        "algorithmic_bytes_per_launch": (D - 2) * 2 * n * d.elem_bytes,
        "avg_launch_ms": res["bucket_ms_per_launch"], "launches_timed": res["bucket_launches"],
        "hbm_bytes_per_launch_by_design": 2 * n * d.elem_bytes,
        "hbm_gbps_by_design": res["bucket_hbm_gbps"], "hbm_frac_by_design": res["bucket_hbm_gbps"] / HBM_PEAK_GBPS,
        "traffic": pmc_traffic(workload, "bucket16"),
        "note": "frac counts the algorithm's bytes (2*n*s per digit pass, SURVEY 8(d)) for the D-2 passes this launch stands for; "
                "it can exceed 1 because they run in LDS -- and only as many of the top ones as a bucket of that size needs are "
                "run (three at 2^30 u64), the neighbours that still agree afterwards are mended by the skipped digits -- while the kernel reads and writes the array once "
                "(hbm_*_by_design; traffic = rocprofv3 PMC)",
        "per_sort_ms": {"count16 + marginal": res.get("count16_ms_per_sort"), "total16 + scan16": res.get("scan16_ms_per_sort"),
                        "sweeps (2)": 2 * res["sweep_ms_per_launch"], "bucket16": res["bucket_ms_per_launch"]},
        "sweep": sweep,
    }


def pmc_traffic(workload, kernel="sweep"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json; bench.py cannot run the profiler on itself)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(workload, {})
            if kernel != "sweep":
                e = e.get(kernel, {})
            return e.get("traffic_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3-1b-u64 (configs[2], the largest single-GPU configuration) at N=1 and per GPU at N>1 (weak scaling)")
    ap.add_argument("--extra", default=",".join(EXTRA_DEFAULT),
                    help="comma list of further workloads measured with the same --steps/--warmup and reported under 'extra' at N=1; '' = none")
    ap.add_argument("--extra-sharded", default="c4-slice-512m-u32",
                    help="N>1: comma list of further per-GPU workloads reported under 'extra' (default: the configs[3] slice, "
                         "2^29 u32 per GPU = 4B keys at N=8); '' = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 collectives: nccl (= RCCL over xGMI, the measured path); gloo only to rehearse the N>1 code "
                         "path on a box with fewer GPUs than ranks (ranks then share devices, exchange staged through the host)")
    ap.add_argument("--exchange", default="first", choices=["first", "one", "per-pass"],
                    help="multi-GPU schedule: 'first' = partition by the top digit, ONE all-to-all, one local sort (default); "
                         "'one' = local sort, one all-to-all, local sort; 'per-pass' = one all-to-all per digit pass "
                         "(the reference's loop with chunk == GPU, the north star's wording)")
    ap.add_argument("--chunks", type=int, default=4,
                    help="--exchange first: ranges the exchange is cut into so that the local sort of one range runs while "
                         "the next ones are on the links (1 = no overlap)")
    ap.add_argument("--sub-ranges", type=int, default=4,
                    help="--exchange first: position sub-ranges the partition pass is cut into so that the first batches are on "
                         "the links while the later sub-ranges are still being scattered (1 = partition first, then exchange)")
    ap.add_argument("--cpu-ladder", action="store_true",
                    help="N=1: also time the reference's CPU ladder radix_sort0..5 (oracle restatement) on this host: extra.cpu_ladder")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = HEADLINE  # N > 1: the same configuration per GPU (weak scaling of the N = 1 line)

    if args.gpus > 1 and "RANK" not in os.environ:
        # typed plainly: start the N ranks as fresh children (nothing in this process has touched the GPU: no
        # torch.cuda call, no radix_sort_amd context) and pass their verdict on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)

    import torch
    import radix_sort_amd as rs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    ctx = rs.default_context(local_rank)
    t, logn, gen, param, desc = WORKLOADS[args.workload]
    d = digits_for(rs, t)
    n = 1 << logn

    if world == 1:
        res = run_single(rs, torch, ctx, args.workload, args.steps, args.warmup)
        ms = res["ms_per_sort"]
        line = {
            "metric": "Gkeys/sec (u32)" if t == "u32" else f"Gkeys/sec ({t})",
            "value": res["gkeys_per_s"], "unit": "Gkeys/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u%d" % (8 * d.key_bytes) if t[0] in "u(" else t, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "n_keys": n, "elem_bytes": d.elem_bytes,
                       "passes": d.key_bytes, "radix_bits": 8, "generator": gen,
                       "algorithmic_bytes_per_sort": d.key_bytes * 2 * n * d.elem_bytes,
                       "whole_sort_algorithmic_gbps": res["algorithmic_gbps"],
                       "whole_sort_frac_of_hbm_peak": res["frac_of_hbm_peak"]},
            "roofline": roofline_of(res, args.workload, n, d),
            "paths": res["paths"],
        }
        extra = {}
        for wl in [w for w in args.extra.split(",") if w and w != args.workload]:
            try:
                # (per-launch event pairs would be most of a 40 us sort: the small ones are timed without)
                r = run_single(rs, torch, ctx, wl, args.steps, args.warmup, profile=WORKLOADS[wl][1] > 22)
                extra[wl] = {k: r[k] for k in ("n", "type", "ms_per_sort", "gkeys_per_s", "algorithmic_gbps",
                                               "frac_of_hbm_peak") if k in r}
                extra[wl]["sweep_gbps"] = r.get("sweep_gbps")
                extra[wl]["sweep_frac_of_hbm_peak"] = (r.get("sweep_gbps") or 0.0) / HBM_PEAK_GBPS
                extra[wl]["steps"] = args.steps
                extra[wl]["sweep_traffic_bytes_per_launch"] = pmc_traffic(wl)
                extra[wl]["paths"] = r["paths"]
                if "bucket_ms_per_launch" in r:  # the hybrid: its three stages per sort
                    extra[wl]["hybrid_ms"] = {"count16": r.get("count16_ms_per_sort"), "sweeps (2)": 2 * r["sweep_ms_per_launch"],
                                              "bucket16": r["bucket_ms_per_launch"]}
            except Exception as e:  # noqa: BLE001  (an extra must never kill the headline)
                extra[wl] = {"error": repr(e)}
        if args.cpu_ladder:
            extra["cpu_ladder"] = cpu_ladder()
        if extra:
            line["extra"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(t if t in ("u32", "u64", "(u64,u64)", "(u32,u32)") else "u32", n)
        print(json.dumps(line), flush=True)
        return

    # ---- N > 1: one slice per rank; weak scaling of the N = 1 headline (same type and n per GPU) --------------
    import torch.distributed as dist
    from radix_sort_amd.sharded import ShardedRadixSort
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")
    sorter = ShardedRadixSort()
    agree = dist.new_group(backend="gloo")  # host-side agreement between the ranks (never carries data)
    stream = torch.cuda.current_stream().cuda_stream
    cdev = "cuda" if args.backend == "nccl" else "cpu"
    state = {"exchange": args.exchange, "note": ""}

    def run_sharded(wl):
        t, logn, gen, param, desc = WORKLOADS[wl]
        d = digits_for(rs, t)
        n = 1 << logn
        nbytes = n * d.elem_bytes
        free, _total = torch.cuda.mem_get_info()
        share = max(1, min(world, world // max(1, torch.cuda.device_count())))  # gloo rehearsal: ranks sharing a device
        pool = max(1, min(args.steps, int((free * 0.8 / share - 4 * nbytes) // nbytes)))
        bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(pool)]
        n_per_rank = [n] * world

        def fill(i, b):
            ctx.generate_device(b.data_ptr(), n, d, gen_id(rs, gen), 0x5EED0000 + i, param, rank * n, stream)

        def run(b):  # -> the tensor that holds the sorted slice (the donated form may hand back another buffer)
            if state["exchange"] == "first":
                return sorter.sort_exchange_first(b, d, n_per_rank, chunks=args.chunks, sub_ranges=args.sub_ranges, donate=True)
            if state["exchange"] == "one":
                sorter.sort_one_exchange(b, d, n_per_rank)
            else:
                sorter.sort(b, d, n_per_rank)
            return b

        for i in range(max(1, args.warmup)):
            fill(1000 + i, bufs[0])
            ok, why = 1, ""
            try:
                bufs[0] = run(bufs[0])
                torch.cuda.synchronize()
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=agree)
            if flag.item() == 0 and state["exchange"] == "first":
                # the batched point-to-point exchange has never run between two real GPUs in the builder's hands: if it
                # fails on this node, every rank falls back to the schedule built on ONE all_to_all_single
                state["exchange"] = "one"
                state["note"] = "exchange-first raised on some rank (%s): fell back to --exchange one" % (why or "another rank")
                fill(1000 + i, bufs[0])
                run(bufs[0])
                torch.cuda.synchronize()
            elif flag.item() == 0:
                raise SystemExit("rank %d: sort failed: %s" % (rank, why or "on another rank"))
        done, elapsed = 0, 0.0
        while done < args.steps:  # pool-sized rounds (pool == steps unless memory is short)
            k = min(pool, args.steps - done)
            for i in range(k):
                fill(done + i, bufs[i])
            sum_before = sorter.checksum(bufs[k - 1], d)  # multiset checksum of the round's last batch, before
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                bufs[i] = run(bufs[i])
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            elapsed += time.perf_counter() - t0
            done += k
        el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        # untimed global check of the last batch (ShardedRadixSort.verify): every rank's slice in order and stable,
        # the multiset unchanged (checksums all-reduced), the last key of rank r <= the first key of rank r + 1
        sorter.verify(bufs[k - 1], d, sum_before)
        del bufs
        torch.cuda.empty_cache()
        ms = el.item() * 1e3 / args.steps
        total = n * world
        return {"workload": wl, "desc": desc, "type": t, "n_total": total, "n_per_gpu": n, "d": d, "gen": gen, "ms": ms,
                "gkeys_per_s": total / ms / 1e6, "algorithmic_gbps": d.key_bytes * 2 * total * d.elem_bytes / ms / 1e6,
                "branch": sorter.last_branch + ("; " + state["note"] if state["note"] else ""), "exchange": state["exchange"]}

    r = run_sharded(args.workload)
    extra = {}
    for wl in [w for w in args.extra_sharded.split(",") if w and w != args.workload]:
        try:
            x = run_sharded(wl)
            extra[wl] = {"type": x["type"], "n_keys": x["n_total"], "n_keys_per_gpu": x["n_per_gpu"], "ms_per_sort": x["ms"],
                         "gkeys_per_s": x["gkeys_per_s"], "frac_of_hbm_peak": x["algorithmic_gbps"] / (HBM_PEAK_GBPS * world),
                         "exchange_branch": x["branch"], "steps": args.steps}
        except AssertionError:
            raise  # a wrong result is never swallowed
        except Exception as e:  # noqa: BLE001
            extra[wl] = {"error": repr(e)}
    if rank == 0:
        d, t = r["d"], r["type"]
        line = {
            "metric": "Gkeys/sec (u32)" if t == "u32" else f"Gkeys/sec ({t})",
            "value": r["gkeys_per_s"], "unit": "Gkeys/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u%d" % (8 * d.key_bytes) if t[0] in "u(" else t, "data": "synthetic",
            "config": {"workload": f"{args.workload} per GPU x {world}: {r['desc']}", "n_keys": r["n_total"],
                       "n_keys_per_gpu": r["n_per_gpu"], "elem_bytes": d.elem_bytes, "passes": d.key_bytes, "radix_bits": 8,
                       "generator": r["gen"],
                       "exchange": {"first": "partition by the top digit in %d sub-ranges + all-gather(256 x u64) + exact cuts inside boundary "
                                             "buckets (device-side search, one D2H) + ONE all-to-all-v in %d batches, each range sorted while "
                                             "the next is on the links" % (args.sub_ranges, args.chunks),
                                    "one": "local sort + 256-way splitter search (1 all-reduce per digit) + ONE all-to-all-v + local sort",
                                    "per-pass": "per-pass all-gather(256 x u64) + all-to-all-v"}[r["exchange"]] +
                                   (" (RCCL)" if args.backend == "nccl" else " (gloo, host-staged: rehearsal only)"),
                       "exchange_branch": r["branch"],
                       "verified": "slices sorted and stable, multiset checksum all-reduced, rank-boundary keys in order"},
            "roofline": {"bound": "hbm", "achieved": r["algorithmic_gbps"],
                         "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                         "frac": r["algorithmic_gbps"] / (HBM_PEAK_GBPS * world),
                         "traffic": None, "note": "whole-job algorithmic bytes / wall time (exchange included)"},
        }
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
