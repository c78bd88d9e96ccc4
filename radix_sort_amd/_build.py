"""Builds librsx.so (the C-ABI HIP library) in-tree for gfx950.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU; the resulting
radix_sort_amd/lib/librsx.so is git-ignored but travels to the GPU box.

The library is nine translation units: rsx.hip (the C-ABI and the size-independent kernels) and
rsx_es.hip once per element size (-DRSX_ES=1,2,4,8,12,16,24,32), compiled in parallel and linked
into one shared object.  `python -m radix_sort_amd._build --variant NAME --es 4 -DX=1 ...` relinks a
tuning variant (lib/v/NAME.so) that recompiles only the named size with the extra flags.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "librsx.so")
ELEM_SIZES = (1, 2, 4, 8, 12, 16, 24, 32)
DEPS = ["rsx.hip", "rsx_es.hip", "rsx_device.hpp", "rsx_internal.hpp", "rsx_launch_impl.hpp", "rsx_misc_kernels.hpp", "rsx_small_kernel.hpp", "rsx_mid_kernels.hpp",
        os.path.join("..", "..", "include", "rsx.h")]
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build librsx.so)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def _compile(job):
    src, obj, flags, verbose = job
    cmd = [hipcc()] + CXXFLAGS + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return obj


def _objects(extra, only_es, objdir, verbose, reuse_from=None):
    """Compiles the translation units (in parallel); returns the object list.  With `only_es`, the
    other sizes' objects are taken from `reuse_from` (the main build) instead of being recompiled."""
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    host = os.path.join(objdir, "rsx.o")
    if only_es is None:
        jobs.append(("rsx.hip", host, list(extra), verbose))
        objs.append(host)
    else:
        objs.append(os.path.join(reuse_from, "rsx.o"))
    for es in ELEM_SIZES:
        if only_es is None or es in only_es:
            obj = os.path.join(objdir, f"rsx_es{es}.o")
            jobs.append(("rsx_es.hip", obj, [f"-DRSX_ES={es}"] + list(extra), verbose))
        else:
            obj = os.path.join(reuse_from, f"rsx_es{es}.o")
        objs.append(obj)
    workers = min(len(jobs), max(1, (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(_compile, jobs))
    return objs


def _link(objs, out, verbose):
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("RSX_CXXFLAGS", "").split()
    objs = _objects(extra, None, OBJDIR, verbose)
    _link(objs, LIB, verbose)
    return LIB


def build_variant(name: str, only_es, flags, verbose: bool = False) -> str:
    """lib/v/NAME.so: the main build's objects with the named element sizes recompiled under `flags`
    (load it with RSX_LIBRARY=...).  Flags that the host unit reads too (tile geometry: -DRSX_KPT*, -DRSX_WG*)
    need `--es all`, which recompiles every unit."""
    if stale():
        build(force=True, verbose=verbose)
    vdir = os.path.join(LIBDIR, "v")
    os.makedirs(vdir, exist_ok=True)
    objs = _objects(flags, None if only_es is None else tuple(only_es), os.path.join(vdir, "obj_" + name), verbose,
                    reuse_from=OBJDIR)
    out = os.path.join(vdir, name + ".so")
    _link(objs, out, verbose)
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--variant" in argv:
        i = argv.index("--variant")
        name = argv[i + 1]
        rest = argv[:i] + argv[i + 2:]
        es = [4]
        if "--es" in rest:
            j = rest.index("--es")
            es = None if rest[j + 1] == "all" else [int(x) for x in rest[j + 1].split(",")]  # all: host unit too
            rest = rest[:j] + rest[j + 2:]
        print(build_variant(name, es, rest, verbose=True))
    else:
        print(build(force=True, verbose=True))
