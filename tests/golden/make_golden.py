"""Generates tests/golden/golden.npz: small seeded (input, expected) pairs.

The reference holds no golden vectors (tests.rs uses an unseeded thread_rng) and
cannot be built here (Rust; no cargo/rustc in the image), so the vectors are
produced in this container by TWO independent implementations that must agree
bit-for-bit before a pair is written:
  (1) the C oracle oracle/rsx_oracle.c (restates mod.rs:61-176 / :183-212), and
  (2) numpy's stable argsort on the mapped key (the acceptance property of
      tests.rs:7-23,133-187).
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
import util  # noqa: E402

SMALL = [0, 1, 2, 63, 64, 65, 255, 256, 257]
MEDIUM = [1000]
MEDIUM_DISTS = ["uniform", "two", "zipf"]
MEDIUM2 = 4097  # uniform only
LARGE = 100003
LARGE_TYPES = {"u32": ["uniform"], "(u64,u64)": ["zipf"]}


def cases():
    for t in util.TYPES:
        for dist in util.DISTS:
            for n in SMALL:
                yield t, dist, n
        for n in MEDIUM:
            for dist in MEDIUM_DISTS:
                yield t, dist, n
        yield t, "uniform", MEDIUM2
    for t, dists in LARGE_TYPES.items():
        for dist in dists:
            yield t, dist, LARGE


def main():
    oracle.build()
    out = {}
    for i, (t, dist, n) in enumerate(cases()):
        es, ko, kb, kind = util.TYPES[t]
        lay = oracle.Layout(es, ko, kb, kind)
        seed = 0x5EED0000 + i
        raw = util.make_input(t, n, dist, seed)
        a = oracle.sort_parallel(raw, lay, threads=3)
        b = oracle.sort0(raw, lay)
        c = oracle.numpy_stable_sort(raw, lay)
        assert np.array_equal(a, b) and np.array_equal(a, c), (t, dist, n)
        key = f"{t}|{dist}|{n}|{seed}"
        out["in|" + key] = raw
        out["out|" + key] = a
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)//2} cases -> {path} ({os.path.getsize(path)/1e6:.2f} MB)")


if __name__ == "__main__":
    main()
