"""Register budget of the hot kernels (no GPU needed: hipcc reports it at compile time).

The LDS bucket kernels keep KPT elements, their ranks and a read-back copy in registers under a hard cap (128 VGPRs with
1024 threads per CU).  Twice in round 3 a harmless-looking change of their loop (a second call site, a loop-carried
element array) made the compiler spill hundreds of registers and the sort 5-10x slower -- bit-exact, so no parity test
saw it.  This test compiles the 8-byte unit with -Rpass-analysis=kernel-resource-usage and bounds the spills."""
import functools, os, re, subprocess, tempfile

import pytest

from radix_sort_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@functools.lru_cache(maxsize=None)
def _resources(es):
    with tempfile.TemporaryDirectory() as d:
        cmd = [_build.hipcc()] + _build.CXXFLAGS + [f"-DRSX_ES={es}", "-Rpass-analysis=kernel-resource-usage", "-c",
                                                    os.path.join(_build.CSRC, "rsx_es.hip"), "-o", os.path.join(d, "o.o")]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
    out, name = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


@pytest.mark.parametrize("es", [8, 16])
def test_bucket_and_sweep_kernels_fit_their_registers(es):
    res = _resources(es)
    seen = 0
    for name, r in res.items():
        if "rsx_bucket16_kernel" in name or "rsx_bucket_sort_kernel" in name or "rsx_sweep_kernel" in name or "rsx_small_sort_kernel" in name:
            seen += 1
            assert r.get("VGPRs Spill", 0) <= 16, (name, r)
            assert r.get("ScratchSize [bytes/lane]", 0) <= 128, (name, r)
        if "rsx_bucket16_kernel" in name:  # 2 workgroups of 512 / 3 of 256 per CU need 4 / 3 waves per SIMD
            assert r.get("Occupancy [waves/SIMD]", 0) >= (3 if "Li256E" in name else 4), (name, r)
    assert seen >= 8, sorted(res)


@pytest.mark.parametrize("es", [8, 12, 16])
def test_gated_kernels_dispatch_without_a_scratch_penalty(es):
    """Every hybrid sort enqueues rsx_bucket16_medium_kernel behind a gate.  A kernel with ~1 KB of scratch per lane takes
    20-25 us to DISPATCH even when its gate sends it home at once (measured on MI355X: 968 bytes -> 20-25 us, 32 bytes ->
    4 us; 6 % of a 2^23-key u64 sort): its scratch stays small (one call site per inlined routine, medium_kpt_for)."""
    res = _resources(es)
    names = [n for n in res if "rsx_bucket16_medium_kernel" in n]
    assert names, sorted(res)
    for name in names:
        assert res[name].get("ScratchSize [bytes/lane]", 0) <= 256, (name, res[name])
