"""The on-device input generators (the shapes of the reference's src/distr.rs) against their CPU
restatement in oracle/ (orc_generate): CPU tests check the restatement's distributions, the GPU test that
rsx_generate_device gives the same bytes for the same (seed, index)."""
import numpy as np
import pytest

import util


def _lay(orc, t):
    return orc.Layout(*util.TYPES[t])


def test_restated_generators_have_the_reference_shapes(orc):
    n = 400000
    u32 = _lay(orc, "u32")
    # KeyUniform (distr.rs:40-52): every byte of the key uniform
    a = orc.generate(n, u32, orc.GEN_UNIFORM, 1).view("<u4")
    assert abs(a.astype(np.float64).mean() / 2 ** 32 - 0.5) < 0.01
    # Zipf, exponent 1 (distr.rs:54-76): log-uniform over [0, 2^32): each octave equally likely
    z = orc.generate(n, u32, orc.GEN_ZIPF, 2, 1.0).view("<u4").astype(np.float64) + 1
    octave = np.floor(np.log2(z)).astype(int)
    share = np.bincount(octave, minlength=32) / n
    assert np.all(np.abs(share - 1 / 32) < 0.004), share
    # step-uniform (distr.rs:78-106): n equally spaced values s, 2s, .., n s with s = MAX / (n + 1)
    st = orc.generate(n, u32, orc.GEN_STEP, 3, 16.0).view("<u4")
    s = (2 ** 32 - 1) // 17
    assert set(np.unique(st).tolist()) == {s * (i + 1) for i in range(16)}
    # geometric (distr.rs:3-38 MyExp): failures before the first success, mean (1 - p) / p
    for p in (0.5, 0.01, 0.001):
        g = orc.generate(n, u32, orc.GEN_GEOMETRIC, 4, p).view("<u4").astype(np.float64)
        assert abs(g.mean() - (1 - p) / p) < 0.02 * (1 - p) / p + 0.01, (p, g.mean())
        assert abs((g == 0).mean() - p) < 0.01
    # (key, 0) pairs (distr.rs:22-26,42-52) and (key, index) pairs
    pair = _lay(orc, "(u64,u64)")
    kz = orc.generate(1000, pair, orc.GEN_UNIFORM, 5, index_base=7000, payload_zero=True).view("<u8").reshape(-1, 2)
    ki = orc.generate(1000, pair, orc.GEN_UNIFORM, 5, index_base=7000).view("<u8").reshape(-1, 2)
    assert (kz[:, 1] == 0).all() and (ki[:, 1] == np.arange(7000, 8000)).all() and (kz[:, 0] == ki[:, 0]).all()
    # sorted / reversed / constant
    assert (orc.generate(100, u32, orc.GEN_SORTED, 0, index_base=5).view("<u4") == np.arange(5, 105)).all()
    assert (orc.generate(100, u32, orc.GEN_REVERSED, 0).view("<u4") == np.arange(99, -1, -1)).all()
    assert (orc.generate(100, u32, orc.GEN_CONSTANT, 0, 77.0).view("<u4") == 77).all()


GENS = [("GEN_UNIFORM", 0.0), ("GEN_ZIPF", 1.0), ("GEN_STEP", 16.0), ("GEN_SORTED", 0.0), ("GEN_REVERSED", 0.0),
        ("GEN_CONSTANT", 123456.0), ("GEN_GEOMETRIC", 0.001), ("GEN_GEOMETRIC", 0.37)]


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["u32", "u64", "(u64,u64)", "(u32,u32)", "u128", "u16", "(pay32+u32)"])
@pytest.mark.parametrize("gen,param", GENS)
def test_device_generators_match_the_restatement(orc, t, gen, param):
    import torch
    import radix_sort_amd as rs
    ctx = rs.default_context(torch.cuda.current_device())
    d = rs.RadixDigits(*util.TYPES[t])
    lay = _lay(orc, t)
    n, base = 300007, 5_000_000_000  # index_base beyond 2^32: 64-bit indices
    for pz in (False, True):
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda")
        g = getattr(rs, gen) | (rs.GEN_PAYLOAD_ZERO if pz else 0)
        ctx.generate_device(x.data_ptr(), n, d, g, 0x5EED0000 + 17, param, base)
        want = orc.generate(n, lay, getattr(orc, gen), 0x5EED0000 + 17, param, base, payload_zero=pz)
        assert np.array_equal(x.cpu().numpy(), want), (t, gen, param, pz)
