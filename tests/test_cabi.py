"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/rsx.h
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from radix_sort_amd import _build, _lib
    _build.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from radix_sort_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 14
    assert sorted(_lib.SYMBOLS) == syms, "ctypes binding list out of sync with include/rsx.h"
    for s in syms:
        assert hasattr(lib, s), f"librsx.so does not export {s}"


def test_version_and_strerror(lib):
    assert lib.rsx_version() == 100
    assert lib.rsx_strerror(0) == b"ok"
    for code in range(-7, 0):
        assert lib.rsx_strerror(code) not in (b"", b"unknown status")


def test_code_object_is_gfx950():
    from radix_sort_amd import _build
    blob = open(_build.LIB, "rb").read()
    assert b"gfx950" in blob
    assert b"rsx_sweep_kernel" in blob and b"rsx_hist_kernel" in blob


def test_layout_struct_matches_header():
    from radix_sort_amd._lib import Layout
    assert ctypes.sizeof(Layout) == 16
    assert [f[0] for f in Layout._fields_] == ["elem_bytes", "key_offset", "key_bytes", "key_kind"]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly(lib):
    """Without a device the product path must raise, never fall back to a CPU sort."""
    import radix_sort_amd as rs
    h = ctypes.c_void_p()
    assert lib.rsx_ctx_create(-1, ctypes.byref(h)) == -5  # RSX_ERR_NODEVICE
    with pytest.raises(rs.RsxError):
        rs.radix_sort(np.arange(10, dtype=np.uint32)[::-1].copy())


def test_null_ctx_is_an_error_not_a_crash(lib):
    lay = __import__("radix_sort_amd")._lib.Layout(4, 0, 4, 0)
    assert lib.rsx_sort_device(None, None, None, 10, ctypes.byref(lay), None) == -1
    assert lib.rsx_ctx_destroy(None) == -1
    assert lib.rsx_last_error(None) == b"null context"


def test_product_does_not_touch_the_oracle():
    """The package may not import/link/execute anything under oracle/ (or any CPU sort)."""
    pkg = os.path.join(ROOT, "radix_sort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f
                assert "liborc" not in src, f
