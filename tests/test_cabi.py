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
    assert lib.rsx_version() == 200
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


def test_options_validate_without_a_device(lib):
    """rsx_ctx_set_option / rsx_ctx_get_info reject null contexts (no device needed for that)."""
    out = ctypes.c_uint64(0)
    assert lib.rsx_ctx_set_option(None, 1, 0) == -1
    assert lib.rsx_ctx_get_info(None, 1, ctypes.byref(out)) == -1


def test_no_stray_environment_switches():
    """The production library reads no tuning switches from the environment: the ablation
    bits exist only under -DRSX_TUNING, geometry knobs only through rsx_ctx_set_option."""
    src = open(os.path.join(ROOT, "radix_sort_amd", "csrc", "rsx.hip")).read()
    envs = set(re.findall(r'getenv\("([A-Z_]+)"\)', src))
    assert envs <= {"RSX_VERBOSE", "RSX_DEBUG"}, envs
    # RSX_DEBUG only inside the RSX_TUNING block
    i = src.index('getenv("RSX_DEBUG")')
    assert "#ifdef RSX_TUNING" in src[max(0, i - 300):i]
    impl = open(os.path.join(ROOT, "radix_sort_amd", "csrc", "rsx_launch_impl.hpp")).read()
    for m in re.finditer(r'getenv\("([A-Z_]+)"\)', impl):
        assert "#ifdef RSX_TUNING" in impl[max(0, m.start() - 200):m.start()], m.group(1)


def test_product_does_not_touch_the_oracle():
    """The package may not import/link/execute anything under oracle/ (or any CPU sort)."""
    pkg = os.path.join(ROOT, "radix_sort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f
                assert "liborc" not in src, f
