"""GPU: the C++ host mirror (radix_sort_amd/cxx/radix_sort.hpp) over the C-ABI, on the reference's
own test shapes (src/radix_sort/tests.rs: 1e6 elements per built-in type, vs the std sort)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from radix_sort_amd import _build
    lib = _build.build()
    exe = str(tmp_path / "cxx_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cxx_mirror_test.cpp"),
                           lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_mirror_compiles_and_links(tmp_path):
    """CPU: header + test program compile against include/rsx.h and link with librsx.so."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cxx_mirror_reference_test_shapes(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "ALL OK" in out.stdout
