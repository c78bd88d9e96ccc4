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


def _build_demo(tmp_path):
    from radix_sort_amd import _build
    lib = _build.build()
    exe = str(tmp_path / "bench_demo")
    # plain g++ against the HIP runtime API (the macro only tells the HIP headers which platform they are on)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "radix_sort_amd", "cxx", "bench_demo.cpp"), lib,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_bench_demo_compiles(tmp_path):
    """CPU: the reference's bench protocol (main.rs:101-127) over the C++ mirror builds and links."""
    assert os.path.exists(_build_demo(tmp_path))


@pytest.mark.gpu
def test_bench_demo_protocol_and_dataset_files(tmp_path):
    """GPU: two rungs of the size ladder for both pair types (host drop-in + device-resident, output
    checked), then the raw dataset round trip of main.rs:47-99 (headerless native-endian file)."""
    exe = _build_demo(tmp_path)
    out = subprocess.run([exe, "--sizes", "0.05,0.1", "--runs", "2", "--device", "--check"], capture_output=True,
                         text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    assert out.stdout.count("GB of data in: ") == 4 and "TYPE: u32/u32 RUNS: 2" in out.stdout and "TYPE: u64/u64 RUNS: 2" in out.stdout
    f = str(tmp_path / "pairs_u64.bin")
    assert subprocess.run([exe, "--gen-data", "0.02", "u64", f], timeout=300).returncode == 0
    assert os.path.getsize(f) == int(0.02 * 1e9 / 16) * 16
    out = subprocess.run([exe, "--data", "u64", f, f], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Sorted 2 file(s)" in out.stdout, out.stdout + out.stderr
