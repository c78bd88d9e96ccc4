#!/bin/bash
# The reference's bench protocol (main.rs:101-127: size ladder, both pair types) through the C++ mirror:
# tools/bench_demo_ladder.sh > profiles/rNN_bench_demo_ladder.txt   (on the GPU box; builds bench_demo with g++)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$root/radix_sort_amd/lib/librsx.so
exe=/tmp/bench_demo_$$
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $exe $root/radix_sort_amd/cxx/bench_demo.cpp $lib -L/opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,$(dirname $lib) -Wl,-rpath,/opt/rocm/lib || exit 1
$exe --sizes 0.5,1.0,2.0,4.0 --runs 3 --device --check
rm -f $exe
