#!/bin/bash
# Builds librsx.so and prints register/LDS use of the sweep kernels. usage: tools/build.sh [filter-regex]
# RSX_CXXFLAGS: extra -D knobs; RSX_OUT: output path (variants for A/B runs, loaded with RSX_LIBRARY=...)
cd /root/repo/radix_sort_amd/csrc || exit 1
mkdir -p ../lib
filt=${1:-"sweep_kernelILi(4|8|16)E.*EjLb0ELb1"}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $RSX_CXXFLAGS -o ${RSX_OUT:-../lib/librsx.so} rsx.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy" | grep -E -A3 "error|$filt" | sed -e 's/.*remark: //' -e 's/\[-Rpass.*//'
