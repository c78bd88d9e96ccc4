#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/step2; mkdir -p $o
bash tools/ab.sh "main h1 h2 h1b4 h2b4 h4b4 h1b16" "c2-256m-u32" 2>&1 | tee $o/ab_hist.log
RSX_LIBRARY=radix_sort_amd/lib/v/stamps.so python tools/stamps.py u32 2>&1 | tee $o/stamps_u32.log
