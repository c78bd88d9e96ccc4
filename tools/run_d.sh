mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/r3/tr16; mkdir -p $d
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 tools/dbg16b.py > $d/log.txt 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); cut -c1-200 $f | head -12
