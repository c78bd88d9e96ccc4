cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
W="u64-128m u64-256m u64-512m c3-1b-u64 f64-128m"
for r in 1 2; do for m in 0 8; do echo "== MAXR=$m round $r"; MAXR=$m python tools/perf.py $W 2>&1 | grep -v amdgpu; done; done > gpurun_out/r3/ab_regions8.txt 2>&1
cat gpurun_out/r3/ab_regions8.txt
