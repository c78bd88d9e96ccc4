cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
W="c3-1b-u64 zipf-256m-u64 pairs-256m-u32u32"
AB_ROUNDS=2 bash tools/ab.sh "base w1024k6" "$W" > gpurun_out/r3/ab_w1024a.txt 2>&1
MAXR=8 AB_ROUNDS=2 bash tools/ab.sh "base w1024k7" "$W" > gpurun_out/r3/ab_w1024b.txt 2>&1
cat gpurun_out/r3/ab_w1024a.txt; echo "---- MAXR=8"; cat gpurun_out/r3/ab_w1024b.txt
