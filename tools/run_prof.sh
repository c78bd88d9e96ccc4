cd $GRAFT_REPO_ROOT
bash tools/profile_all.sh r03 "c3-1b-u64 target-1b-u32 c2-256m-u32 zipf-256m-u32 step16-256m-u32 zipf-256m-u64 c5-slice-128m-pairs-zipf u16-256m u8-256m c1-1m-u32" "c3-1b-u64 c2-256m-u32 zipf-256m-u32 u16-256m u8-256m" > gpurun_out/prof_r03.log 2>&1
tail -n 5 gpurun_out/prof_r03.log
