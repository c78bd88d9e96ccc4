mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "u32 16" "u32 20" "u64 20" "u32 22"; do set -- $cfg; d=gpurun_out/r3/tr_$1_$2; mkdir -p $d
  rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tools/mid_trace.py $1 $2 > $d/log.txt 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1); echo "== $cfg"; python3 tools/trace_summary.py $f | tail -n 14; done > gpurun_out/r3/mid_trace.txt 2>&1
cat gpurun_out/r3/mid_trace.txt
bash tools/profile_all.sh r03n "u16-256m u8-256m" "u16-256m u8-256m" > gpurun_out/r3/prof_narrow.log 2>&1; tail -n 30 gpurun_out/r3/prof_narrow.log
