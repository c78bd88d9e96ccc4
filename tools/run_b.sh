mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/mid_sweep.py > gpurun_out/r3/mid5.txt 2>&1; cat gpurun_out/r3/mid5.txt
for cfg in "u32 16" "u32 20" "u64 20"; do set -- $cfg; d=gpurun_out/r3/trb_$1_$2; mkdir -p $d
  rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tools/mid_trace.py $1 $2 > $d/log.txt 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1); echo "== $cfg"; python3 tools/trace_summary.py $f | tail -n 5; done > gpurun_out/r3/mid_trace2.txt 2>&1
cat gpurun_out/r3/mid_trace2.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "middle or alternative or around or 1e6 or capture or two_streams or distributions" > gpurun_out/r3/pytest_mid4.log 2>&1; tail -n 5 gpurun_out/r3/pytest_mid4.log
