#!/bin/bash
# first GPU contact of the round: smoke, quick perf, chain-length probe, a slice of the parity tests
cd $GRAFT_REPO_ROOT
o=gpurun_out/step1; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || { tail -30 $o/smoke.log; exit 1; }
tail -2 $o/smoke.log
RSX_VERBOSE=1 python tools/perf.py c2-256m-u32 2> $o/perf_verbose.err | tee $o/perf_c2.log
grep -m6 "self-test\|sweep ES" $o/perf_verbose.err
python tools/perf.py c2-256m-u32 target-1b-u32 c3-1b-u64 zipf-256m-u32 step16-256m-u32 c5-slice-128m-pairs-zipf zipf-256m-u64 sorted-256m-u32 2>&1 | tee $o/perf.log
python tools/regions_probe.py c2-256m-u32 2>&1 | tee $o/regions.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge_sizes or golden or alternative or self_tests or one_context or unreserved or contention or options" 2>&1 | tail -15 | tee $o/pytest_subset.log
