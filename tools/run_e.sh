mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 100 python tools/dbg16b.py 2>&1 | grep "2\^"
timeout -k 5 100 python tools/dbg16b.py 2>&1 | grep "2\^28"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -n 4
