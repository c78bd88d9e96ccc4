mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_byte or odd_element" 2>&1 | tail -n 25
