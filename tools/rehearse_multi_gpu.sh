#!/bin/bash
# Rehearses bench.py's N > 1 path on a box with ONE GPU: R ranks share the device, collectives go through
# gloo (staged through the host), every --exchange schedule, the cross-rank verification included.  Rates are
# meaningless (host-staged); what is exercised is the code path the driver runs with RCCL on a multi-GPU node.
# usage: tools/rehearse_multi_gpu.sh [ranks=2] [workload=c2-256m-u32]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=${1:-2}; wl=${2:-c2-256m-u32}
o=gpurun_out/rehearse; mkdir -p $o
for ex in first one per-pass; do
  python bench.py --gpus $R --steps 2 --warmup 1 --backend gloo --exchange $ex --workload $wl > $o/n${R}_$ex.log 2>&1
  echo "== N=$R gloo --exchange $ex: rc $?"; grep -a '^{"metric"' $o/n${R}_$ex.log | cut -c1-300 || tail -5 $o/n${R}_$ex.log
done
