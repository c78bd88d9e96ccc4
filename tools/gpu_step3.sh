#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/step3; mkdir -p $o
for ex in first; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --exchange $ex --workload c2-256m-u32 > $o/n2_$ex.log 2>&1
  echo "== N=2 gloo $ex: rc $?"; grep -a '^{"metric"' $o/n2_$ex.log | cut -c1-420 || tail -5 $o/n2_$ex.log
done



