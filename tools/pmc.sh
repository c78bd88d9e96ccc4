#!/bin/bash
# PMC counter passes for the sweep kernel (separate rocprofv3 runs, no tracing domains mixed in).
# usage: tools/pmc.sh <tag> <workload>
tag=$1; wl=${2:-c2-256m-u32}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 tools/perf.py $wl > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-60:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    if "sweep" in k or "hist" in k:
        print(k)
        for c, v in acc[k].items():
            print(f"   {c:28s} total {v:.4g}  per-dispatch {v / cnt[(k, c)]:.4g}  (dispatches {cnt[(k, c)]})")
PY
done
