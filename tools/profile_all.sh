#!/bin/bash
# Round evidence: rocprofv3 kernel-trace stats (bench.py under the profiler) and PMC passes for a list of
# workloads.  usage: tools/profile_all.sh <round-tag> "<workloads for trace+traffic>" "<workloads for all PMC sets>"
# Leaves gpurun_out/profiles_<tag>/<workload>_{kernel_stats.csv,bench.json,pmc.txt}; copy into profiles/.
tag=$1; wls=$2; full=$3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$root/bench.py" ] || { echo "no bench.py under $root"; exit 1; }; out=$root/gpurun_out/profiles_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
for wl in $wls; do
  d=$out/trace_$wl; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --workload $wl --steps 8 --warmup 2 --extra '' --no-cpu-baseline > $out/${wl}_bench.json 2> $d/stderr.log
  find $d -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/${wl}_kernel_stats.csv
  echo "== $wl"; cat $out/${wl}_bench.json | cut -c1-400; head -6 $out/${wl}_kernel_stats.csv
  sets=("FETCH_SIZE" "WRITE_SIZE")
  if [[ " $full " == *" $wl "* ]]; then
    sets=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
          "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE")
  fi
  : > $out/${wl}_pmc.txt
  i=0
  for set in "${sets[@]}"; do
    i=$((i+1)); p=$out/pmc_${wl}_$i
    rocprofv3 --pmc $set --output-format csv -d $p -o p -- python3 tools/perf.py $wl > $p.log 2>&1
    f=$(find $p -name '*counter_collection.csv' | head -1)
    python3 - "$f" >> $out/${wl}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-96:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(acc):
    if any(w in k for w in ("sweep", "hist", "expand", "bucket", "tile", "count16", "total16")):
        for c, v in acc[k].items():
            print(f"{k}  {c:24s} total {v:.6g}  per-dispatch {v / cnt[(k, c)]:.6g}  dispatches {cnt[(k, c)]}")
PY
    rm -rf $p
  done
  rm -rf $d
  cat $out/${wl}_pmc.txt | grep -E "FETCH|WRITE" | head -8
done
