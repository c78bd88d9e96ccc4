#!/bin/bash
# A/B timing of library variants on ONE box: tools/ab.sh "variant[@RSX_DEBUG] ..." "workload ..."
# (variants: radix_sort_amd/lib/v/NAME.so; two interleaved rounds to expose drift)
for round in $(seq 1 ${AB_ROUNDS:-2}); do
for vv in $1; do
  v=${vv%%@*}; dbg=0; [[ "$vv" == *@* ]] && dbg=${vv#*@}
  echo "== $vv (round $round)"
  RSX_DEBUG=$dbg RSX_LIBRARY=$(dirname $0)/../radix_sort_amd/lib/v/$v.so python tools/perf.py $2 > /tmp/ab_one.log 2>&1
  tail -n $(echo $2 | wc -w) /tmp/ab_one.log
  if grep -q "Memory access fault\|core dump" /tmp/ab_one.log; then echo "GPU FAULT in $vv: stopping"; exit 1; fi
done
done
