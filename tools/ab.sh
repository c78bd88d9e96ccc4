#!/bin/bash
# A/B timing of library variants on ONE box: tools/ab.sh "variant[@RSX_DEBUG] ..." "workload ..."
# (variants: radix_sort_amd/lib/v/NAME.so; two interleaved rounds to expose drift)
for round in 1 2; do
for vv in $1; do
  v=${vv%%@*}; dbg=0; [[ "$vv" == *@* ]] && dbg=${vv#*@}
  echo "== $vv (round $round)"
  RSX_DEBUG=$dbg RSX_LIBRARY=$(dirname $0)/../radix_sort_amd/lib/v/$v.so python tools/perf.py $2 2>&1 | tail -n $(echo $2 | wc -w)
done
done
