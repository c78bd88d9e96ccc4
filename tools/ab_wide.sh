#!/bin/bash
# A/B of library variants over the wide-key sizes and the middle sizes on ONE box:
#   tools/ab_wide.sh "variant ..."   (variants: radix_sort_amd/lib/v/NAME.so)
root=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do
for v in $1; do
  echo "== $v (round $round)"
  RSX_LIBRARY=$root/radix_sort_amd/lib/v/$v.so MODES=1 LGS=${LGS:-23,24,25,26,27,28,29,30} python $root/tools/wide_probe.py "u64" > /tmp/abw.log 2>&1 || { cat /tmp/abw.log; exit 1; }
  cat /tmp/abw.log
  if [ $round = 1 ]; then
    RSX_LIBRARY=$root/radix_sort_amd/lib/v/$v.so MODES=1 LGS=${LGS2:-22,24,26,28} python $root/tools/wide_probe.py "(u64,u64)" "u128" > /tmp/abw.log 2>&1 || { cat /tmp/abw.log; exit 1; }
    cat /tmp/abw.log
    RSX_LIBRARY=$root/radix_sort_amd/lib/v/$v.so KEYS="u64;(u64,u64);u128" python $root/tools/mid_sweep.py > /tmp/abw.log 2>&1 || { cat /tmp/abw.log; exit 1; }
    cat /tmp/abw.log
  fi
done
done
