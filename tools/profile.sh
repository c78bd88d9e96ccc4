#!/bin/bash
# Runs bench.py under rocprofv3 (kernel trace + stats) on the GPU box and leaves the
# summaries under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
# usage: tools/profile.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --extra '' --no-cpu-baseline "$@" > $out/bench.json 2> $out/stderr.log
find $out -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cat $out/bench.json
head -20 $out/kernel_stats.csv
