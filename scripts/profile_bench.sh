#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# usage: scripts/profile_bench.sh <tag> [extra bench args]
# Writes small summaries to gpurun_out/prof_<tag>/ (the big traces stay in /tmp on the box).
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/prof_$TAG; K=$R/gpurun_out/prof_$TAG
mkdir -p $O $K
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $K/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/$C -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-flat "$@" > $K/$C.log 2>&1
done
cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { head -1 $f > $K/${C}_jv.csv; grep -E "jv::" $f >> $K/${C}_jv.csv; }
done
ls -la $K
