#!/bin/bash
# Round 6, GPU session LV: how much of a query's traversal is spent ABOVE level 0?  Phase clocks (gs_prof) with the new per-level
# counters (expansions and clocks at the levels > 0: searchOneLayer with topK = 1 + setEntryPointsFromPreviousLayer), headline index.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6lv; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_PROF=1" \
  timeout 1500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 --rerank 74 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench.err | cut -c1-420 | awk '!seen[$0]++' | tee -a $O/summary.txt
