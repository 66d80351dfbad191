#!/bin/bash
# Round 6, GPU session T: where do the fused rerank's 4 ms go?  Phase clocks of the traversal (gs_prof) with the rerank inside the
# wave and outside it, same index, same process.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6t; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_FUSED_RERANK=1,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_FUSED_RERANK=0,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_FUSED_RERANK=1;JVECTOR_HIP_GS_FUSED_RERANK=0" \
  timeout 1500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench.err | cut -c1-420 | awk '!seen[$0]++' | tee -a $O/summary.txt
