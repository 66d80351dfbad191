#!/bin/bash
# Round 6, GPU session E: finer phase clocks of the register-table bound form (wait for the row / scoring rounds / owner sum + finish,
# passes per expansion, survivor histogram) at 8 and at 4 waves per CU — where does the chain serialise now?
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_WAVES_PER_CU=4,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 1500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
