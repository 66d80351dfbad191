#!/bin/bash
# Round 4, GPU session P: UB8 — the pair-lane kernel with an 8-bit upper-bound table per wave (fresh neighbours that provably cannot be
# popped skip their exact score): parity on hardware, then the headline sweep (3 / 4 waves per CU) with phase clocks
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py -m gpu -q -x -k "upper_bound" > $O/pytest_ub8.log 2>&1; echo "pytest ub8 rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_ub8.log | tee -a $O/summary.txt
U="JVECTOR_HIP_GS_UB8=1"
JVECTOR_BENCH_ENV_SWEEP="$U;$U,JVECTOR_HIP_GS_UB8_PER_CU=3;$U,JVECTOR_HIP_GS_UB8_PER_CU=5;$U,JVECTOR_HIP_GS_CAND_CAP=128;$U,JVECTOR_HIP_GS_PROF=1,JVECTOR_HIP_GRAPH_TIMING=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-330 | awk '!seen[$0]++' | tee -a $O/summary.txt
