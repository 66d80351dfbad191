#!/bin/bash
# Round 5, GPU session F: UBR scoring with all 24 codebook rows of a lane requested at once (the compiler parks 16 table registers in scratch around the pass)
# (they were fetched pair by pair), the bound's cross-lane reads issued eight look-ups at a time + v_perm byte select, no scratch in the loop
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py -m gpu -q -x > $O/pytest_ubr.log 2>&1; echo "pytest ubr rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_ubr.log | tee -a $O/summary.txt
U="JVECTOR_HIP_GS_UBR=1"
JVECTOR_BENCH_ENV_SWEEP="$U;$U,JVECTOR_HIP_GS_PROF=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
