#!/bin/bash
# Round 4, GPU session E: (1) the one-wave form with wave-scope sync points / readlane broadcasts / DPP reductions (the default
# line of this run vs 98.4 ms before); (2) the workgroup form with a PARTIAL table (first P subspaces in LDS, the rest table-free)
# so that 2-3 workgroups — control waves — share a CU; parity of both first
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py -m gpu -q -x > $O/pytest_trav.log 2>&1; echo "pytest traversal rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_trav.log | tee -a $O/summary.txt
W="JVECTOR_HIP_GS_WGX=1"
JVECTOR_BENCH_ENV_SWEEP="$W;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4,JVECTOR_HIP_GS_WGX_LUT_M=48;$W,JVECTOR_HIP_GS_WGX_PER_CU=3,JVECTOR_HIP_GS_WGX_WAVES=4;$W,JVECTOR_HIP_GS_WGX_PER_CU=3,JVECTOR_HIP_GS_WGX_WAVES=3;$W,JVECTOR_HIP_GS_WGX_PER_CU=4,JVECTOR_HIP_GS_WGX_WAVES=3;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4,JVECTOR_HIP_GS_PROF=1,JVECTOR_HIP_GRAPH_TIMING=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
tail -c 3000 $O/bench_sweep.json | cut -c1-3000 >> $O/summary.txt
