#!/bin/bash
# Round 4, GPU session F: workgroup form, full table vs partial tables with 2-3 workgroups per CU (LDS tier of the visited set
# reserved this time), phase clocks of the 2-per-CU form
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py -m gpu -q -x -k "workgroup_form" > $O/pytest_wgx.log 2>&1; echo "pytest wgx rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_wgx.log | tee -a $O/summary.txt
W="JVECTOR_HIP_GS_WGX=1,JVECTOR_HIP_GRAPH_TIMING=1"
JVECTOR_BENCH_ENV_SWEEP="$W;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=3;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4,JVECTOR_HIP_GS_WGX_LUT_M=48;$W,JVECTOR_HIP_GS_WGX_PER_CU=3,JVECTOR_HIP_GS_WGX_WAVES=4;$W,JVECTOR_HIP_GS_WGX_PER_CU=3,JVECTOR_HIP_GS_WGX_WAVES=3;$W,JVECTOR_HIP_GS_WGX_PER_CU=4,JVECTOR_HIP_GS_WGX_WAVES=3;$W,JVECTOR_HIP_GS_WGX_PER_CU=2,JVECTOR_HIP_GS_WGX_WAVES=4,JVECTOR_HIP_GS_PROF=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-330 | awk '!seen[$0]++' | tee -a $O/summary.txt
