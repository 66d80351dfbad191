#!/bin/bash
# Round 4, GPU session D: workgroup form with DPP wave reductions / readlane broadcasts / unrolled table build: parity on hardware,
# lean control wave (gx_control: 32-bit score-word scans and DPP reductions); sweep + phase clocks
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py -m gpu -q -x -k "workgroup_form" > $O/pytest_wgx.log 2>&1; echo "pytest wgx rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_wgx.log | tee -a $O/summary.txt
W="JVECTOR_HIP_GS_WGX=1"
JVECTOR_BENCH_ENV_SWEEP="$W;$W,JVECTOR_HIP_GS_WGX_WAVES=4;$W,JVECTOR_HIP_GS_WGX_DEPTH=0;$W,JVECTOR_HIP_GS_WGX_SLOTS=32;$W,JVECTOR_HIP_GS_PROF=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
tail -c 1500 $O/bench_sweep.json | cut -c1-1500 >> $O/summary.txt
