#!/bin/bash
# Round 5, GPU session B: UBR — the register-table bound form (prebuilt tables, bpermute look-ups, survivors compacted, eight lanes each,
# trimmed candidate tier): parity on hardware, then the headline with the form off (the line) and on (sweep: trim period, phase clocks)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_ref_native_gpu.py -m gpu -q -x > $O/pytest_ubr.log 2>&1; echo "pytest ubr rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_ubr.log | tee -a $O/summary.txt
U="JVECTOR_HIP_GS_UBR=1"
JVECTOR_BENCH_ENV_SWEEP="$U;$U,JVECTOR_HIP_GS_UBR_TRIM=8;$U,JVECTOR_HIP_GS_UBR_TRIM=48;$U,JVECTOR_HIP_GS_PROF=1,JVECTOR_HIP_GRAPH_TIMING=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
tail -c 600 $O/bench_sweep.json | head -c 400 | tee -a $O/summary.txt
