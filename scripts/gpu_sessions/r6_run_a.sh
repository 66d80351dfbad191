#!/bin/bash
# Round 6, GPU session A: how does the headline traversal scale with the number of resident waves per CU?  VERDICT r5 #1 asks for 16
# waves per CU (<= 128 VGPRs); before building that, measure the slope on the existing kernel: 2 / 4 / 6 / 8 workers per CU, same LDS
# sizing (gs_waves_per_cu only caps the worker count), with the phase clocks at 4 and 8.  If 6 -> 8 already bends, 8 -> 12/16 pays little.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_WAVES_PER_CU=2;JVECTOR_HIP_GS_WAVES_PER_CU=4;JVECTOR_HIP_GS_WAVES_PER_CU=6;JVECTOR_HIP_GS_WAVES_PER_CU=7;JVECTOR_HIP_GS_WAVES_PER_CU=8;JVECTOR_HIP_GS_WAVES_PER_CU=4,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 1500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-600 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6a")
l=json.loads(open(os.path.join(d,"bench_sweep.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"))
PY
