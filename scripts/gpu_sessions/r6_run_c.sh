#!/bin/bash
# Round 6, GPU session C: (1) the rerank gather with its candidate list split evenly over the wavefronts and an rw-row tile (14 instead
# of 9 waves per CU); parity of every exact-score path; (2) the headline again; (3) A/B: the level descriptor pinned in SGPRs (188 bytes
# of scratch: does removing the kernarg re-reads pay for the spills?)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_ubr_gpu.py tests/test_zz_sharded_graph_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
JVECTOR_HIP_LIBRARY=$R/build/variants/libjvector_hip_pin.so timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_pin.json 2> $O/bench_pin.err
echo "bench pin rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6c")
for f in ("bench.json","bench_pin.json"):
    try:
        l=json.loads(open(os.path.join(d,f)).read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), l.get("recall_at_10"), l.get("config",{}).get("rerankK"))
    except Exception as e:
        print(f, "failed", e)
PY
