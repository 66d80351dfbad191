#!/bin/bash
# Round 6, GPU session N: the rerank with the lists' remainders packed several queries per wavefront: parity of every exact-score path,
# then the headline (rerankK 74 -> 64 + 10 rows: six queries' remainders per wavefront) with and without the packing.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6n; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_nvq_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_zz_sharded_graph_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_EXACT_NO_PACK=1" \
  timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|evaluate" $O/bench.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6n")
l=[json.loads(x) for x in open(os.path.join(d,"bench.json")).read().strip().splitlines() if x.startswith("{")][-1]
print("DEFAULT (packed)", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), l.get("rerank_roofline_frac"))
PY
