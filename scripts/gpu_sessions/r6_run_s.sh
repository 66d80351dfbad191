#!/bin/bash
# Round 6, GPU session S: the rerank fused into the traversal wave (gs_body.h gs_rr_round, option gs_fused_rerank).
# Parity first (the new test + the bound-form and traversal suites), then the headline with the option on / off.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6s; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
for f in 1 0; do
  JVECTOR_HIP_GS_FUSED_RERANK=$f timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sub-workloads --no-cpu-baseline --no-flat --index-cache /tmp/idx10m.npz > $O/bench_f$f.json 2> $O/bench_f$f.err
  echo "bench fused=$f rc=$?" | tee -a $O/summary.txt
  python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench_f$f.json").read().strip().splitlines()[-1])
print("fused=$f", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), l["config"]["rerankK"], l["recall_at_10"], l["roofline"].get("frac"), l["roofline"].get("frac_traversal_bytes_only"), l["roofline"].get("fused_rerank_rows"))
PY
done
