#!/bin/bash
# GPU session R: top-k short-row kernel (parity + effect on the 1M step)
set -u
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_graph_search.py tests/test_zz_device_traversal_gpu.py -x -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -5
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_TOPK_RADIX=1;JVECTOR_HIP_GS_TIE_CHECK=1" timeout 900 python bench.py --n 1000000 --steps 4 --warmup 1 --no-flat --no-cpu-baseline > $O/bench.out 2> $O/bench.err
grep -E "sweep" $O/bench.err | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2r/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "recall_at_10", "matches_gpu_topk") if k in d}, d.get("kernel_ms_per_step"))
PY
