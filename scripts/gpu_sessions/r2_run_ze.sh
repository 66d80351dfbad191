#!/bin/bash
# GPU session ZE: flat-scan filter with workgroup-staged survivors — parity, fuzz, C2 line, C3 flat mode
set -u
O=gpurun_out/r2ze; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_sharded_cabi.py -x -q -m gpu -k "flat or filtered or c2_full or sharded" > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python scripts/fuzz_kernels.py 45 9 2>&1 | grep -v amdgpu | tail -3
timeout 600 python bench.py --workload c2 > $O/c2_bench_line.json 2> $O/c2.err
timeout 900 python bench.py --mode flat --steps 5 --no-cpu-baseline > $O/c3_flat.json 2> $O/c3_flat.err
python - <<'PY'
import json
for f in ("c2_bench_line", "c3_flat"):
    d = json.loads([l for l in open(f"gpurun_out/r2ze/{f}.json").read().splitlines() if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["recall_at_10"], d["kernel_ms_per_step"], (d.get("cpu_baseline") or {}).get("matches_gpu_topk"))
PY
