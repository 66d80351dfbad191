#!/bin/bash
# GPU session ZF (last of the round): register top-k up to 4096-entry rows, the whole -m gpu suite, C2 line
set -u
O=gpurun_out/r2zf; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py --workload c2 > $O/c2_bench_line.json 2> $O/c2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2zf/c2_bench_line.json").read().splitlines() if l.startswith("{")][-1])
print("c2", d["value"], d["ms_per_step"], d["recall_at_10"], d["kernel_ms_per_step"], d["cpu_baseline"]["matches_gpu_topk"])
PY
