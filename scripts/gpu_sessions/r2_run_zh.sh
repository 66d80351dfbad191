#!/bin/bash
# GPU session ZH: does a 131 072-query step amortise the persistent launch's tail further?  (10M, default otherwise)
set -u
O=gpurun_out/r2zh; mkdir -p $O
timeout 900 python bench.py --queries 131072 --steps 6 --warmup 1 --no-flat --no-cpu-baseline > $O/q131k.json 2> $O/q131k.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2zh/q131k.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["recall_at_10"], d["config"]["rerankK"], d["kernel_ms_per_step"])
PY
