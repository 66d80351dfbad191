#!/bin/bash
set -u
O=gpurun_out/r2t3; mkdir -p $O
( time timeout 1200 python bench.py > $O/default_bench_line.json 2> $O/default_bench.err ) 2> $O/default_bench.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2t3/default_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "recall_at_10", "recall_se")}, d["config"]["rerankK"], d["kernel_ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["l2_gather"]["frac"], d["rerank"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["scalar_value"], d["cpu_baseline"]["matches_gpu_topk"], d.get("avg_expanded"), d.get("avg_visited"), d.get("flat_mode", {}).get("value"))
PY
cat $O/default_bench.time | tr '\n' ' '
