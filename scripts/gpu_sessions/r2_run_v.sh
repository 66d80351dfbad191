#!/bin/bash
# GPU session V: final validation of the round — whole -m gpu suite, smoke(), default bench line
set -u
O=gpurun_out/r2v; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py > $O/default_bench_line.json 2> $O/default_bench.err ) 2> $O/default_bench.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2v/default_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "recall_at_10")}, d["kernel_ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["matches_gpu_topk"])
PY
cat $O/default_bench.time | tr '\n' ' '
