#!/bin/bash
# GPU session T2: default bench after the finer rerankK ladder, the rocprofv3 recipe over it, whole -m gpu suite, smoke
set -u
O=gpurun_out/r2t2; mkdir -p $O
( time timeout 1200 python bench.py > $O/default_bench_line.json 2> $O/default_bench.err ) 2> $O/default_bench.time
grep -E "calibrate.*graph|evaluate" $O/default_bench.err | tail -6
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2t2/default_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "recall_at_10", "recall_se")}, d["config"]["rerankK"], d["kernel_ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
timeout 2400 bash scripts/profile_r2.sh r2_10m_v7 > $O/profile.log 2>&1
tail -2 $O/profile.log | cut -c1-200
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
