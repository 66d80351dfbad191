#!/bin/bash
# Round 4, GPU session Y: closing validation of the final library (after the robust prune kernel lost a third of its instructions) — the whole -m gpu suite, smoke, the default bench command, then a kernel trace of one build + search run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4y; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-120 | tee -a $O/summary.txt
( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>> $O/summary.txt; echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sub-run|batch sweep|evaluate" $O/bench_default.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4y")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench_default.json")).read().splitlines() if x.startswith("{")][-1])
    print("DEFAULT", l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["roofline"]["frac"], l.get("graph_build_s"), (l.get("cpu_baseline") or {}).get("matches_gpu_topk"))
    for k in ("hard_case","literal_c3","c2","c5","c4_one_shard"):
        s=l.get(k) or {}
        print(k, s.get("value"), s.get("unit"), s.get("recall_at_10"), (s.get("config") or {}).get("rerankK"), s.get("error"))
except Exception as e:
    print("no line", e)
PY
