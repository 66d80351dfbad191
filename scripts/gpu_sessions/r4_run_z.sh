#!/bin/bash
# Round 4, GPU session Z (the round's last GPU minutes): the robust prune with a slot's entries spread over the test's idle lanes
# (rd_split, rd_pair_sum_split) — parity both ways, the headline build, C5.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4z; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_retain_diverse.py tests/test_builder.py tests/test_zz_build_score_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "evaluate" $O/bench.err | cut -c1-200 | tail -1 | tee -a $O/summary.txt
timeout 200 python bench.py --gpus 1 --sub-line --workload c5 --n 10000000 > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4z")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench.json")).read().splitlines() if x.startswith("{")][-1])
    print("SPLIT", l["value"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["graph_build_s"], json.dumps(l["graph_build"]))
except Exception as e:
    print("no line", e)
try:
    l=json.loads([x for x in open(os.path.join(d,"c5.json")).read().splitlines() if x.startswith("{")][-1])
    print("C5", l["value"], json.dumps(l["seconds"]), json.dumps(l["recall_at_10_by_rerankK"]))
except Exception as e:
    print("no c5 line", e)
PY
