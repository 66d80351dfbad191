#!/bin/bash
# Round 4, GPU session T: the robust prune TABLE-FREE (retain_diverse_tf_kernel: pair-table entries recomputed from the L2-resident
# codebook instead of 4-byte look-ups into a 12.6 / 25 MB table that misses L2) — parity both ways, then the headline build and C5.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4t; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_retain_diverse.py tests/test_builder.py -m gpu -x -q > $O/pytest_tf.txt 2>&1
echo "pytest (table-free) rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_tf.txt | tee -a $O/summary.txt
JVECTOR_HIP_RD_TABLE_FREE=0 timeout 600 python -m pytest tests/test_retain_diverse.py -m gpu -x -q > $O/pytest_table.txt 2>&1
echo "pytest (table) rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_table.txt | tee -a $O/summary.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "evaluate" $O/bench.err | cut -c1-200 | tail -2 | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --sub-line --workload c5 --n 10000000 > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4t")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench.json")).read().splitlines() if x.startswith("{")][-1])
    print("HEADLINE", l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["graph_build_s"])
    print("build", json.dumps(l["graph_build"]))
except Exception as e:
    print("no line", e)
try:
    l=json.loads([x for x in open(os.path.join(d,"c5.json")).read().splitlines() if x.startswith("{")][-1])
    print("C5", l["value"], json.dumps(l["seconds"]), json.dumps(l["recall_at_10_by_rerankK"]))
except Exception as e:
    print("no c5 line", e)
PY
