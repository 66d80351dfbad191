#!/bin/bash
# Round 4, GPU session S: four lanes per fresh neighbour in expansions with <= 16 of them (gs_quad, the pair-lane kernels), and the
# four-lane compacted form for M > 96 (the C5 build's searches).  Parity first, then the headline with the path on / off, then C5.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py -m gpu -x -q -k "compacted_pair or matches_oracle or large_batch or engineered or accept" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.txt | tee -a $O/summary.txt
for quad in 1 0; do
  JVECTOR_HIP_GS_QUAD=$quad timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_quad$quad.json 2> $O/bench_quad$quad.err
  echo "bench quad=$quad rc=$?" | tee -a $O/summary.txt
  grep -E "evaluate" $O/bench_quad$quad.err | cut -c1-200 | tail -2 | tee -a $O/summary.txt
done
timeout 900 python bench.py --gpus 1 --sub-line --workload c5 --n 10000000 > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4s")
for q in (1,0):
    try:
        l=json.loads([x for x in open(os.path.join(d,"bench_quad%d.json"%q)).read().splitlines() if x.startswith("{")][-1])
        print("QUAD",q, l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["kernel_ms_per_step"]["gsearch"], l["graph_build_s"], l["graph_build"]["search_s"])
    except Exception as e:
        print("no line", q, e)
try:
    l=json.loads([x for x in open(os.path.join(d,"c5.json")).read().splitlines() if x.startswith("{")][-1])
    print("C5", l["value"], json.dumps(l["seconds"]))
except Exception as e:
    print("no c5 line", e)
PY
