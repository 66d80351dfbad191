#!/bin/bash
# Round 5, GPU session R: (1) the reference-order parity tests in full (session Q stopped at a bookkeeping assert; the adjacency was equal);
# (2) SORTED LISTS (bl_sorted_lists = 1: the classic path's symmetric scores stored, lists kept sorted — must build the identical
# graph): identity test on the device, then the headline build (seconds; rerankK / recall / QPS must equal the classic build's) and
# BASELINE config 5 at 10M x 1536 (nodes/s), against 44.9 s / rerankK 74-76 and 183 k nodes/s
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_builder_reference_order_gpu.py tests/test_builder.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|reference order|builder|FAILED|Error" $O/pytest.log | tail -14 | tee -a $O/summary.txt
JVECTOR_HIP_BL_SORTED_LISTS=1 timeout 900 python bench.py --no-sub-workloads --no-cpu-baseline --steps 5 > $O/c3_sorted.out 2> $O/c3_sorted.err; echo "c3 sorted rc=$?" | tee -a $O/summary.txt
grep -E "\[evaluate\]" $O/c3_sorted.err | tail -3 | tee -a $O/summary.txt
python - $O/c3_sorted.out <<'PY' | tee -a $O/summary.txt
import json,sys
l=[json.loads(x) for x in open(sys.argv[1]).read().strip().splitlines() if x.startswith("{")][-1]
print("  C3 sorted lists:", l["value"], l["unit"], "ms/step", l["ms_per_step"], "recall", l.get("recall_at_10"), "rerankK", l["config"].get("rerankK"))
PY
python - <<'PY' | tee -a $O/summary.txt
import json
d=json.load(open("bench_full.json"))
print("  build:", d.get("graph_build_s"), d.get("graph_build"), "avg_expanded", d.get("avg_expanded"))
PY
cp bench_full.json $O/c3_sorted_full.json 2>/dev/null
JVECTOR_HIP_BL_SORTED_LISTS=1 timeout 900 python bench.py --workload c5 --n 10000000 --no-cpu-baseline > $O/c5_sorted.out 2> $O/c5_sorted.err; echo "c5 sorted rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
d=json.load(open("bench_full.json"))
print("  C5 sorted lists:", d["value"], d["unit"], d.get("seconds"), d.get("build"), d.get("recall_at_10_by_rerankK"), "prune_roofline_frac", d.get("prune_roofline_frac"))
PY
cp bench_full.json $O/c5_sorted_full.json 2>/dev/null
