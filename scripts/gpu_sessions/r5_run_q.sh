#!/bin/bash
# Round 5, GPU session Q: the builder in reference order (bl_ref_order = 1) — parity tests on the device (one-node batches == the
# oracle's one-thread GraphIndexBuilder; layered; batched contract), then what the mode does to the headline build (seconds, rerankK,
# QPS) and to BASELINE config 5 (nodes/s) against the default mode's numbers of session K / G (44.9 s, rerankK 74-76, 183 k nodes/s)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_builder_reference_order_gpu.py tests/test_builder.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|reference order|builder" $O/pytest.log | tail -12 | tee -a $O/summary.txt
JVECTOR_HIP_BL_REF_ORDER=1 timeout 900 python bench.py --no-sub-workloads --no-cpu-baseline --steps 5 > $O/c3_ref.out 2> $O/c3_ref.err; echo "c3 ref rc=$?" | tee -a $O/summary.txt
grep -E "^\[build\]|\[evaluate\]|rerankK" $O/c3_ref.err | tail -8 | tee -a $O/summary.txt
python - $O/c3_ref.out <<'PY' | tee -a $O/summary.txt
import json,sys
l=[json.loads(x) for x in open(sys.argv[1]).read().strip().splitlines() if x.startswith("{")][-1]
print("  C3 ref-order:", l["value"], l["unit"], "ms/step", l["ms_per_step"], "recall", l.get("recall_at_10"), "rerankK", l["config"].get("rerankK"), "build_s", l.get("graph_build_s"), "avg_expanded", l.get("avg_expanded"))
PY
cp bench_full.json $O/c3_ref_full.json 2>/dev/null
JVECTOR_HIP_BL_REF_ORDER=1 timeout 900 python bench.py --workload c5 --no-cpu-baseline > $O/c5_ref.out 2> $O/c5_ref.err; echo "c5 ref rc=$?" | tee -a $O/summary.txt
python - $O/c5_ref.out <<'PY' | tee -a $O/summary.txt
import json,sys
l=[json.loads(x) for x in open(sys.argv[1]).read().strip().splitlines() if x.startswith("{")][-1]
print("  C5 ref-order:", l["value"], l["unit"], "ms/step", l["ms_per_step"], {k: l[k] for k in l if "prune" in k or "build" in k or "recall" in k})
PY
tail -5 $O/c5_ref.err | tee -a $O/summary.txt
