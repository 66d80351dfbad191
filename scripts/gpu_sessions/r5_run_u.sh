#!/bin/bash
# Round 5, GPU session U: the builder's searches in the register-table bound form (gs_ubrc = 1: UBR over the compacted fresh list of
# the 64-wide working rows) — identity test on the device (same graph, byte for byte), then the headline build: search seconds against
# the plain compacted form's 21.0 s, same rerankK / recall expected
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_builder.py -m gpu -q -x -k "bound_form or sorted_lists" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
JVECTOR_HIP_GS_UBRC=1 timeout 900 python bench.py --no-sub-workloads --no-cpu-baseline --no-flat --steps 5 > $O/c3_ubrc.out 2> $O/c3_ubrc.err; echo "c3 ubrc rc=$?" | tee -a $O/summary.txt
grep -E "\[evaluate\]" $O/c3_ubrc.err | tail -2 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
d=json.load(open("bench_full.json"))
print("   ", d["value"], "QPS rerankK", d["config"].get("rerankK"), "recall", d.get("recall_at_10"), "avg_expanded", d.get("avg_expanded"), "build", {k: round(v, 2) if isinstance(v, float) else v for k, v in d.get("graph_build", {}).items() if k in ("search_s", "prune_s", "backlink_s", "total_s", "reprunes", "visited", "expanded")})
PY
cp bench_full.json $O/c3_ubrc_full.json 2>/dev/null
