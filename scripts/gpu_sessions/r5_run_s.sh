#!/bin/bash
# Round 5, GPU session S: why does the batched build in reference order need rerankK 150 where the classic build needs 74 (session Q)?
# Two ablations on the headline build: (V1) the classic path's symmetric scores, stored, WITH the diverseBefore shortcut
# (bl_sorted_lists = 2); (V2) the reference's stored asymmetric scores WITHOUT the shortcut (bl_ref_order = 2).  Build seconds and the
# rerankK the calibration settles on; plus the 30 000-node comparison of the three list forms at one rerankK.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5s; mkdir -p $O
cd $R
for V in "BL_SORTED_LISTS=2" "BL_REF_ORDER=2"; do
  env JVECTOR_HIP_$V timeout 900 python bench.py --no-sub-workloads --no-cpu-baseline --no-flat --steps 3 > $O/c3_$V.out 2> $O/c3_$V.err; echo "c3 $V rc=$?" | tee -a $O/summary.txt
  grep -E "\[evaluate\]" $O/c3_$V.err | tail -2 | tee -a $O/summary.txt
  python - <<'PY' | tee -a $O/summary.txt
import json
d=json.load(open("bench_full.json"))
print("   ", d["value"], "QPS rerankK", d["config"].get("rerankK"), "recall", d.get("recall_at_10"), "avg_expanded", d.get("avg_expanded"), "build", {k: round(v, 2) if isinstance(v, float) else v for k, v in d.get("graph_build", {}).items() if k in ("search_s", "prune_s", "backlink_s", "total_s", "reprunes")})
PY
done
