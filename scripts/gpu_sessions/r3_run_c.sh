#!/bin/bash
# Round 3, GPU session C: (1) speculative prefetch A/B on the cached 10M index   (2) builder parameter exploration at 10M: which
# construction settings lower the rerankK the graph needs for recall@10 >= 0.95 (every 10 rerankK is ~6 % QPS)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_gpu.log | tee -a $O/summary.txt
C=/tmp/jv_index_10m.npz
SW="JVECTOR_HIP_GS_PREFETCH=1,JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_PREFETCH=1;JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_PREFETCH=1,JVECTOR_HIP_GS_V1_LOG2=0,JVECTOR_HIP_GS_PROF=1"
JVECTOR_BENCH_ENV_SWEEP="$SW" timeout 900 python bench.py --index-cache $C --steps 10 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "sweep|prof\] clocks|evaluate|\[build\] \{" $O/bench_10m.err | cut -c1-300 | tee -a $O/summary.txt
run() { # tag, extra args
  tag=$1; shift
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-flat --cal-queries 4096 --eval-queries 4096 "$@" > $O/v_$tag.json 2> $O/v_$tag.err
  python - "$O/v_$tag.json" "$tag" <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    l=json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    b=l.get("graph_build") or {}
    print("VARIANT", sys.argv[2], "rerankK", l["config"]["rerankK"], "recall %.4f"%l["recall_at_10"], "QPS %.0f"%l["value"], "gsearch_ms %.2f"%l["kernel_ms_per_step"]["gsearch"], "build_s %.1f"%l["graph_build_s"], "search/prune/backlink %.1f/%.1f/%.1f"%(b.get("search_s",0),b.get("prune_s",0),b.get("backlink_s",0)), "reprunes", b.get("reprunes"), "avg_exp %.1f"%l["avg_expanded"])
except Exception as e:
    print("VARIANT", sys.argv[2], "failed", e)
PY
}
run ovf20 --build-overflow 2.0
run ovf15_beam150 --build-overflow 1.5 --build-beam 150
run alpha14 --build-alpha 1.4
run batch32k --build-max-batch 32768
tail -5 $O/v_ovf20.err | cut -c1-200
