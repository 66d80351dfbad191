#!/bin/bash
# Round 3, GPU session D: block-argmin pq_encode on hardware (parity + speed), more builder variants, the secondary workload
# lines (C2, C4 one shard, C5 at full size) and the latent-dimension sensitivity points of the headline
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_goldens.py tests/test_builder.py tests/test_zz_pq_train_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_gpu.log | tee -a $O/summary.txt
run() { # tag, extra args
  tag=$1; shift
  timeout 700 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-flat --cal-queries 4096 --eval-queries 4096 "$@" > $O/v_$tag.json 2> $O/v_$tag.err
  python - "$O/v_$tag.json" "$tag" <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    l=json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    b=l.get("graph_build") or {}
    print("VARIANT", sys.argv[2], "rerankK", l["config"]["rerankK"], "recall %.4f"%l["recall_at_10"], "QPS %.0f"%l["value"], "gsearch_ms %.2f"%l["kernel_ms_per_step"]["gsearch"], "build_s %.1f"%l["graph_build_s"], "search/prune/backlink %.1f/%.1f/%.1f"%(b.get("search_s",0),b.get("prune_s",0),b.get("backlink_s",0)), "reprunes", b.get("reprunes"), "avg_exp %.1f"%l["avg_expanded"], "encode", l.get("encode"))
except Exception as e:
    print("VARIANT", sys.argv[2], "failed", e)
PY
}
run ovf20_beam150 --build-beam 150
run ovf20_beam125 --build-beam 125
run latent64 --latent 64
run latent128 --latent 128
timeout 600 python bench.py --workload c2 > $O/c2.json 2> $O/c2.err; tail -1 $O/c2.json | cut -c1-400 | tee -a $O/summary.txt
timeout 900 python bench.py --workload c4 --eval-queries 1024 > $O/c4_1shard.json 2> $O/c4_1shard.err; grep "\[c4\]" $O/c4_1shard.err | tee -a $O/summary.txt; tail -1 $O/c4_1shard.json | cut -c1-500 | tee -a $O/summary.txt
timeout 1200 python bench.py --workload c5 --n 10000000 > $O/c5_10m.json 2> $O/c5_10m.err; tail -1 $O/c5_10m.json | cut -c1-600 | tee -a $O/summary.txt
