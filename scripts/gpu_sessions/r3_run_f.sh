#!/bin/bash
# Round 3, GPU session F: GraphSearcher objects on the device traversal (session kernels) — parity on hardware, then their QPS
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_graph_search.py tests/test_zz_device_traversal_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 900 python scripts/searcher_bench.py > $O/searcher_bench.json 2> $O/searcher_bench.err; tail -1 $O/searcher_bench.json | tee -a $O/summary.txt; tail -3 $O/searcher_bench.err | cut -c1-300
# queries per step: 65536 vs 131072 on the cached 10M index
C=/tmp/jv_index_10m.npz
for QS in 65536 131072; do
  timeout 900 python bench.py --index-cache $C --steps 8 --warmup 2 --no-cpu-baseline --no-flat --queries $QS > $O/bench_q$QS.json 2> $O/bench_q$QS.err
  python - "$O/bench_q$QS.json" $QS <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    l=json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    print("Q", sys.argv[2], "QPS %.0f"%l["value"], "rerankK", l["config"]["rerankK"], "recall %.4f"%l["recall_at_10"], l["kernel_ms_per_step"])
except Exception as e:
    print("Q", sys.argv[2], "failed", e)
PY
done
