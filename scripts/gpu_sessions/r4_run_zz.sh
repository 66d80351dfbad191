#!/bin/bash
# Round 4, the last 100 GPU-seconds: the prune kernel's new staging (ids / scores / code rows: independent wide loads) — parity on
# hardware first, then a 1M-vector build with it and without it (the 10M build is the driver's to time).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4zz; mkdir -p $O
cd $R
timeout 45 python -m pytest tests/test_retain_diverse.py tests/test_builder.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
for w in 1 0; do
  JVECTOR_HIP_RD_WIDE_STAGE=$w timeout 30 python bench.py --n 1000000 --queries 16384 --steps 2 --warmup 1 --cal-queries 1024 --eval-queries 2048 --no-flat --no-cpu-baseline --no-sub-workloads > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  echo "bench wide=$w rc=$?" | tee -a $O/summary.txt
  python - $O/bench_wide$w.json <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    l=json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    print(l["config"]["rerankK"], l["recall_at_10"], l["graph_build_s"], json.dumps(l["graph_build"]))
except Exception as e:
    print("no line", e)
PY
done
