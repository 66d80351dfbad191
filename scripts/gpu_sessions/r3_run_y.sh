#!/bin/bash
# Session Y (round 3): spill-tier refill in the traversal kernel: parity (traversal / searcher tests, fuzz_traversal), the
# searcher-object rates again (threshold searches were 12x the plain search), and the default bench line (must not move).
mkdir -p gpurun_out/r3_y && export TMPDIR=/tmp
K=gpurun_out/r3_y
timeout 900 python -m pytest tests -m gpu -q -k "searcher or session or traversal or graph_search or builder" > $K/pytest.log 2>&1; echo "pytest rc=$?" >> $K/summary.txt; tail -2 $K/pytest.log >> $K/summary.txt
timeout 200 python scripts/fuzz_traversal.py 45 31 > $K/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_traversal.log >> $K/summary.txt
timeout 200 python scripts/fuzz_build.py 30 32 > $K/fuzz_build.log 2>&1; echo "fuzz_build rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_build.log >> $K/summary.txt
SEARCHER_BENCH_TIMING=1 timeout 600 python scripts/searcher_bench.py > $K/searcher_bench.json 2> $K/searcher_bench.err; echo "searcher_bench rc=$?" >> $K/summary.txt
cat $K/searcher_bench.json >> $K/summary.txt
timeout 900 python bench.py --no-flat --no-cpu-baseline > $K/bench.json 2> $K/bench.err; echo "bench rc=$?" >> $K/summary.txt
python - <<'PY' >> gpurun_out/r3_y/summary.txt
import json
l = json.loads(open("gpurun_out/r3_y/bench.json").read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s")}, l["config"]["rerankK"], l["kernel_ms_per_step"])
PY
