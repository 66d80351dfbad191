#!/bin/bash
# Session AE (round 3): adjacency rows wider than 64 (traversal chunk loop, frontier kernels' blockIdx.y chunks, host multi-word
# masks) + the corrected ragged-quantizer build test on the MI355X: whole -m gpu suite, fuzz_traversal with degrees up to 130,
# smoke, and the default bench line (the non-pair kernels were recompiled around the chunk loop; the headline uses the pair form).
mkdir -p gpurun_out/r3_ae && export TMPDIR=/tmp
K=gpurun_out/r3_ae
timeout 1200 python -m pytest tests -m gpu -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_gpu.log | tail -1 >> $K/summary.txt
timeout 300 python scripts/fuzz_traversal.py 45 41 > $K/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_traversal.log >> $K/summary.txt
timeout 300 python scripts/fuzz_searcher.py 35 41 > $K/fuzz_searcher.log 2>&1; echo "fuzz_searcher rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_searcher.log >> $K/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
timeout 900 python bench.py --no-cpu-baseline > $K/bench_default.json 2> $K/bench_default.err; echo "bench rc=$?" >> $K/summary.txt
python - <<'PY' >> gpurun_out/r3_ae/summary.txt
import json
try:
    l = [json.loads(x) for x in open("gpurun_out/r3_ae/bench_default.json") if x.startswith("{")][-1]
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s")}, l["config"]["rerankK"], l["roofline"]["frac"])
except Exception as e:
    print("bench:", e)
PY
