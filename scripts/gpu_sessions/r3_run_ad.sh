#!/bin/bash
# Session AD (round 3): the device traversal's GENERIC kernels (any PQ geometry) on the MI355X: whole -m gpu suite (new: generic
# shapes, ragged-quantizer build), the two traversal fuzzers with arbitrary (D, M) pairs + gs_generic, generic_bench.py (QPS device
# vs host on 200-d PQ-25 / 100-d PQ-12 / 128-d PQ-16 specialised vs forced generic), and the default bench line once more.
mkdir -p gpurun_out/r3_ad && export TMPDIR=/tmp
K=gpurun_out/r3_ad
timeout 1200 python -m pytest tests -m gpu -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_gpu.log | tail -1 >> $K/summary.txt
timeout 300 python scripts/fuzz_traversal.py 50 31 > $K/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_traversal.log >> $K/summary.txt
timeout 300 python scripts/fuzz_searcher.py 50 31 > $K/fuzz_searcher.log 2>&1; echo "fuzz_searcher rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_searcher.log >> $K/summary.txt
timeout 600 python scripts/generic_bench.py > $K/generic_bench.json 2> $K/generic_bench.err; echo "generic_bench rc=$?" >> $K/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
timeout 900 python bench.py > $K/bench_default.json 2> $K/bench_default.err; echo "bench rc=$?" >> $K/summary.txt
python - <<'PY' >> gpurun_out/r3_ad/summary.txt
import json
try:
    g = json.load(open("gpurun_out/r3_ad/generic_bench.json"))
    for k, v in g["shapes"].items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print("generic_bench:", e)
try:
    l = [json.loads(x) for x in open("gpurun_out/r3_ad/bench_default.json") if x.startswith("{")][-1]
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s")}, l["config"]["rerankK"], l["roofline"]["frac"])
except Exception as e:
    print("bench:", e)
PY
