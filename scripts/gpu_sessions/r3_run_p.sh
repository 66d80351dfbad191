#!/bin/bash
# Session P (round 3): differential fuzzing on the final library (NVQ, scoring kernels, traversal) + the default bench line again.
mkdir -p gpurun_out/r3_p && export TMPDIR=/tmp
K=gpurun_out/r3_p
timeout 200 python scripts/fuzz_nvq.py 90 11 > $K/fuzz_nvq.log 2>&1; echo "fuzz_nvq rc=$?" >> $K/summary.txt; tail -2 $K/fuzz_nvq.log >> $K/summary.txt
timeout 200 python scripts/fuzz_nvq.py 60 12 >> $K/fuzz_nvq.log 2>&1; echo "fuzz_nvq(2) rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_nvq.log >> $K/summary.txt
timeout 200 python scripts/fuzz_kernels.py 60 21 > $K/fuzz_kernels.log 2>&1; echo "fuzz_kernels rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_kernels.log >> $K/summary.txt
timeout 200 python scripts/fuzz_traversal.py 60 22 > $K/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_traversal.log >> $K/summary.txt
timeout 900 python bench.py > $K/bench_default.json 2> $K/bench_default.err; echo "bench rc=$?" >> $K/summary.txt
python - <<'PY' >> gpurun_out/r3_p/summary.txt
import json
try:
    l = json.loads(open("gpurun_out/r3_p/bench_default.json").read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s", "ground_truth_s")}, l["config"]["rerankK"], l["roofline"]["frac"], l["cpu_baseline"]["value"], l["cpu_baseline"].get("matches_gpu_topk"))
except Exception as e:
    print("bench line unreadable", e)
PY
