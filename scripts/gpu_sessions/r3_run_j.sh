#!/bin/bash
# Session J (round 3): NVQ gather with 128-byte tiles + the packed short division; full NVQ parity file; rates; headline with NVQ rerank.
mkdir -p gpurun_out/r3_j && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_nvq_gpu.py -q > gpurun_out/r3_j/pytest_nvq.log 2>&1; echo "pytest_nvq rc=$?" >> gpurun_out/r3_j/summary.txt
tail -8 gpurun_out/r3_j/pytest_nvq.log >> gpurun_out/r3_j/summary.txt
timeout 600 python scripts/nvq_bench.py 2000000 768 2 16384 95 > gpurun_out/r3_j/nvq_bench.json 2> gpurun_out/r3_j/nvq_bench.err; echo "nvq_bench rc=$?" >> gpurun_out/r3_j/summary.txt
cat gpurun_out/r3_j/nvq_bench.json >> gpurun_out/r3_j/summary.txt
timeout 900 python bench.py --reranker nvq --no-flat --no-cpu-baseline > gpurun_out/r3_j/bench_nvq.json 2> gpurun_out/r3_j/bench_nvq.err; echo "bench_nvq rc=$?" >> gpurun_out/r3_j/summary.txt
grep -a "nvq\]\|evaluate" gpurun_out/r3_j/bench_nvq.err | tail -5 >> gpurun_out/r3_j/summary.txt
python - <<'PY' >> gpurun_out/r3_j/summary.txt
import json
try:
    l = json.loads(open("gpurun_out/r3_j/bench_nvq.json").read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "recall_se", "reranker", "nvq", "kernel_ms_per_step")}, l["config"]["rerankK"], l.get("rerank"))
except Exception as e:
    print("bench line unreadable", e)
PY
