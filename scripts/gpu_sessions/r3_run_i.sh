#!/bin/bash
# Session I (round 3): NVQ on hardware — parity tests, kernel rates, and the headline with NVQ rows as the reranker.
mkdir -p gpurun_out/r3_i && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_nvq_gpu.py -x -q > gpurun_out/r3_i/pytest_nvq.log 2>&1; echo "pytest_nvq rc=$?" >> gpurun_out/r3_i/summary.txt
tail -5 gpurun_out/r3_i/pytest_nvq.log >> gpurun_out/r3_i/summary.txt
timeout 600 python scripts/nvq_bench.py 2000000 768 2 16384 95 > gpurun_out/r3_i/nvq_bench.json 2> gpurun_out/r3_i/nvq_bench.err; echo "nvq_bench rc=$?" >> gpurun_out/r3_i/summary.txt
cat gpurun_out/r3_i/nvq_bench.json >> gpurun_out/r3_i/summary.txt
timeout 900 python bench.py --reranker nvq --no-flat --no-cpu-baseline > gpurun_out/r3_i/bench_nvq.json 2> gpurun_out/r3_i/bench_nvq.err; echo "bench_nvq rc=$?" >> gpurun_out/r3_i/summary.txt
grep -a "nvq\]\|calibrate\|evaluate" gpurun_out/r3_i/bench_nvq.err | tail -30 >> gpurun_out/r3_i/summary.txt
python - <<'PY' >> gpurun_out/r3_i/summary.txt
import json
try:
    l = json.loads(open("gpurun_out/r3_i/bench_nvq.json").read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "recall_se", "reranker", "nvq", "kernel_ms_per_step")}, l["config"]["rerankK"], l.get("rerank"))
except Exception as e:
    print("bench line unreadable", e)
PY
