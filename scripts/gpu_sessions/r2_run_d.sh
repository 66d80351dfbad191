#!/bin/bash
# round-2 GPU session D: dense MFMA rewrite + adc_mq XCD map parity, micro-benchmarks, then the 10M profile (profile_r2.sh)
set -u
O=gpurun_out/r2d; mkdir -p $O
timeout 600 python -m pytest tests/test_zz_exact_dense_gpu.py tests/test_gpu_parity.py tests/test_sharded.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.log
timeout 400 python scripts/microbench.py --out $O/microbench.json > $O/microbench.log 2>&1; python - <<'PY'
import json
d = json.load(open("gpurun_out/r2d/microbench.json"))
for k, v in d.items():
    if k.startswith("exact") or k.startswith("adc_scan") or k.startswith("hbm"):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
bash scripts/profile_r2.sh r2_10m 10000000 2>&1 | tail -30
