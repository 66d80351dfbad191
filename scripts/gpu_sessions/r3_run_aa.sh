#!/bin/bash
# Session AA (round 3): conservative pre-filter in the flat scan's epilogue (no similarity transform for sure-fail pairs):
# parity (flat search / ADC tests, fuzz_kernels) and the C2 / C4 / flat-mode rates (before: 260 k, 7.8 k QPS).
mkdir -p gpurun_out/r3_aa && export TMPDIR=/tmp
K=gpurun_out/r3_aa
timeout 900 python -m pytest tests -m gpu -q -k "flat or adc or sharded or parity" > $K/pytest.log 2>&1; echo "pytest rc=$?" >> $K/summary.txt; tail -2 $K/pytest.log >> $K/summary.txt
timeout 200 python scripts/fuzz_kernels.py 60 41 > $K/fuzz_kernels.log 2>&1; echo "fuzz_kernels rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_kernels.log >> $K/summary.txt
timeout 600 python bench.py --workload c2 --no-cpu-baseline > $K/c2.json 2> $K/c2.err; echo "c2 rc=$?" >> $K/summary.txt
timeout 900 python bench.py --workload c4 --no-cpu-baseline > $K/c4.json 2> $K/c4.err; echo "c4 rc=$?" >> $K/summary.txt
python - <<'PY' >> gpurun_out/r3_aa/summary.txt
import json
for f in ("c2", "c4"):
    try:
        l = json.loads(open(f"gpurun_out/r3_aa/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10")}, l["config"].get("rerankK"), l["roofline"]["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
