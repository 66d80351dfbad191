#!/bin/bash
set -u
O=gpurun_out/r2f; mkdir -p $O
timeout 300 python -m pytest tests/test_retain_diverse.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
for Q in 16384 32768 65536; do
  JVECTOR_HIP_GRAPH_TIMING=1 timeout 600 python bench.py --n 1000000 --queries $Q --steps 4 --warmup 1 --no-flat --no-cpu-baseline --rerank 125 --cal-queries 1024 --eval-queries 1024 > $O/bench_1m_q$Q.json 2> $O/bench_1m_q$Q.err
  python - <<PY
import json
d=json.load(open("$O/bench_1m_q$Q.json"))
print($Q, round(d["value"]), "QPS", round(d["ms_per_step"],2), "ms/step", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v})
PY
  grep -c "overflow=[1-9]" $O/bench_1m_q$Q.err
done
