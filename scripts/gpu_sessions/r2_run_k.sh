#!/bin/bash
set -u
O=gpurun_out/r2k; mkdir -p $O
timeout 600 python -m pytest tests/test_builder.py -x -q -m gpu -s 2>&1 | tail -3 | tee $O/pytest.log
for P in 2; do
  timeout 900 python bench.py --n 1000000 --build-passes $P --steps 4 --warmup 1 --no-flat --no-cpu-baseline > $O/bench_1m_p$P.json 2> $O/bench_1m_p$P.err
  grep -E "\[build\] \{|evaluate|Error|error|Traceback" $O/bench_1m_p$P.err | tail -4 | cut -c1-500
  python - <<PY
import json
d=json.load(open("$O/bench_1m_p$P.json"))
print("1M passes $P", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1))
PY
  timeout 2400 python bench.py --build-passes $P --no-flat --no-cpu-baseline > $O/bench_10m_p$P.json 2> $O/bench_10m_p$P.err
  grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_10m_p$P.err | tail -12 | cut -c1-500
  python - <<PY
import json
d=json.load(open("$O/bench_10m_p$P.json"))
print("10M passes $P", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1), d["kernel_ms_per_step"])
PY
done
