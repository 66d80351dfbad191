#!/bin/bash
# round-2 GPU session J: layered engine-built graph — unit test, 1M comparison, then the 10M default bench
set -u
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_builder.py -x -q -m gpu -s 2>&1 | tail -4 | tee $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --n 1000000 --steps 4 --warmup 1 --no-flat --no-cpu-baseline > $O/bench_1m_engine.json 2> $O/bench_1m_engine.err
grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_1m_engine.err | tail -8 | cut -c1-600
python - <<PY
import json
d=json.load(open("$O/bench_1m_engine.json"))
print("1M engine", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1))
PY
JVECTOR_HIP_GRAPH_TIMING=1 timeout 2400 python bench.py --no-flat --no-cpu-baseline > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_10m.err | tail -14 | cut -c1-600
python - <<PY
import json
d=json.load(open("$O/bench_10m.json"))
print("10M engine", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1), d["kernel_ms_per_step"])
PY
