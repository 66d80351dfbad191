#!/bin/bash
set -u
O=gpurun_out/r2n; mkdir -p $O
for BEAM in 200; do
  timeout 2400 python bench.py --build-beam $BEAM --no-flat --no-cpu-baseline > $O/bench_10m_beam$BEAM.json 2> $O/bench_10m_beam$BEAM.err
  grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_10m_beam$BEAM.err | tail -8 | cut -c1-400
  python - <<PY
import json
d=json.load(open("$O/bench_10m_beam$BEAM.json"))
print("10M beam $BEAM", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1))
PY
done
