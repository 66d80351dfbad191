#!/bin/bash
set -u
O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests/test_retain_diverse.py tests/test_zz_build_score_gpu.py tests/test_builder.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 1200 python bench.py --workload c5 --n 1000000 > $O/bench_c5_1m.json 2> $O/bench_c5_1m.err
python - <<PY
import json
c=json.load(open("$O/bench_c5_1m.json")); print("C5 1M", round(c["value"]), "nodes/s", {k: round(v,2) for k,v in c["seconds"].items()}, c["recall_at_10_by_rerankK"])
PY
timeout 2400 python bench.py --no-flat --no-cpu-baseline > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "\[build\] \{|evaluate|Error|error|Traceback" $O/bench_10m.err | tail -4 | cut -c1-500
python - <<PY
import json
d=json.load(open("$O/bench_10m.json"))
print("10M", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "build_s", round(d["graph_build_s"],1))
PY
