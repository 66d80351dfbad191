#!/bin/bash
# round-2 GPU session H: the batched builder — unit test, then the headline search on an ENGINE-built 1M graph vs the synthetic one, then C5
set -u
O=gpurun_out/r2h; mkdir -p $O
timeout 600 python -m pytest tests/test_builder.py tests/test_zz_device_traversal_gpu.py -x -q -m gpu -s 2>&1 | tail -6 | tee $O/pytest.log
for G in engine synthetic; do
  timeout 900 python bench.py --n 1000000 --graph $G --steps 4 --warmup 1 --no-flat --no-cpu-baseline > $O/bench_1m_$G.json 2> $O/bench_1m_$G.err
  grep -E "\[build\] \{|calibrate|evaluate|Error|error" $O/bench_1m_$G.err | tail -12 | cut -c1-400
  python - <<PY
import json
d=json.load(open("$O/bench_1m_$G.json"))
print("$G", round(d["value"]), "QPS rerankK", d["config"]["rerankK"], "recall", round(d["recall_at_10"],4), "visited", round(d["avg_visited"]), "expanded", round(d["avg_expanded"]), "build_s", round(d["graph_build_s"],1))
PY
done
timeout 1200 python bench.py --workload c5 --n 1000000 > $O/bench_c5_1m.json 2> $O/bench_c5_1m.err; tail -4 $O/bench_c5_1m.err | cut -c1-300; cat $O/bench_c5_1m.json | cut -c1-1800
