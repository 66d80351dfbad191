#!/bin/bash
# Round 4, GPU session R: the compacted pair form (graph_search_pairc_kernel) — parity on hardware, then what it does to the build
# (the builder's searches run over 64-wide working rows: until now one lane per neighbour)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4r; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py tests/test_builder.py -m gpu -x -q -k "compacted_pair or builder_gpu or two_tier" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.txt | tee -a $O/summary.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "calibrate|evaluate|layered" $O/bench.err | cut -c1-300 | tail -8 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4r")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench.json")).read().splitlines() if x.startswith("{")][-1])
    print("PAIRC", l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["graph_build_s"])
    print("build", json.dumps(l["graph_build"]))
except Exception as e:
    print("no line", e)
PY
