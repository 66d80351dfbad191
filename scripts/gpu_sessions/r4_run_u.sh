#!/bin/bash
# Round 4, GPU session U: two graph-quality knobs of the layered build — alpha = 1.0 for the insert phase (the improve pass keeps
# 1.2: DiskANN's two-pass schedule), and a wider beam (200) for the improve pass only.  Calibrated rerankK / QPS / build time each.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
run() {  # name, env assignment
  env $2 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench $1 rc=$?" | tee -a $O/summary.txt
  grep -E "calibrate|evaluate" $O/bench_$1.err | cut -c1-200 | tail -4 | tee -a $O/summary.txt
}
run insert_alpha100 JVECTOR_HIP_BL_INSERT_ALPHA_X100=100
run improve_beam200 JVECTOR_HIP_BL_IMPROVE_BEAM=200
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4u")
for n in ("insert_alpha100","improve_beam200"):
    try:
        l=json.loads([x for x in open(os.path.join(d,"bench_%s.json"%n)).read().splitlines() if x.startswith("{")][-1])
        print(n, l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["avg_visited"], l["graph_build_s"], json.dumps(l["graph_build"]))
    except Exception as e:
        print("no line", n, e)
PY
