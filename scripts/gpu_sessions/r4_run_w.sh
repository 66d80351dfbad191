#!/bin/bash
# Round 4, GPU session W: the robust prune with INCREMENTAL tests (RdParams::chunk: a candidate remembers the slots it was tested
# against and their largest similarity; a test walks only the new slots, `chunk` at a time, and stops at the first violation).
# Parity, then the headline build with chunk 8 / 64, then C5.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4w; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_retain_diverse.py tests/test_builder.py tests/test_zz_build_score_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
for ch in 8 64; do
  JVECTOR_HIP_RD_CHUNK=$ch timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_chunk$ch.json 2> $O/bench_chunk$ch.err
  echo "bench chunk=$ch rc=$?" | tee -a $O/summary.txt
  grep -E "evaluate" $O/bench_chunk$ch.err | cut -c1-200 | tail -1 | tee -a $O/summary.txt
done
timeout 900 python bench.py --gpus 1 --sub-line --workload c5 --n 10000000 > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4w")
for ch in (8,64):
    try:
        l=json.loads([x for x in open(os.path.join(d,"bench_chunk%d.json"%ch)).read().splitlines() if x.startswith("{")][-1])
        print("CHUNK",ch, l["value"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["graph_build_s"], json.dumps(l["graph_build"]))
    except Exception as e:
        print("no line", ch, e)
try:
    l=json.loads([x for x in open(os.path.join(d,"c5.json")).read().splitlines() if x.startswith("{")][-1])
    print("C5", l["value"], json.dumps(l["seconds"]), json.dumps(l["recall_at_10_by_rerankK"]))
except Exception as e:
    print("no c5 line", e)
PY
