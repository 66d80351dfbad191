#!/bin/bash
# Round 6, GPU session K: after the UB8 / LUTR axes left gs_worker and the bound scan's blocks went XCD-aware: the whole -m gpu suite,
# smoke, the headline, and the flat filter's counters again (profile_r6.sh section B only) for traffic_r6.json.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1
echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6k")
l=[json.loads(x) for x in open(os.path.join(d,"bench.json")).read().strip().splitlines() if x.startswith("{")][-1]
print("headline", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), "flat_mode", (l.get("workloads") or {}).get("flat_mode"))
PY
# ---- flat filter counters, XCD-aware order ----
K=$R/gpurun_out/prof_r6_10m; OO=/tmp/prof_r6k; C=/tmp/jv_index_10000000.npz
mkdir -p $K $OO
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --n 10000000 --index-cache $C --steps 1 --warmup 1 --no-flat --no-cpu-baseline --no-sub-workloads --cal-queries 256 --eval-queries 256 > $K/cache_build.log 2>&1
extract() { f=$(find $OO/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
FLATC2="--workload c2 --no-cpu-baseline --steps 5 --warmup 1"
FLATC4="--workload c4 --no-cpu-baseline --steps 3 --warmup 1"
FLATFM="--n 10000000 --index-cache $C --no-cpu-baseline --no-sub-workloads --steps 1 --warmup 1 --cal-queries 256 --eval-queries 256 --rerank 74"
for W in c2 c4 fm; do
  case $W in c2) A="$FLATC2";; c4) A="$FLATC4";; fm) A="$FLATFM";; esac
  for CTR in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    T=${W}_$(echo $CTR | cut -d' ' -f1)
    timeout 700 rocprofv3 --pmc $CTR --output-format csv -d $OO/$T -o bench -- python $R/bench.py $A > $K/$T.log 2>&1
    extract $T
  done
  timeout 700 rocprofv3 --kernel-trace --output-format csv -d $OO/${W}_trace -o bench -- python $R/bench.py $A > $K/${W}_trace.log 2>&1
  f=$(find $OO/${W}_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/${W}_kernel_trace_jv.csv; grep -E "jv::adc|jv::topk|jv::exact" $f >> $K/${W}_kernel_trace_jv.csv; }
done
ls $K | wc -l
