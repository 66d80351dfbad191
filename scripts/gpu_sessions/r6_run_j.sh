#!/bin/bash
# Round 6, GPU session J: the bound scan with its blocks in XCD-aware order (a tile's 16 query groups back to back on ONE XCD) against the
# plain (group, tile) order: parity of the flat paths, then C4 one shard, C2 and the headline's flat_mode, each both ways; FETCH_SIZE of
# the new order at C4.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_flat_bq_gpu.py tests/test_gpu_parity.py tests/test_zz_sharded_graph_gpu.py tests/test_sharded.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
for w in c4 c2; do
  for ord in xcd plain; do
    if [ $ord = plain ]; then export JVECTOR_HIP_ADC_BQ_PLAIN_ORDER=1; else unset JVECTOR_HIP_ADC_BQ_PLAIN_ORDER; fi
    timeout 900 python bench.py --workload $w --no-cpu-baseline > $O/${w}_$ord.json 2> $O/${w}_$ord.err
    echo "$w $ord rc=$?" | tee -a $O/summary.txt
  done
done
unset JVECTOR_HIP_ADC_BQ_PLAIN_ORDER
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_j -o bench -- python $R/bench.py --workload c4 --no-cpu-baseline --steps 3 --warmup 1 > $O/c4_fetch.log 2>&1
f=$(find /tmp/prof_j -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $O/c4_fetch_jv.csv; grep -E "adc_bq_kernel" $f >> $O/c4_fetch_jv.csv; }
cd $R
python - <<'PY' | tee -a $O/summary.txt
import json,os,csv,statistics
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6j")
for f in ("c4_xcd.json","c4_plain.json","c2_xcd.json","c2_plain.json"):
    try:
        l=json.loads(open(os.path.join(d,f)).read().strip().splitlines()[-1])
        r=l.get("roofline",{})
        print(f, round(l["value"]), "QPS", l.get("kernel_ms_per_step"), "frac", r.get("frac"), "scan ms", r.get("bound_scan_ms_per_launch"), "exact ms", r.get("exact_stage_ms_per_launch"), "traffic", r.get("traffic"), r.get("hbm_traffic_over_compulsory"))
    except Exception as e:
        print(f, "failed", e)
try:
    rows=list(csv.DictReader(open(os.path.join(d,"c4_fetch_jv.csv"))))
    g=max(int(r["Grid_Size"]) for r in rows)
    v=[float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"])==g]
    print("c4 adc_bq_kernel FETCH_SIZE x2 per launch (XCD order): %.2f GB over %d launches" % (statistics.mean(v)*2048/1e9, len(v)))
except Exception as e:
    print("fetch failed", e)
PY
