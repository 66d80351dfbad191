#!/bin/bash
# Session AB (round 3): closing validation of the final library: whole -m gpu suite, smoke, the default bench line, and the
# traversal kernel's rocprofv3 duration + FETCH / WRITE counters once more (gs_refill / phase loop are in the kernel now).
mkdir -p gpurun_out/r3_ab && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_ab
timeout 1200 python -m pytest tests -m gpu -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_gpu.log | tail -1 >> $K/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
C=/tmp/jv_index_10000000.npz
timeout 900 python bench.py --index-cache $C > $K/bench_default.json 2> $K/bench_default.err; echo "bench rc=$?" >> $K/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof/stats -o b -- python $R/bench.py --index-cache $C --no-cpu-baseline --no-flat --steps 5 > $K/prof_stats.log 2>&1
cp /tmp/abprof/stats/*kernel_stats.csv $K/kernel_stats.csv 2>/dev/null
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --output-format csv -d /tmp/abprof/$CTR -o b -- python $R/bench.py --index-cache $C --no-cpu-baseline --no-flat --steps 3 --warmup 1 > $K/prof_$CTR.log 2>&1
  f=$(find /tmp/abprof/$CTR -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/pmc_$CTR.csv; grep -E "graph_search_kernel|exact_gather_tr" $f >> $K/pmc_$CTR.csv; }
done
cd $R
python - <<'PY' >> gpurun_out/r3_ab/summary.txt
import json, csv
l = [json.loads(x) for x in open("gpurun_out/r3_ab/bench_default.json") if x.startswith("{")][-1]
print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s")}, l["config"]["rerankK"], l["roofline"]["frac"], l["cpu_baseline"]["value"], l["cpu_baseline"].get("matches_gpu_topk"), l["kernel_ms_per_step"])
for r in csv.DictReader(open("gpurun_out/r3_ab/kernel_stats.csv")):
    if "graph_search_kernel" in r["Name"] or "exact_gather_tr" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
