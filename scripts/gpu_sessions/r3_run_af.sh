#!/bin/bash
# Session AF (round 3): the GraphSearcher-object rates on the FINAL library (the session kernels were recompiled around the chunk loop
# and the generic form) and a rocprofv3 kernel trace of generic_bench.py (which traversal build served which shape, with durations).
mkdir -p gpurun_out/r3_af && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_af
timeout 400 python scripts/searcher_bench.py > $K/searcher_bench.json 2> $K/searcher_bench.err; echo "searcher_bench rc=$?" >> $K/summary.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/afprof -o g -- python $R/scripts/generic_bench.py --n 100000 --queries 8192 --host-queries 512 > $K/generic_prof.json 2> $K/generic_prof.err; echo "generic prof rc=$?" >> $K/summary.txt
cd $R
f=$(find /tmp/afprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/kernel_stats.csv; grep -E "graph_search|frontier|exact_gather|retain_diverse" $f >> $K/kernel_stats.csv; }
python - <<'PY' >> gpurun_out/r3_af/summary.txt
import json, csv
try:
    s = json.load(open("gpurun_out/r3_af/searcher_bench.json"))
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in s.items() if not isinstance(v, (dict, list))})
except Exception as e:
    print("searcher_bench:", e)
try:
    for r in csv.DictReader(open("gpurun_out/r3_af/kernel_stats.csv")):
        print(r["Name"][:90], r["Calls"], r["AverageNs"])
except Exception as e:
    print("stats:", e)
PY
