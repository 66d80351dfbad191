#!/bin/bash
# Session W (round 3): resume() on the device (the session kernel replays the searcher's history): parity + cost.
mkdir -p gpurun_out/r3_w && export TMPDIR=/tmp
K=gpurun_out/r3_w
timeout 900 python -m pytest tests -m gpu -q -k "searcher or session or traversal or graph_search" > $K/pytest.log 2>&1; echo "pytest rc=$?" >> $K/summary.txt; tail -3 $K/pytest.log >> $K/summary.txt
timeout 200 python scripts/fuzz_build.py 40 9 > $K/fuzz_build.log 2>&1; echo "fuzz_build rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_build.log >> $K/summary.txt
timeout 600 python scripts/searcher_bench.py > $K/searcher_bench.json 2> $K/searcher_bench.err; echo "searcher_bench rc=$?" >> $K/summary.txt
cat $K/searcher_bench.json >> $K/summary.txt
