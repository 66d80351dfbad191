#!/bin/bash
# Session V (round 3): robust-prune kernel evaluating up to 8 candidates per batch (speculative, verdicts consumed in order): parity (prune / builder tests, fuzz_build) and the
# C5 build at 2M (before: search 5.2 s, prune 3.9 s, backlink 2.5 s) and the C3 build.
mkdir -p gpurun_out/r3_v && export TMPDIR=/tmp
K=gpurun_out/r3_v
timeout 600 python -m pytest tests -m gpu -q -k "retain or builder or build_score" > $K/pytest.log 2>&1; echo "pytest rc=$?" >> $K/summary.txt; tail -2 $K/pytest.log >> $K/summary.txt
timeout 200 python scripts/fuzz_build.py 45 5 > $K/fuzz_build.log 2>&1; echo "fuzz_build rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_build.log >> $K/summary.txt
timeout 900 python bench.py --workload c5 --n 2000000 --no-cpu-baseline > $K/c5_2m.json 2> $K/c5_2m.err; echo "c5 rc=$?" >> $K/summary.txt
grep -a "\[build\]" $K/c5_2m.err | tail -1 >> $K/summary.txt
timeout 900 python bench.py --no-flat --no-cpu-baseline --steps 3 > $K/c3.json 2> $K/c3.err; echo "c3 rc=$?" >> $K/summary.txt
grep -a "\[build\] {" $K/c3.err | tail -1 | cut -c1-400 >> $K/summary.txt
grep -a "evaluate" $K/c3.err | tail -1 >> $K/summary.txt
