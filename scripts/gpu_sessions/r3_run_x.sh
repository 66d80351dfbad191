#!/bin/bash
# Session X (round 3): where does a threshold search of GraphSearcher objects spend its time (kernel / log replay / rerank stage)?
mkdir -p gpurun_out/r3_x && export TMPDIR=/tmp
K=gpurun_out/r3_x
SEARCHER_BENCH_TIMING=1 timeout 600 python scripts/searcher_bench.py > $K/searcher_bench.json 2> $K/searcher_bench.err; echo "rc=$?" >> $K/summary.txt
grep -a "searcher objects\]" $K/searcher_bench.err | sort | uniq -c | sort -rn | head -20 >> $K/summary.txt
cat $K/searcher_bench.json >> $K/summary.txt
