#!/bin/bash
# Session AH (round 3, last GPU minutes): the C2 line and the one-shard C4 line on the FINAL library.
mkdir -p gpurun_out/r3_ah && export TMPDIR=/tmp
K=gpurun_out/r3_ah
timeout 110 python bench.py --workload c2 > $K/c2.json 2> $K/c2.err; echo "c2 rc=$?" >> $K/summary.txt; tail -1 $K/c2.json | cut -c1-300 >> $K/summary.txt
timeout 150 python bench.py --workload c4 --eval-queries 1024 --no-cpu-baseline > $K/c4_1shard.json 2> $K/c4_1shard.err; echo "c4 rc=$?" >> $K/summary.txt; tail -1 $K/c4_1shard.json | cut -c1-300 >> $K/summary.txt
