#!/bin/bash
# Session R (round 3): does overlapping one batch's rerank with another batch's traversal (two contexts = two streams) pay?
mkdir -p gpurun_out/r3_r && export TMPDIR=/tmp
K=gpurun_out/r3_r
JVECTOR_BENCH_IN_FLIGHT=2 timeout 900 python bench.py --no-flat --no-cpu-baseline --steps 12 > $K/bench_if2.json 2> $K/bench_if2.err; echo "if2 rc=$?" >> $K/summary.txt
grep -a "in-flight\|evaluate" $K/bench_if2.err >> $K/summary.txt
