#!/bin/bash
# Session AI (round 3, last GPU minutes): the C5 line (10M x 1536 build) on the FINAL library — its searches run on the M = 192
# one-lane-per-neighbour kernel, which was recompiled around the wide-row chunk loop.
mkdir -p gpurun_out/r3_ai && export TMPDIR=/tmp
K=gpurun_out/r3_ai
timeout 225 python bench.py --workload c5 --n 10000000 --no-cpu-baseline > $K/c5.json 2> $K/c5.err; echo "c5 rc=$?" >> $K/summary.txt; tail -1 $K/c5.json | cut -c1-400 >> $K/summary.txt
