#!/bin/bash
# Session S (round 3): where does the C5 build (1536 dimensions, PQ-192) spend its kernel time?  rocprofv3 kernel stats at 2M nodes.
mkdir -p gpurun_out/r3_s && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_s
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5prof -o c5 -- python $R/bench.py --workload c5 --n 2000000 --no-cpu-baseline > $K/c5_2m.json 2> $K/c5_2m.err; echo "c5 rc=$?" >> $K/summary.txt
cp /tmp/c5prof/*kernel_stats.csv $K/c5_kernel_stats.csv 2>/dev/null
grep -a "\[build\]" $K/c5_2m.err | tail -3 >> $K/summary.txt
head -25 $K/c5_kernel_stats.csv | cut -c1-200 >> $K/summary.txt
