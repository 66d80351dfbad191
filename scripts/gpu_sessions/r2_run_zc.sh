#!/bin/bash
# GPU session ZC: refresh the secondary workload lines with the final code (C2 flat search, C4 one shard)
set -u
O=gpurun_out/r2zc; mkdir -p $O
timeout 900 python bench.py --workload c2 > $O/c2_bench_line.json 2> $O/c2.err
tail -c 400 $O/c2_bench_line.json; echo
timeout 900 python bench.py --workload c4 > $O/c4_bench_line.json 2> $O/c4.err
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/c4_bench_line.json | tail -c 500; echo
