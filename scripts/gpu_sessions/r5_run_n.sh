#!/bin/bash
# Round 5, GPU session N: kernel trace of C2 and of one C4 shard with the two-stage filter (which of its kernels takes the time?)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -o bench -- python $R/bench.py --workload $W --no-cpu-baseline > $O/$W.log 2>&1
  cp /tmp/prof_$W/*kernel_stats.csv $O/${W}_kernel_stats.csv 2>/dev/null
  head -12 $O/${W}_kernel_stats.csv | cut -c1-200
done
