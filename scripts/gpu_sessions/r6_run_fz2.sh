#!/bin/bash
# Round 6, GPU session FZ2: two more seeds of the traversal fuzzer on the final library (gs_defer among the randomised knobs).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6fz2; mkdir -p $O
cd $R
for seed in 21 22; do
  timeout 420 python scripts/fuzz_traversal.py 300 $seed > $O/fuzz_traversal_$seed.log 2>&1
  echo "fuzz_traversal seed $seed rc=$?" | tee -a $O/summary.txt
  tail -2 $O/fuzz_traversal_$seed.log | cut -c1-400 | tee -a $O/summary.txt
done
