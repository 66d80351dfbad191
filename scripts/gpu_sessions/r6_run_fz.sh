#!/bin/bash
# Round 6, GPU session FZ: the hardware fuzzers on the library with deferred scores (gs_defer): random problems (1 - 3 layers, PQ-96 and
# others, every similarity, filters, ties, tiny tables) through the device traversal and the searcher objects against the oracle —
# gs_defer on / off and from level 1 / 2 among the randomised knobs.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6fz; mkdir -p $O
cd $R
for spec in "fuzz_traversal 420 11" "fuzz_traversal 420 12" "fuzz_searcher 240 7" "fuzz_kernels 180 7"; do
  set -- $spec
  timeout $(( $2 + 120 )) python scripts/$1.py $2 $3 > $O/$1_$3.log 2>&1
  echo "$1 seed $3 rc=$?" | tee -a $O/summary.txt
  tail -3 $O/$1_$3.log | cut -c1-400 | tee -a $O/summary.txt
done
