#!/bin/bash
# Round 5, GPU session V: hardware fuzzers on the final library — the traversal (with the register-table bound forms, over the row and
# over the compacted list, switched on / off and their trim interval in the draw), GraphSearcher objects, the batched kernels
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5v; mkdir -p $O
cd $R
timeout 400 python scripts/fuzz_traversal.py 240 11 > $O/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz_traversal.log | tee -a $O/summary.txt
timeout 200 python scripts/fuzz_searcher.py 70 5 > $O/fuzz_searcher.log 2>&1; echo "fuzz_searcher rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz_searcher.log | tee -a $O/summary.txt
timeout 200 python scripts/fuzz_kernels.py 70 5 > $O/fuzz_kernels.log 2>&1; echo "fuzz_kernels rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz_kernels.log | tee -a $O/summary.txt
timeout 200 python scripts/fuzz_build.py 60 5 > $O/fuzz_build.log 2>&1; echo "fuzz_build rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz_build.log | tee -a $O/summary.txt
