#!/bin/bash
# Round 4, GPU session N: differential fuzz of the traversal on hardware with the workgroup form in the draw (forced with random wave
# counts / slots / request depths / partial tables, forbidden, or left to AUTO): every search must equal the oracle bit for bit
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
timeout 400 python scripts/fuzz_traversal.py 200 41 > $O/fuzz_traversal.log 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -3 $O/fuzz_traversal.log | tee -a $O/summary.txt
timeout 300 python scripts/fuzz_build.py 60 7 > $O/fuzz_build.log 2>&1; echo "fuzz build rc=$?" | tee -a $O/summary.txt; tail -2 $O/fuzz_build.log | tee -a $O/summary.txt
