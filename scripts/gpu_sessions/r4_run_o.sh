#!/bin/bash
# Round 4, GPU session O: design study — how many scored neighbours could be dropped behind a quantized upper-bound table (scripts/ub8_study.py)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4o; mkdir -p $O
cd $R
timeout 900 python scripts/ub8_study.py 10000000 76 > $O/ub8_study.log 2>&1; echo "rc=$?" | tee -a $O/summary.txt; tail -14 $O/ub8_study.log | tee -a $O/summary.txt
