#!/bin/bash
# Round 6, GPU session Q: the profiling recipe's section A again on the round's final library (packed rerank remainders, trim 16,
# builder searches through gs_ubr_pass): kernel trace + stats, FETCH / WRITE, SQ / TCP / TD groups of the headline.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SKIP_FLAT=1 bash $R/scripts/profile_r6.sh r6_10m 10000000 > $R/gpurun_out/prof_r6_q.log 2>&1
tail -5 $R/gpurun_out/prof_r6_q.log
