#!/bin/bash
# Round 6, GPU session W2: the profiling recipe's section A on the library with DEFERRED scores above level 1 (gs_defer, default on):
# kernel trace + stats, FETCH / WRITE, SQ / TCP / TD groups of the headline — the traversal kernel's counters as it now is (tag r6g_10m).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SKIP_FLAT=1 bash $R/scripts/profile_r6.sh r6g_10m 10000000 > $R/gpurun_out/prof_r6_w2.log 2>&1
tail -5 $R/gpurun_out/prof_r6_w2.log
