#!/bin/bash
# Round 6, GPU session W: the profiling recipe's section A on the library with the fused rerank (kernel trace + stats, FETCH / WRITE,
# SQ / TCP / TD groups of the headline): the traversal kernel now also moves the reranked rows — its counters replace round 6's earlier ones.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SKIP_FLAT=1 bash $R/scripts/profile_r6.sh r6f_10m 10000000 > $R/gpurun_out/prof_r6_w.log 2>&1
tail -5 $R/gpurun_out/prof_r6_w.log
