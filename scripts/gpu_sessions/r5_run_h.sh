#!/bin/bash
# Round 5, GPU session H: the round's rocprofv3 record — scripts/profile_r5.sh (headline: kernel trace + stats, FETCH / WRITE, SQ / TCP / TD
# groups of the UBR traversal and its table kernel; BASELINE config 5 at 4M x 1536: the same groups + TCC for the robust prune)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
bash $R/scripts/profile_r5.sh r5_10m 10000000 4000000 2>&1 | tail -50
