#!/bin/bash
# Round 6, GPU session I: the profiling recipe of the round (scripts/profile_r6.sh): kernel trace + stats and PMC groups of the headline,
# FETCH / WRITE / LDS-conflict counters of the two-stage flat filter at C2, one C4 shard and the headline's flat_mode.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
bash $R/scripts/profile_r6.sh r6_10m 10000000 > $R/gpurun_out/prof_r6.log 2>&1
tail -30 $R/gpurun_out/prof_r6.log
