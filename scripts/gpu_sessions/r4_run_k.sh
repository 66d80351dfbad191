#!/bin/bash
# Round 4, GPU session K: rocprofv3 passes of the default configuration (index with one improveConnections pass: rerankK 75) —
# kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ / TCP / TD groups for the one-wave kernel and, forced, the workgroup form
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
bash scripts/profile_r4.sh r4_10m > $O/profile.log 2>&1; tail -5 $O/profile.log | cut -c1-300 | tee -a $O/summary.txt
python scripts/summarize_profile_r4.py r4_10m > $O/summarize.log 2>&1; tail -25 $O/summarize.log | cut -c1-250 | tee -a $O/summary.txt
mkdir -p $R/gpurun_out/profiles_r4 && cp $R/profiles/r4_10m_* $R/profiles/traffic_r4.json $R/gpurun_out/profiles_r4/ 2>/dev/null
ls $R/gpurun_out/profiles_r4 | tee -a $O/summary.txt
