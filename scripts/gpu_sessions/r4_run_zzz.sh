#!/bin/bash
# Round 4, the last GPU seconds: parity of the final prune kernel on hardware (after its staging lost a struct temporary that went to
# scratch), then the same builder test through the PHASE-CLOCK kernel (rd_prof = 1): where a prune's time goes, for the next round.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4zzz; mkdir -p $O
cd $R
timeout 30 python -m pytest tests/test_retain_diverse.py tests/test_builder.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_HIP_RD_PROF=1 timeout 25 python -m pytest tests/test_builder.py -m gpu -x -q -k "test_builder_gpu" -s > $O/prof.txt 2>&1
echo "prof rc=$?" | tee -a $O/summary.txt
grep "rd prof" $O/prof.txt | sort -t= -k2 -n | tail -25 | cut -c1-330 | tee -a $O/summary.txt
