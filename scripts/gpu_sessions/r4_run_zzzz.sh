#!/bin/bash
# Round 4, the very last GPU seconds: the SQUARE pair table (rd_square = 1) — parity through the C ABI, then the builder test through
# the phase-clock kernel with it (compare "sums" per test with profiles/r4_zzz: 7 100 clocks with the triangular table)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4zzzz; mkdir -p $O
cd $R
timeout 14 python -m pytest tests/test_retain_diverse.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_HIP_RD_PROF=1 JVECTOR_HIP_RD_SQUARE=1 timeout 14 python -m pytest tests/test_builder.py -m gpu -x -q -k "test_builder_gpu" -s > $O/prof.txt 2>&1
echo "prof rc=$?" | tee -a $O/summary.txt
grep "rd prof" $O/prof.txt | sort -t= -k2 -n | tail -16 | cut -c1-330 | tee -a $O/summary.txt
