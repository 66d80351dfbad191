#!/bin/bash
# Round 4, GPU session H: the abort of session G's -m gpu run (inside ProductQuantization.compute with 16 clusters, after 211 tests):
# the training tests alone, then the whole suite verbosely
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_pq_train_gpu.py -m gpu -v > $O/pytest_train.log 2>&1; echo "pytest train rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_train.log | tee -a $O/summary.txt
timeout 1200 python -X faulthandler -m pytest tests -m gpu -v > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; grep -E "FAILED|ERROR|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -12 | tee -a $O/summary.txt
dmesg 2>/dev/null | tail -5 >> $O/summary.txt
