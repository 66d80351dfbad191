#!/bin/bash
# Session AK (round 3, closing): the whole -m gpu suite on the final library.
mkdir -p gpurun_out/r3_ak && export TMPDIR=/tmp
K=gpurun_out/r3_ak
timeout 100 python -m pytest tests -m gpu -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_gpu.log | tail -1 >> $K/summary.txt
grep -a "FAILED\|^E " $K/pytest_gpu.log | head -10 >> $K/summary.txt
