#!/bin/bash
# Session AJ (round 3, last GPU minute): PQ training / refine with fewer than 256 clusters and the PQ entry points on the MI355X
# (the host side changed: pq_create_impl, unpadded training work objects), + smoke.
mkdir -p gpurun_out/r3_aj && export TMPDIR=/tmp
K=gpurun_out/r3_aj
timeout 100 python -m pytest tests/test_zz_pq_train_gpu.py tests/test_gpu_parity.py -m gpu -q > $K/pytest_pq.log 2>&1; echo "pytest_pq rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_pq.log | tail -1 >> $K/summary.txt
grep -a "FAILED\|^E " $K/pytest_pq.log | head -10 >> $K/summary.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
