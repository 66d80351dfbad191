#!/bin/bash
# Session AG (round 3): quantizers with fewer than 256 clusters (padded codebooks) on the MI355X — the whole -m gpu suite on the
# final library (new: test_cluster_counts_below_256, test_graph_search_with_fewer_than_256_clusters) + smoke.
mkdir -p gpurun_out/r3_ag && export TMPDIR=/tmp
K=gpurun_out/r3_ag
timeout 900 python -m pytest tests -m gpu -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt; grep -a "passed\|failed" $K/pytest_gpu.log | tail -1 >> $K/summary.txt
grep -a "FAILED\|^E " $K/pytest_gpu.log | head -20 >> $K/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
