#!/bin/bash
# Session Z (round 3): session kernels for every M the traversal is built for (32, 48, 64, 128, 192 added): parity on hardware.
mkdir -p gpurun_out/r3_z && export TMPDIR=/tmp
K=gpurun_out/r3_z
timeout 900 python -m pytest tests/test_graph_search.py -m gpu -q > $K/pytest.log 2>&1; echo "pytest rc=$?" >> $K/summary.txt; tail -3 $K/pytest.log >> $K/summary.txt
