#!/bin/bash
# Session L (round 3): the whole -m gpu suite (no -x) + smoke on the final library.
mkdir -p gpurun_out/r3_l && export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3_l/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> gpurun_out/r3_l/summary.txt
tail -6 gpurun_out/r3_l/pytest_gpu.log >> gpurun_out/r3_l/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_l/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3_l/summary.txt
