#!/bin/bash
# GPU session ZG (closing): the clean-rebuilt library — whole -m gpu suite and smoke()
set -u
O=gpurun_out/r2zg; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
