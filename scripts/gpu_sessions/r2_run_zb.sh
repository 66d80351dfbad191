#!/bin/bash
# GPU session ZB: the whole -m gpu suite three times in a row (flake check)
set -u
O=gpurun_out/r2zb; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_$i.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_$i.log | tail -2; done
