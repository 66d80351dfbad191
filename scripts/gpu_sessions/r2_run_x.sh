#!/bin/bash
set -u
O=gpurun_out/r2x; mkdir -p $O
timeout 900 python scripts/lut_cache_study.py 1000000 110 2>&1 | grep -v "^\[" | tail -8 | tee $O/lut_cache_1m.log
timeout 1200 python scripts/lut_cache_study.py 10000000 110 2>&1 | grep -v "^\[" | tail -8 | tee $O/lut_cache_10m.log
