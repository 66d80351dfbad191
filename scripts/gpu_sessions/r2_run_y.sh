#!/bin/bash
set -u
O=gpurun_out/r2y; mkdir -p $O
timeout 400 python scripts/fuzz_traversal.py 150 1 2>&1 | grep -v amdgpu | tail -12 | tee $O/fuzz_seed1.log
timeout 400 python scripts/fuzz_traversal.py 150 2 2>&1 | grep -v amdgpu | tail -12 | tee $O/fuzz_seed2.log
