#!/bin/bash
set -u
O=gpurun_out/r2z; mkdir -p $O
timeout 400 python scripts/fuzz_kernels.py 120 1 2>&1 | grep -v amdgpu | tail -12 | tee $O/fuzz_kernels_seed1.log
timeout 400 python scripts/fuzz_kernels.py 120 2 2>&1 | grep -v amdgpu | tail -12 | tee $O/fuzz_kernels_seed2.log
