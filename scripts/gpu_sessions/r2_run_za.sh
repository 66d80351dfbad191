#!/bin/bash
set -u
O=gpurun_out/r2za; mkdir -p $O
timeout 600 python scripts/fuzz_build.py 200 1 2>&1 | grep -v amdgpu | tail -15 | tee $O/fuzz_build_seed1.log
