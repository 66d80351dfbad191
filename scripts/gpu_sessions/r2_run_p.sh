#!/bin/bash
set -u
O=gpurun_out/r2p; mkdir -p $O
JVECTOR_HIP_GS_ADAPT=1 timeout 600 python -m pytest tests/test_graph_search.py tests/test_zz_device_traversal_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_ADAPT=1;JVECTOR_HIP_GS_ADAPT=0;JVECTOR_HIP_GS_ADAPT=1,JVECTOR_HIP_GS_PROF=1" timeout 900 python bench.py --n 1000000 --steps 4 --warmup 1 --no-flat --no-cpu-baseline 2>&1 >/dev/null | grep -E "sweep|gs prof" | tail -5 | cut -c1-300
