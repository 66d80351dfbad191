#!/bin/bash
# GPU session Q: full -m gpu suite (searcher objects, heap-order rerank, device tie resolution) and a 1M bench A/B of the
# rerank tie check (on / off / without the push log = ties go to the host searcher)
set -u
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -5
JVECTOR_HIP_GRAPH_TIMING=1 JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_TIE_CHECK=1;JVECTOR_HIP_GS_TIE_CHECK=0;JVECTOR_HIP_GS_PUSH_LOG=0;JVECTOR_HIP_GS_TIE_CHECK=1;JVECTOR_HIP_GS_TIE_CHECK=0" timeout 900 python bench.py --n 1000000 --steps 4 --warmup 1 --no-flat --no-cpu-baseline > $O/bench.out 2> $O/bench.err
grep -E "sweep" $O/bench.err | cut -c1-200
grep -E "rerank ties" $O/bench.err | tail -3 | cut -c1-250
tail -c 1500 $O/bench.out
