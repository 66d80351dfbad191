#!/bin/bash
# round-2 GPU session E: traversal-kernel variants (A/B in one process at 1M), retain_diverse parity + throughput
set -u
O=gpurun_out/r2e; mkdir -p $O
timeout 600 python -m pytest tests/test_retain_diverse.py tests/test_zz_device_traversal_gpu.py tests/test_zz_build_score_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.log
JVECTOR_HIP_GRAPH_TIMING=1 JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_OCC=3;JVECTOR_HIP_GS_OCC=3,JVECTOR_HIP_GS_WAVES_PER_CU=10;JVECTOR_HIP_GS_CAND_CAP=768;JVECTOR_HIP_GS_CAND_CAP=256;JVECTOR_HIP_GS_VCAP_LOG2=13,JVECTOR_HIP_GS_RETRY=1;JVECTOR_HIP_GS_OCC=3,JVECTOR_HIP_GS_VCAP_LOG2=13,JVECTOR_HIP_GS_RETRY=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 600 python bench.py --n 1000000 --steps 5 --warmup 1 --no-flat --no-cpu-baseline --rerank 125 --cal-queries 1024 --eval-queries 1024 > $O/bench_1m.json 2> $O/bench_1m.err
grep -E "sweep|gs prof|evaluate|Error|error|device\] Q=16384" $O/bench_1m.err | sort | uniq -c | sort -rn | head -30
