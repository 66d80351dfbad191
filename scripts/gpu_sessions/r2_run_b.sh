#!/bin/bash
# round-2 GPU session B: new tests, phase profile of the traversal kernel at 1M, C2 line, the default 10M bench
set -u
O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5 | tee $O/pytest.log
JVECTOR_HIP_GRAPH_TIMING=1 JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_WAVES_PER_CU=4;JVECTOR_HIP_GS_WAVES_PER_CU=6;JVECTOR_HIP_GS_CAND_CAP=256;JVECTOR_HIP_GS_CAND_CAP=512" \
  timeout 600 python bench.py --n 1000000 --steps 5 --warmup 1 --no-flat --no-cpu-baseline > $O/bench_1m.json 2> $O/bench_1m.err
grep -E "sweep|gs prof|calibrate|evaluate|Error|error" $O/bench_1m.err | tail -30; head -c 1500 $O/bench_1m.json; echo
timeout 600 python bench.py --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -5 $O/bench_c2.err; cat $O/bench_c2.json; echo
timeout 1200 python bench.py > $O/bench_10m.json 2> $O/bench_10m.err; tail -12 $O/bench_10m.err; cat $O/bench_10m.json
