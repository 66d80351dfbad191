#!/bin/bash
# round-2 GPU session G: the whole -m gpu suite, the default bench line, then the rocprofv3 profile of the same run
set -u
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
JVECTOR_HIP_GRAPH_TIMING=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
grep -E "calibrate|evaluate|Error|error" $O/bench_default.err | tail; grep -c "overflow=[1-9]" $O/bench_default.err; head -c 900 $O/bench_default.json; echo
bash scripts/profile_r2.sh r2_10m_v2 10000000 2>&1 | tail -5 | cut -c1-300
