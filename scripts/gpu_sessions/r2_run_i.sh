#!/bin/bash
# round-2 GPU session I: the default bench (engine-built Vamana graph at 10M) + its rocprofv3 profile
set -u
O=gpurun_out/r2i; mkdir -p $O
JVECTOR_HIP_GRAPH_TIMING=1 timeout 2400 python bench.py --index-cache /tmp/jv_index_10000000.npz > $O/bench_default.json 2> $O/bench_default.err
grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_default.err | tail -14 | cut -c1-500; grep -c "overflow=[1-9]" $O/bench_default.err; head -c 1500 $O/bench_default.json; echo
bash scripts/profile_r2.sh r2_10m_v3 10000000 2>&1 | tail -4 | cut -c1-300
