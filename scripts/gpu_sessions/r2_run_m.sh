#!/bin/bash
# round-2 GPU session M (final state): whole -m gpu suite, default bench, profile of the same run
set -u
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.log
JVECTOR_HIP_GRAPH_TIMING=1 timeout 2400 python bench.py --index-cache /tmp/jv_index_10000000.npz > $O/bench_default.json 2> $O/bench_default.err
grep -E "evaluate|Error|error|Traceback" $O/bench_default.err | tail -5 | cut -c1-400; grep "Q=65536" $O/bench_default.err | sort | uniq -c | cut -c1-220; head -c 500 $O/bench_default.json; echo
bash scripts/profile_r2.sh r2_10m_v5 10000000 2>&1 | tail -3 | cut -c1-200
