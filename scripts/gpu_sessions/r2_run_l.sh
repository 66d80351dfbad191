#!/bin/bash
# round-2 GPU session L (final state): whole -m gpu suite, smoke, default bench, profile of the same run, C5 line
set -u
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
JVECTOR_HIP_GRAPH_TIMING=1 timeout 2400 python bench.py --index-cache /tmp/jv_index_10000000.npz > $O/bench_default.json 2> $O/bench_default.err
grep -E "\[build\] \{|calibrate|evaluate|Error|error|Traceback" $O/bench_default.err | tail -16 | cut -c1-400; head -c 600 $O/bench_default.json; echo
bash scripts/profile_r2.sh r2_10m_v4 10000000 2>&1 | tail -3 | cut -c1-200
timeout 1200 python bench.py --workload c5 --n 1000000 > $O/bench_c5_1m.json 2> $O/bench_c5_1m.err; tail -2 $O/bench_c5_1m.err | cut -c1-300; head -c 400 $O/bench_c5_1m.json; echo
