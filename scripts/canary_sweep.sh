#!/bin/bash
# A/B of the device-traversal kernel variants with the standalone canary (seconds per point, no Python): run on the GPU box,
#   gpurun --timeout 600 -- 'bash scripts/canary_sweep.sh'
# Each line is the canary's JSON (device_qps / host_qps on a random 2M-node, degree-32 graph, 8192 queries, rerankK 150).
set -u
mkdir -p gpurun_out/gs build
g++ -std=c++17 -O2 tools/gs_canary.cpp -o build/gs_canary -Ljvector_amd -ljvector_hip -Wl,-rpath,"$PWD/jvector_amd" || exit 1
N=${N:-2000000}; Q=${Q:-8192}; RK=${RK:-150}
# JVECTOR_HIP_GS_VCAP_LOG2=16: a random digraph has no neighbourhood overlap (~32 new nodes per expansion), so the default
# visited table (sized from real-graph statistics) would overflow for a share of the queries and blur the comparison
run() { echo "## $*"; env JVECTOR_HIP_GS_VCAP_LOG2=16 JVECTOR_HIP_GRAPH_TIMING=1 "$@" timeout 150 build/gs_canary $N $Q 32 $RK 3 2 2>&1 | tail -3; }
{
run JVECTOR_HIP_GS_OCC=2 JVECTOR_HIP_GS_PAIR=1
run JVECTOR_HIP_GS_OCC=2 JVECTOR_HIP_GS_PAIR=0
run JVECTOR_HIP_GS_OCC=4
run JVECTOR_HIP_GS_OCC=2 JVECTOR_HIP_GS_PAIR=1 JVECTOR_HIP_GS_CAND_CAP=512
run JVECTOR_HIP_GS_OCC=2 JVECTOR_HIP_GS_PAIR=1 JVECTOR_HIP_GS_WAVES_PER_CU=4
run JVECTOR_HIP_GS_OCC=4 JVECTOR_HIP_GS_WAVES_PER_CU=12
} | tee gpurun_out/gs/canary_sweep.log
