#!/bin/bash
# First hardware run of the device-resident graph traversal (k_gsearch.hip).  Run through gpurun, e.g.
#   gpurun --timeout 1500 -- 'bash scripts/validate_device_traversal.sh'
# Steps are ordered cheapest-first and each is bounded by its own timeout so that a hang costs minutes, not the box.
set -u
mkdir -p gpurun_out/gs
# device + host identification next to every number this run produces (record rocminfo beside the roofline peak)
{ /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Name: +gfx" | sort | uniq -c; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Flags" | sed 's/Flags:.*avx512f.*/Flags: ... avx512f .../'; nproc; } > gpurun_out/gs/machine.txt 2>&1
export JVECTOR_TEST_DEVICE_TRAVERSAL=1 JVECTOR_TEST_BUILD_SCORE=1 JVECTOR_TEST_ANISOTROPIC=1 JVECTOR_TEST_PQ_TRAIN=1 JVECTOR_TEST_DENSE=1
# 0a. standalone canary (no Python / torch start-up): device traversal == host traversal on a random 200k-node graph, timed
mkdir -p build && g++ -std=c++17 -O2 tools/gs_canary.cpp -o build/gs_canary -Ljvector_amd -ljvector_hip -Wl,-rpath,"$PWD/jvector_amd" > gpurun_out/gs/canary_build.log 2>&1  # (not `make canary`: no rebuild of the .so on the box)
# (a random digraph has no neighbourhood overlap, so a search marks ~32 new nodes per expansion: give the visited table room,
#  otherwise ~1 query in 5 overflows it and is re-run on the host, which is correct but blurs the timing)
for v in 2 0 1; do JVECTOR_HIP_GRAPH_TIMING=1 JVECTOR_HIP_GS_VCAP_LOG2=15 timeout 120 build/gs_canary 200000 2048 32 100 3 $v 2>&1 | tail -4 | tee -a gpurun_out/gs/canary.log; done
if ! grep -q '"identical": true' gpurun_out/gs/canary.log; then echo "standalone canary failed or hung: stopping"; exit 1; fi
# 0. MFMA dense scan + build-time scoring kernels (flat launches, no persistent loops: cannot hang)
timeout 300 python -m pytest tests/test_zz_exact_dense_gpu.py tests/test_zz_build_score_gpu.py tests/test_zz_anisotropic_gpu.py tests/test_zz_pq_train_gpu.py -q 2>&1 | tail -8 | tee gpurun_out/gs/pytest_bs.log
# 0b. canary: the smallest traversal case under a short timeout, so a hang in the new kernel costs 3 minutes, not 10
timeout 180 python -m pytest tests/test_zz_device_traversal_gpu.py -x -q -k "test_device_traversal_matches_oracle and 1-False-128-16" 2>&1 | tail -5 | tee gpurun_out/gs/pytest_canary.log
if ! grep -q " passed" gpurun_out/gs/pytest_canary.log; then echo "canary failed: stopping"; exit 1; fi
# 1. parity on small graphs (5 shapes x 3 similarity functions, partitions/spills, refusal of unsupported shapes)
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/gs/pytest.log
# 2. 1M-vector bench, both traversals, same index (about a minute each)
for t in host device; do
  JVECTOR_HIP_GRAPH_TIMING=1 timeout 600 python bench.py --n 1000000 --steps 5 --warmup 1 --no-flat --no-cpu-baseline \
      --traversal $t > gpurun_out/gs/bench_1m_$t.json 2> gpurun_out/gs/bench_1m_$t.err
  tail -c 600 gpurun_out/gs/bench_1m_$t.err; head -c 400 gpurun_out/gs/bench_1m_$t.json; echo
done
# 2b. register-allocation variant: 4 waves/SIMD (128 VGPRs, 16 resident queries per CU)
JVECTOR_HIP_GS_OCC=4 JVECTOR_HIP_GRAPH_TIMING=1 timeout 600 python bench.py --n 1000000 --steps 5 --warmup 1 --no-flat --no-cpu-baseline \
    --traversal device > gpurun_out/gs/bench_1m_device_occ4.json 2> gpurun_out/gs/bench_1m_device_occ4.err
tail -c 400 gpurun_out/gs/bench_1m_device_occ4.err; head -c 300 gpurun_out/gs/bench_1m_device_occ4.json; echo
# 2c. one lane per neighbour instead of the pair-lane scoring
JVECTOR_HIP_GS_PAIR=0 JVECTOR_HIP_GRAPH_TIMING=1 timeout 600 python bench.py --n 1000000 --steps 5 --warmup 1 --no-flat --no-cpu-baseline \
    --traversal device > gpurun_out/gs/bench_1m_device_nopair.json 2> gpurun_out/gs/bench_1m_device_nopair.err
tail -c 400 gpurun_out/gs/bench_1m_device_nopair.err; head -c 300 gpurun_out/gs/bench_1m_device_nopair.json; echo
# 3. the headline configuration with the device traversal
if [ "${GS_FULL:-1}" = "1" ]; then
  JVECTOR_HIP_GRAPH_TIMING=1 timeout 900 python bench.py --traversal device > gpurun_out/gs/bench_10m_device.json \
      2> gpurun_out/gs/bench_10m_device.err
  tail -c 800 gpurun_out/gs/bench_10m_device.err; cat gpurun_out/gs/bench_10m_device.json
fi
# 4. per-kernel micro-benchmarks incl. the measured HBM copy / triad ceilings of this box
timeout 400 python scripts/microbench.py --out gpurun_out/gs/microbench.json > gpurun_out/gs/microbench.log 2>&1; tail -c 600 gpurun_out/gs/microbench.log
