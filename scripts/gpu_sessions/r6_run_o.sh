#!/bin/bash
# Round 6, GPU session O: the register-table bound form for EUCLIDEAN searches (lower bucket edges) — parity (tables == restatement,
# searches == oracle, builder searches), its speed against the plain pair form at 1M x 768 L2; then the hardware fuzzers on the round's
# library (traversal, kernels, searcher objects).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6o; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py tests/test_builder.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 900 python scripts/l2_ubr_bench.py > $O/l2_ubr.json 2> $O/l2_ubr.err
echo "l2 bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/l2_ubr.json | cut -c1-900 | tee -a $O/summary.txt
for f in fuzz_traversal fuzz_kernels fuzz_searcher; do
  timeout 420 python scripts/$f.py 300 6 > $O/$f.log 2>&1
  echo "$f rc=$?" | tee -a $O/summary.txt
  tail -2 $O/$f.log | cut -c1-300 | tee -a $O/summary.txt
done
