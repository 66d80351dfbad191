#!/bin/bash
# Round 6, GPU session G: the scoring pass as a function (gs_ubr_pass, eight lanes per survivor always) — same speed as the inline loop
# of session D (49.0 ms)?  Then the driver's command with the new sub-runs (reference_order, literal_c3 at 10M, hard_case with a CPU leg):
# wall time and the compact line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
t0=$(date +%s)
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$? wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
tail -1 $O/bench_default.out | cut -c1-4200 | tee -a $O/summary.txt
grep -E "sub-run|evaluate|\[build\]" $O/bench_default.err | cut -c1-300 | tee -a $O/summary.txt
cp bench_full.json $O/ 2>/dev/null
