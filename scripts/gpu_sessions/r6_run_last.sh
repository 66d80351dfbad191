#!/bin/bash
# Round 6, LAST closing session (head of the round): the tree as committed — the whole -m gpu suite, smoke(), the driver's command.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6last; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1
echo "smoke rc=$?" | tee -a $O/summary.txt
tail -1 $O/smoke.txt | tee -a $O/summary.txt
t0=$(date +%s)
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err
echo "bench rc=$? wall $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
tail -1 $O/bench_default.out | cut -c1-4200 | tee -a $O/summary.txt
grep -E "sub-run|evaluate|\[build\]" $O/bench_default.err | cut -c1-300 | awk '!seen[$0]++' | tee -a $O/summary.txt
cp bench_full.json $O/ 2>/dev/null
