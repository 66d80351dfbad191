#!/bin/bash
# Round 6, GPU session R: graph-quality study (scripts/exact_prune_study.py: what an exact-score improve pass would buy at 10M),
# then the -m gpu suite on the library with the push-log count in a scalar register.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6r; mkdir -p $O
cd $R
timeout 1500 python scripts/exact_prune_study.py --n 10000000 --variants base,B,A,AB --out $O/study.json > $O/study.out 2> $O/study.err
echo "study rc=$?" | tee $O/summary.txt
grep '\[study\]' $O/study.err | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
