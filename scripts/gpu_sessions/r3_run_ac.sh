#!/bin/bash
# Session AC (round 3): differential fuzz of GraphSearcher OBJECTS on the MI355X (scripts/fuzz_searcher.py: threshold / floor /
# acceptOrds searches + random resume chains against the oracle, every session-kernel shape, small capacities so that spills,
# refills, retries and host fallbacks occur) + a last run of the traversal and kernel fuzzers on the final library.
mkdir -p gpurun_out/r3_ac && export TMPDIR=/tmp
K=gpurun_out/r3_ac
for S in 11 12; do
  timeout 400 python scripts/fuzz_searcher.py 110 $S > $K/fuzz_searcher_$S.log 2>&1; echo "fuzz_searcher seed $S rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_searcher_$S.log >> $K/summary.txt
done
timeout 300 python scripts/fuzz_traversal.py 60 21 > $K/fuzz_traversal.log 2>&1; echo "fuzz_traversal rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_traversal.log >> $K/summary.txt
timeout 300 python scripts/fuzz_kernels.py 45 21 > $K/fuzz_kernels.log 2>&1; echo "fuzz_kernels rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_kernels.log >> $K/summary.txt
