#!/bin/bash
# Session Q (round 3): NVQ fuzzing again (NaN payloads compared as equal), three seeds.
mkdir -p gpurun_out/r3_q && export TMPDIR=/tmp
K=gpurun_out/r3_q
for s in 11 12 13; do
  timeout 200 python scripts/fuzz_nvq.py 60 $s >> $K/fuzz_nvq.log 2>&1; echo "fuzz_nvq seed $s rc=$?" >> $K/summary.txt; tail -1 $K/fuzz_nvq.log >> $K/summary.txt
done
