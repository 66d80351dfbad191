#!/bin/bash
# Session O (round 3): MFMA tile scan with two staging buffers of 16 columns and one barrier per chunk; parity, rates, counters.
mkdir -p gpurun_out/r3_o && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_o
timeout 300 python -m pytest tests/test_zz_exact_dense_gpu.py -q > $K/pytest_dense.log 2>&1; echo "pytest_dense(w3) rc=$?" >> $K/summary.txt
JVECTOR_HIP_ED_WAVES=0 timeout 300 python -m pytest tests/test_zz_exact_dense_gpu.py -q >> $K/pytest_dense.log 2>&1; echo "pytest_dense(w0) rc=$?" >> $K/summary.txt
for w in 3 4 0; do
  JVECTOR_HIP_ED_WAVES=$w timeout 300 python scripts/dense_bench.py > $K/dense_w$w.json 2> $K/dense_w$w.err; echo "dense w=$w rc=$?" >> $K/summary.txt
  cat $K/dense_w$w.json >> $K/summary.txt
done
cd /tmp
JVECTOR_HIP_ED_WAVES=3 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/edprof/w3 -o ed -- python $R/scripts/dense_bench.py 1000000 768 > $K/prof_w3.log 2>&1
f=$(find /tmp/edprof/w3 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/pmc_w3.csv; grep -E "exact_dense" $f >> $K/pmc_w3.csv; }
