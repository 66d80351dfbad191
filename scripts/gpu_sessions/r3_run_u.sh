#!/bin/bash
# Session U (round 3): what bounds the robust-prune kernel?  FETCH_SIZE / TCC hit-miss / wait counters of retain_diverse_kernel (C5, 1M nodes).
mkdir -p gpurun_out/r3_u && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_u
cd /tmp
for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  T=$(echo $G | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $G --output-format csv -d /tmp/rdprof/$T -o rd -- python $R/bench.py --workload c5 --n 1000000 --no-cpu-baseline > $K/prof_$T.log 2>&1
  f=$(find /tmp/rdprof/$T -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/pmc_$T.csv; grep -E "retain_diverse" $f >> $K/pmc_$T.csv; }
done
ls -la $K > $K/summary.txt
