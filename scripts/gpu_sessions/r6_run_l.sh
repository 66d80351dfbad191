#!/bin/bash
# Round 6, GPU session L: (1) trims of the candidate tier are cheap now (1.9 k instead of 8.7 k clocks): does trimming more often —
# a tighter pop threshold earlier, a shorter tier to scan — pay?  gs_ubr_trim 16 / 24 / 32 / 48 (default) / 64.  (2) C2's flat-filter
# counters again with the block order it now gets (plain: 64 query groups).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_UBR_TRIM=16;JVECTOR_HIP_GS_UBR_TRIM=24;JVECTOR_HIP_GS_UBR_TRIM=32;JVECTOR_HIP_GS_UBR_TRIM=64;JVECTOR_HIP_GS_UBR_TRIM=96;JVECTOR_HIP_GS_UBR_TRIM=48" \
  timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|evaluate" $O/bench.err | cut -c1-300 | tee -a $O/summary.txt
K=$R/gpurun_out/prof_r6_10m; OO=/tmp/prof_r6l
mkdir -p $K $OO
cd /tmp && export TMPDIR=/tmp
extract() { f=$(find $OO/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
A="--workload c2 --no-cpu-baseline --steps 5 --warmup 1"
for CTR in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  T=c2_$(echo $CTR | cut -d' ' -f1)
  timeout 700 rocprofv3 --pmc $CTR --output-format csv -d $OO/$T -o bench -- python $R/bench.py $A > $K/$T.log 2>&1
  extract $T
done
timeout 700 rocprofv3 --kernel-trace --output-format csv -d $OO/c2_trace -o bench -- python $R/bench.py $A > $K/c2_trace.log 2>&1
f=$(find $OO/c2_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/c2_kernel_trace_jv.csv; grep -E "jv::adc|jv::topk|jv::exact" $f >> $K/c2_kernel_trace_jv.csv; }
