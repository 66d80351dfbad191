#!/bin/bash
# Round-6 profiling recipe (GPU box).  Passes, never combined with trace domains (one --pmc group per run):
#  A. the DEFAULT 10M workload, search steps only (index cached on /tmp by the plain run that precedes it):
#     kernel-trace + stats | FETCH_SIZE | WRITE_SIZE | SQ / TCP / TD groups of the traversal and its table kernel
#  B. the two-stage flat filter (adc_bq_kernel + the survivors' exact stage) at its three benched shapes — BASELINE C2 (1M x 128, PQ-16,
#     1024 queries), one C4 shard (12.5M x 768, PQ-96, 256 queries), the headline's flat_mode (10M x 768, 256 queries):
#     FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE   
set -u
TAG=${1:-r6_10m}; N=${2:-10000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/prof_$TAG; K=$R/gpurun_out/prof_$TAG; C=/tmp/jv_index_$N.npz
mkdir -p $O $K
/opt/rocm/bin/rocminfo > $K/rocminfo.txt 2>&1
ARGS="--n $N --index-cache $C --no-cpu-baseline --no-sub-workloads --no-flat"
cd /tmp && export TMPDIR=/tmp
[ -f $C ] || timeout 900 python $R/bench.py --n $N --index-cache $C --steps 1 --warmup 1 --no-flat --no-cpu-baseline --no-sub-workloads --cal-queries 256 --eval-queries 256 > $K/cache_build.log 2>&1
extract() { f=$(find $O/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py $ARGS > $K/stats.log 2>&1
cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
f=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/kernel_trace_jv.csv; grep -E "jv::" $f >> $K/kernel_trace_jv.csv; }
SHORT="$ARGS --steps 3 --warmup 1 --cal-queries 1024 --eval-queries 1024"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --output-format csv -d $O/$CTR -o bench -- python $R/bench.py $SHORT > $K/$CTR.log 2>&1
  extract $CTR
done
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU"
G2="TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_LDS"
G3="TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/g$i -o bench -- python $R/bench.py $SHORT > $K/g$i.log 2>&1
  extract g$i
done
# ---- B. the flat filter at its three shapes (SKIP_FLAT=1: section A only) ----
[ "${SKIP_FLAT:-0}" = 1 ] && { ls $K | wc -l; exit 0; }
FLATC2="--workload c2 --no-cpu-baseline --steps 5 --warmup 1"
FLATC4="--workload c4 --no-cpu-baseline --steps 3 --warmup 1"
FLATFM="--n $N --index-cache $C --no-cpu-baseline --no-sub-workloads --steps 1 --warmup 1 --cal-queries 256 --eval-queries 256 --rerank 74"
for W in c2 c4 fm; do
  case $W in c2) A="$FLATC2";; c4) A="$FLATC4";; fm) A="$FLATFM";; esac
  for CTR in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    T=${W}_$(echo $CTR | cut -d' ' -f1)
    timeout 700 rocprofv3 --pmc $CTR --output-format csv -d $O/$T -o bench -- python $R/bench.py $A > $K/$T.log 2>&1
    extract $T
  done
  timeout 700 rocprofv3 --kernel-trace --output-format csv -d $O/${W}_trace -o bench -- python $R/bench.py $A > $K/${W}_trace.log 2>&1
  f=$(find $O/${W}_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/${W}_kernel_trace_jv.csv; grep -E "jv::adc|jv::topk|jv::exact" $f >> $K/${W}_kernel_trace_jv.csv; }
done
ls -la $K | head -60; tail -1 $K/stats.log | cut -c1-400
