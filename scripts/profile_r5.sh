#!/bin/bash
# Round-5 profiling recipe (GPU box): the DEFAULT 10M workload under rocprofv3, search steps only (index cached on /tmp by the plain run
# that precedes it) — the traversal is the register-table bound form now.  Passes, never combined with trace domains:
#   kernel-trace + stats | FETCH_SIZE | WRITE_SIZE | SQ / TCP / TD groups of the traversal and its table kernel
# then BASELINE config 5 (--workload c5, 4M x 1536: the same kernels and per-node work as 10M, a third of the time) under the same
# groups + TCC hit/miss for retain_diverse_kernel and the builder's search kernel.
set -u
TAG=${1:-r5_10m}; N=${2:-10000000}; N5=${3:-4000000}
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
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS"
G2="TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
G3="TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT"
G4="TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/g$i -o bench -- python $R/bench.py $SHORT > $K/g$i.log 2>&1
  extract g$i
done
# ---- BASELINE config 5: the robust prune and the builder's search under the same groups ----
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $G --output-format csv -d $O/c5_g$i -o bench -- python $R/bench.py --workload c5 --n $N5 --no-cpu-baseline --eval-queries 256 > $K/c5_g$i.log 2>&1
  extract c5_g$i
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o bench -- python $R/bench.py --workload c5 --n $N5 --no-cpu-baseline --eval-queries 256 > $K/c5_stats.log 2>&1
cp $O/c5_stats/*kernel_stats.csv $K/c5_kernel_stats.csv 2>/dev/null
ls -la $K | head -40; tail -2 $K/stats.log | cut -c1-400; tail -2 $K/c5_stats.log | cut -c1-400
