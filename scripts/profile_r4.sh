#!/bin/bash
# Round-4 profiling recipe (GPU box): the DEFAULT 10M workload under rocprofv3, search steps only (index cached on /tmp by the
# plain default run that precedes it).  Passes, never combined with trace domains:
#   kernel-trace + stats | FETCH_SIZE | WRITE_SIZE | LDS conflicts | TCC hit rate | g1..g8: SQ wait / TA / TCP / TD counters of
#   the traversal kernel (the wishlist is filtered against `rocprofv3 -L` of the box)
set -u
TAG=${1:-r4_10m}; N=${2:-10000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/prof_$TAG; K=$R/gpurun_out/prof_$TAG; C=/tmp/jv_index_$N.npz
mkdir -p $O $K
/opt/rocm/bin/rocminfo > $K/rocminfo.txt 2>&1
ARGS="--n $N --index-cache $C --no-cpu-baseline --no-sub-workloads"
cd /tmp && export TMPDIR=/tmp
[ -f $C ] || timeout 900 python $R/bench.py --n $N --index-cache $C --steps 1 --warmup 1 --no-flat --no-cpu-baseline --cal-queries 256 --eval-queries 256 > $K/cache_build.log 2>&1
extract() { f=$(find $O/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py $ARGS > $K/stats.log 2>&1
cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
f=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/kernel_trace_jv.csv; grep -E "jv::" $f >> $K/kernel_trace_jv.csv; }
SHORT="$ARGS --steps 3 --warmup 1 --no-sub-workloads"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --output-format csv -d $O/$CTR -o bench -- python $R/bench.py $SHORT > $K/$CTR.log 2>&1
  extract $CTR
done
rocprofv3 -L > $K/counters_list.txt 2>&1
python - "$K/counters_list.txt" > $K/groups.txt <<'PY'
import re,sys
have=set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", open(sys.argv[1]).read()))
groups=[["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_INSTS_VMEM_RD","SQ_INSTS_VALU","SQ_INSTS_LDS"],
        ["TCP_TOTAL_ACCESSES_sum","TCP_TCC_READ_REQ_sum","TCP_TOTAL_CACHE_ACCESSES_sum","TCP_TA_TCP_STATE_READ_sum"],
        ["TD_TD_BUSY_sum","TD_TC_STALL_sum","GRBM_GUI_ACTIVE","GRBM_COUNT"]]
for g in groups:
    g=[c for c in g if c in have]
    if g: print(" ".join(g))
PY
i=0
while read -r G; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/g$i -o bench -- python $R/bench.py $SHORT --no-flat > $K/g$i.log 2>&1
  extract g$i
done < $K/groups.txt
# the same three groups with the WORKGROUP form of the traversal forced onto the benched batch (JVECTOR_HIP_GS_WGX=1): what the LDS table
# does to the L2 -> L1 request count per expansion (verdict r3 #1: TCP_TCC_READ_REQ before / after)
i=0
while read -r G; do
  i=$((i+1))
  JVECTOR_HIP_GS_WGX=1 timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/w$i -o bench -- python $R/bench.py $SHORT --no-flat > $K/w$i.log 2>&1
  extract w$i
done < $K/groups.txt
ls -la $K; tail -3 $K/stats.log | cut -c1-600
