#!/bin/bash
# Round 3, GPU session A: first hardware run of the two-tier visited set (LDS tier + global tier) of the traversal kernel.
#   1. the whole -m gpu suite   2. 1M A/B of the LDS tier sizes (bench's env sweep)   3. 10M default bench (index cached on /tmp)
#   4. rocprofv3: kernel stats, FETCH / WRITE, TCC hit rate, and whatever TA / TCP / SQ-wait counters this rocprofv3 knows
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
# ---- 1M A/B
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=10;JVECTOR_HIP_GS_V1_LOG2=11;JVECTOR_HIP_GS_V1_LOG2=12,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_V1_LOG2=0,JVECTOR_HIP_GS_PROF=1" JVECTOR_HIP_GRAPH_TIMING=0 \
  timeout 600 python bench.py --n 1000000 --steps 5 --warmup 2 --no-flat --no-cpu-baseline --queries 16384 > $O/bench_1m.json 2> $O/bench_1m.err
grep -E "sweep|prof|evaluate|calibrate.*0\.9[5-9]" $O/bench_1m.err | tail -20 | tee -a $O/summary.txt
# ---- 10M: cache the index, default bench, sweep
C=/tmp/jv_index_10m.npz
timeout 900 python bench.py --index-cache $C --steps 1 --warmup 1 --no-flat --no-cpu-baseline --cal-queries 256 --eval-queries 256 > $O/cache_build.log 2>&1
ls -la $C >> $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=11;JVECTOR_HIP_GS_V1_LOG2=12,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_V1_LOG2=0,JVECTOR_HIP_GS_PROF=1" \
  timeout 900 python bench.py --index-cache $C --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "sweep|prof|evaluate" $O/bench_10m.err | tail -12 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
for f in ("bench_1m","bench_10m"):
    try:
        d=json.loads([l for l in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3a",f+".json")).read().splitlines() if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["recall_at_10"], d["config"]["rerankK"], d["kernel_ms_per_step"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "no line", e)
PY
# ---- rocprofv3 (search steps only, rerankK pinned to what the default run calibrated)
RK=$(python -c "import json;print(json.loads([l for l in open('$O/bench_10m.json').read().splitlines() if l.startswith('{')][-1])['config']['rerankK'])" 2>/dev/null || echo 100)
ARGS="--index-cache $C --no-cpu-baseline --no-flat --rerank $RK --cal-queries 256 --eval-queries 256 --steps 4 --warmup 1"
P=/tmp/prof_r3a; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
extract() { f=$(find $P/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $O/pmc_$1.csv; grep -E "graph_search|exact_gather" $f >> $O/pmc_$1.csv; }; }
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python $R/bench.py $ARGS > $O/stats.log 2>&1
cp $P/stats/*kernel_stats.csv $O/ 2>/dev/null
f=$(find $P/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { head -1 $f > $O/kernel_trace_jv.csv; grep -E "jv::" $f >> $O/kernel_trace_jv.csv; }
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $CTR --output-format csv -d $P/$CTR -o bench -- python $R/bench.py $ARGS > $O/$CTR.log 2>&1
  extract $CTR
done
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/TCC -o bench -- python $R/bench.py $ARGS > $O/TCC.log 2>&1; extract TCC
# wishlist groups: keep the names this rocprofv3 lists
python - "$O/counters_list.txt" > $O/groups.txt <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
have=set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", txt))
groups=[["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_INSTS_VMEM_RD","SQ_INSTS_VALU","SQ_INSTS_LDS"],
        ["SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_INST_CYCLES_VMEM","SQ_WAIT_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_SALU"],
        ["TA_TA_BUSY_sum","TA_BUSY_avr","TA_BUSY_max"],["TA_ADDR_STALLED_BY_TC_CYCLES_sum","TA_DATA_STALLED_BY_TC_CYCLES_sum"],
        ["TA_FLAT_READ_WAVEFRONTS_sum","TA_BUFFER_WAVEFRONTS_sum","TA_FLAT_WAVEFRONTS_sum"],
        ["TCP_TOTAL_ACCESSES_sum","TCP_TCC_READ_REQ_sum","TCP_TOTAL_CACHE_ACCESSES_sum","TCP_TA_TCP_STATE_READ_sum"],
        ["TCP_PENDING_STALL_CYCLES_sum","TCP_TCP_TA_DATA_STALL_CYCLES_sum","TCP_GATE_EN1_sum","TCP_GATE_EN2_sum"],
        ["TD_TD_BUSY_sum","TD_TC_STALL_sum","GRBM_GUI_ACTIVE","GRBM_COUNT"]]
for g in groups:
    g=[c for c in g if c in have]
    if g: print(" ".join(g))
PY
i=0
while read -r G; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $G --output-format csv -d $P/g$i -o bench -- python $R/bench.py $ARGS > $O/g$i.log 2>&1
  extract g$i
done < $O/groups.txt
ls -la $O | tee -a $O/summary.txt | tail -40
