#!/bin/bash
# Round 5, GPU session E: SQ / TCP / TD counters of the UBR traversal kernel (and the table kernel) at the headline shape — is the form
# bound by VALU issue now that the gathers are a third of what they were?
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
K=$R/gpurun_out/r5e; O=/tmp/prof_r5e; C=/tmp/jv_index_10m.npz
mkdir -p $K $O
cd /tmp && export TMPDIR=/tmp
export JVECTOR_HIP_GS_UBR=1
timeout 900 python $R/bench.py --index-cache $C --steps 1 --warmup 1 --no-flat --no-cpu-baseline --no-sub-workloads --cal-queries 256 --eval-queries 256 > $K/cache_build.log 2>&1
SHORT="--index-cache $C --no-cpu-baseline --no-sub-workloads --no-flat --steps 3 --warmup 1 --rerank 76 --cal-queries 256 --eval-queries 256"
extract() { f=$(find $O/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $O/g$i -o bench -- python $R/bench.py $SHORT > $K/g$i.log 2>&1
  extract g$i
done
python - <<'PY' | tee $K/summary.txt
import csv,glob,os,collections
K=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/r5e")
for f in sorted(glob.glob(K+"/g*_jv.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in agg.items():
        if "graph_search" in k or "ubr_table" in k:
            print(os.path.basename(f), k, {a: f"{b:.4g}" for a,b in v.items()})
PY
