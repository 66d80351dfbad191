#!/bin/bash
# Session K (round 3): whole -m gpu suite on the library with the reranker indirection; NVQ kernels under rocprofv3 (kernel stats,
# FETCH_SIZE, VALU counters — separate passes); nvq_bench with its CPU leg; the headline graph with a second build pass.
mkdir -p gpurun_out/r3_k && export TMPDIR=/tmp
R=$PWD; K=$R/gpurun_out/r3_k
timeout 900 python -m pytest tests -m gpu -x -q > $K/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $K/summary.txt
tail -4 $K/pytest_gpu.log >> $K/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $K/smoke.log 2>&1; echo "smoke rc=$?" >> $K/summary.txt
timeout 600 python scripts/nvq_bench.py 2000000 768 2 16384 95 > $K/nvq_bench.json 2> $K/nvq_bench.err; echo "nvq_bench rc=$?" >> $K/summary.txt
cat $K/nvq_bench.json >> $K/summary.txt
cd /tmp
export NVQ_BENCH_CPU=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nvqprof/stats -o nvq -- python $R/scripts/nvq_bench.py 2000000 768 2 16384 95 > $K/prof_stats.log 2>&1
cp /tmp/nvqprof/stats/*kernel_stats.csv $K/nvq_kernel_stats.csv 2>/dev/null
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  T=$(echo $G | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $G --output-format csv -d /tmp/nvqprof/$T -o nvq -- python $R/scripts/nvq_bench.py 500000 768 2 16384 95 > $K/prof_$T.log 2>&1
  f=$(find /tmp/nvqprof/$T -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/pmc_$T.csv; grep -E "jv::nvq" $f >> $K/pmc_$T.csv; }
done
unset NVQ_BENCH_CPU
cd $R
timeout 900 python bench.py --build-passes 2 --no-flat --no-cpu-baseline > $K/bench_passes2.json 2> $K/bench_passes2.err; echo "bench_passes2 rc=$?" >> $K/summary.txt
grep -a "\[build\]\|evaluate\|calibrate" $K/bench_passes2.err | tail -14 >> $K/summary.txt
python - <<'PY' >> $K/summary.txt
import json
try:
    l = json.loads(open("gpurun_out/r3_k/bench_passes2.json").read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("value", "ms_per_step", "recall_at_10", "graph_build_s")}, l["config"]["rerankK"])
except Exception as e:
    print("bench line unreadable", e)
PY
