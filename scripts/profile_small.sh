#!/bin/bash
# Bounded profiling recipe (runs on the GPU box via gpurun).  Lessons of r1: NEVER profile the 10M graph-mode setup
# (millions of tiny torch launches in the synthetic index build make rocprofv3 crawl) and never run --pmc on it.
#   flat  : 10M flat mode, kernel-trace stats + FETCH_SIZE + WRITE_SIZE        (the ADC-scan kernel)
#   graph : 1M graph mode, kernel-trace stats only                              (frontier kernel durations)
#   fpmc  : frontier kernel FETCH_SIZE / WRITE_SIZE on a random 1M graph        (scripts/frontier_pmc.py)
#   gsearch : device-resident traversal kernel on the same random 1M graph: kernel-trace stats, then FETCH/WRITE PMC
#   dense : MFMA tile form of the exact scan, 256 x 1M x 768 (scripts/dense_pmc.py): kernel-trace stats, FETCH/WRITE, MFMA busy
set -u
WHAT=${1:-flat}; TAG=${2:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/prof_${TAG}_$WHAT; K=$R/gpurun_out/prof_${TAG}_$WHAT
mkdir -p $O $K
cd /tmp && export TMPDIR=/tmp
extract() { f=$(find $O/$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $K/$1_jv.csv; grep -E "jv::" $f >> $K/$1_jv.csv; }; }
case $WHAT in
flat)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --mode flat --steps 5 --warmup 1 --no-cpu-baseline > $K/stats.log 2>&1
  cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/$C -o bench -- python $R/bench.py --mode flat --steps 2 --warmup 1 --no-cpu-baseline --rerank 400 > $K/$C.log 2>&1
    extract $C
  done ;;
graph)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --mode graph --n 1000000 --steps 3 --warmup 1 --no-cpu-baseline --no-flat --rerank 600 > $K/stats.log 2>&1
  cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null ;;
fpmc)
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $C --output-format csv -d $O/$C -o fr -- python $R/scripts/frontier_pmc.py > $K/$C.log 2>&1
    extract $C
  done ;;
gsearch)
  export JVECTOR_HIP_GRAPH_TRAVERSAL=device
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o gs -- python $R/scripts/frontier_pmc.py > $K/stats.log 2>&1
  cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $C --output-format csv -d $O/$C -o gs -- python $R/scripts/frontier_pmc.py > $K/$C.log 2>&1
    extract $C
  done ;;
dense)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o dense -- python $R/scripts/dense_pmc.py > $K/stats.log 2>&1
  cp $O/stats/*kernel_stats.csv $K/ 2>/dev/null
  for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
    timeout 150 rocprofv3 --pmc $C --output-format csv -d $O/$C -o dense -- python $R/scripts/dense_pmc.py 64 > $K/$C.log 2>&1
    extract $C
  done ;;
esac
ls -la $K
