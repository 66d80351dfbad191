#!/bin/bash
# Round 6, GPU session H: sessions F and G measured a library whose exchange-area accesses had become FLAT operations (the aligned
# pointer went through an integer).  With that fixed: the headline with 8 lanes per survivor (default) and with 32 / 16 / 8 by survivor
# count (variant), and BASELINE config 5.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_builder.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
for v in default varlps; do
  if [ $v = default ]; then unset JVECTOR_HIP_LIBRARY; else export JVECTOR_HIP_LIBRARY=$R/build/variants/libjvector_hip_$v.so; fi
  timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_$v.json 2> $O/bench_$v.err
  echo "bench $v rc=$?" | tee -a $O/summary.txt
done
export JVECTOR_HIP_LIBRARY=$R/build/variants/libjvector_hip_varlps.so
timeout 600 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py -m gpu -x -q > $O/pytest_varlps.txt 2>&1
echo "pytest varlps rc=$?" | tee -a $O/summary.txt
tail -1 $O/pytest_varlps.txt | tee -a $O/summary.txt
unset JVECTOR_HIP_LIBRARY
timeout 900 python bench.py --workload c5 --n 10000000 --no-cpu-baseline > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6h")
for f in ("bench_default.json","bench_varlps.json","c5.json"):
    try:
        l=json.loads(open(os.path.join(d,f)).read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), l.get("seconds"))
    except Exception as e:
        print(f, "failed", e)
PY
