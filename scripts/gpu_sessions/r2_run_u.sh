#!/bin/bash
# GPU session U: flat-scan kernel with non-temporal code loads — fabric traffic (FETCH_SIZE) and time, NT on / off
set -u
O=$PWD/gpurun_out/r2u; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for nt in 1 0; do
  export JVECTOR_HIP_ADC_MQ_NT=$nt
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/u_$nt -o bench -- python $R/bench.py --mode flat --steps 3 --warmup 1 --no-cpu-baseline --cal-queries 256 --eval-queries 256 > $O/flat_nt$nt.json 2> $O/flat_nt$nt.err
  f=$(find /tmp/u_$nt -name "*counter_collection.csv" | head -1)
  python - "$f" $nt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "adc_mq_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v.sort()
    big = [x for x in v if x > 0.5 * v[-1]]
    print(f"NT={sys.argv[2]} {k}: {len(big)} timed-shape launches, FETCH_SIZE avg {sum(big)/len(big)/1e6:.1f} M units (x2 KiB-corrected GB: {sum(big)/len(big)*1024*2/1e9:.2f})")
PY
  python - <<PY
import json
d = json.loads(open("$O/flat_nt$nt.json").read().strip().splitlines()[-1])
print("NT=$nt", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"))
PY
done
