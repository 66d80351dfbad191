#!/bin/bash
# GPU session T (final state of round 2): the default bench line, the rocprofv3 recipe over it, BASELINE C5 at its full size
set -u
O=gpurun_out/r2t; mkdir -p $O
( time timeout 1200 python bench.py > $O/default_bench_line.json 2> $O/default_bench.err ) 2> $O/default_bench.time
tail -c 600 $O/default_bench_line.json; echo; cat $O/default_bench.time | tr '\n' ' '; echo
timeout 2400 bash scripts/profile_r2.sh r2_10m_v6 > $O/profile.log 2>&1
tail -5 $O/profile.log | cut -c1-300
( time timeout 1500 python bench.py --workload c5 --n 10000000 > $O/c5_10m_bench_line.json 2> $O/c5_10m.err ) 2> $O/c5.time
tail -c 900 $O/c5_10m_bench_line.json; echo; cat $O/c5.time | tr '\n' ' '; echo
