#!/bin/bash
# round-2 GPU session C: L2 gather micro-benchmark, sharded C ABI over real RCCL, C2 line
set -u
O=gpurun_out/r2c; mkdir -p $O
timeout 120 build/gather_bench 2>&1 | tee $O/gather_bench.log
timeout 600 python -m pytest tests/test_sharded_cabi.py tests/test_sharded.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_sharded.log
# can two RCCL ranks share the one GPU?  (expected: RCCL refuses duplicate devices — recorded either way)
cat > /tmp/two_ranks.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
rank = int(sys.argv[1])
import jvector_amd as J
from jvector_amd.sharded import Communicator
import test_sharded_cabi as T
ctx = J.HipContext(0)
p = "/tmp/jv_uid"
if rank == 0:
    open(p + ".tmp", "wb").write(Communicator.unique_id(ctx)); os.replace(p + ".tmp", p)
else:
    while not os.path.exists(p): time.sleep(0.01)
comm = Communicator(ctx, rank, 2, open(p, "rb").read())
T.run_sharded_equals_single(J, ctx, comm, 2, rank=rank, world=2, N=40000, D=128, M=16, rerank_k=100)
print("rank", rank, "OK: 2 ranks x 2 shards == single index (real RCCL)")
PY
rm -f /tmp/jv_uid; (timeout 120 python /tmp/two_ranks.py 0 > $O/two_ranks_0.log 2>&1 &) ; timeout 120 python /tmp/two_ranks.py 1 > $O/two_ranks_1.log 2>&1; sleep 2; tail -3 $O/two_ranks_0.log $O/two_ranks_1.log
timeout 600 python bench.py --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -12 $O/bench_c2.err; head -c 1200 $O/bench_c2.json; echo
