#!/bin/bash
# GPU session S: pq_encode scalar-cache variant vs LDS variant (parity + timing at 1M x 768 and 1M x 1536 / PQ-192)
set -u
O=gpurun_out/r2s; mkdir -p $O
for mode in sgpr lds; do
  if [ $mode = lds ]; then export JVECTOR_HIP_ENCODE_LDS=1; else unset JVECTOR_HIP_ENCODE_LDS; fi
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encode or reconstruction or c2_full" > $O/pytest_$mode.log 2>&1
  grep -E "passed|failed|error" $O/pytest_$mode.log | tail -2
  timeout 600 python - <<'PY'
import os, time, torch, numpy as np
import jvector_amd as J
ctx = J.HipContext(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for D, M in ((768, 96), (1536, 192), (128, 16)):
    N = 1_000_000
    base = torch.randn(N, D, generator=g, device=dev)
    cb = base[:256].reshape(256, M, D // M).permute(1, 0, 2).contiguous().reshape(-1)
    pq = J.ProductQuantization.from_codebooks(ctx, D, M, cb.cpu().numpy())
    out = torch.empty(N, M, dtype=torch.uint8, device=dev)
    pq.encode_all(base, out=out); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        pq.encode_all(base, out=out)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"encode[{'lds' if os.environ.get('JVECTOR_HIP_ENCODE_LDS') else 'sgpr'}] {N}x{D} PQ-{M}: {ms:.3f} ms  {3*256*D*N/ms/1e9:.1f} Tflop/s non-fused  checksum {int(out.sum())}")
PY
done
