#!/usr/bin/env python3
"""TEST HARNESS — `python tests/mock/bench_on_mock.py <bench.py arguments>`: bench.py's own main() with the engine bound to
the mock device (tests/mock: the C ABI's host sources over a CPU stand-in of the HIP runtime), torch on the CPU, gloo in place
of RCCL for torch.distributed and the shared-memory RCCL shim (JVECTOR_HIP_RCCL_PATH) for the engine's own communicator.
It exists so that the CPU suite can run the REAL command line of a multi-GPU bench run — `--gpus 2` given to ONE process,
which must start its two ranks itself — without a GPU.  bench.py re-executes sys.argv[0], i.e. this file, for every rank.
Numbers are meaningless; the control flow and the JSON contract are what is checked.  Never used by the product."""
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "mock"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("JVECTOR_HIP_HOST_THREADS", "1")
os.environ.setdefault("JV_MOCK_DEVICES", "8")   # one mock device per rank
os.environ.setdefault("OMP_NUM_THREADS", "2")
os.environ.setdefault("MKL_NUM_THREADS", "2")


def main():
    import build_mock
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    L._lib = lib
    if not os.environ.get("JVECTOR_HIP_RCCL_PATH"):
        import test_sharded_cabi as TS
        os.environ["JVECTOR_HIP_RCCL_PATH"] = TS.build_shim()
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    import bench
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **kw: real_init("gloo")  # rank / world / master from the launcher's environment

    class TorchProxy:
        cuda = types.SimpleNamespace(set_device=lambda *_a: None, synchronize=lambda *_a: None,
                                     current_stream=lambda *_a: types.SimpleNamespace(cuda_stream=0))

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    bench.torch = TorchProxy()
    bench.main()


if __name__ == "__main__":
    main()
