"""Bind jvector_amd to the mock library (tests/mock/build_mock.py) for the lifetime of a process or a `with` block.
TEST HARNESS: lets CPU-only tests (and the spawned ranks of the world_size-2 tests) drive the C ABI's host logic."""
import contextlib
import ctypes as C
import os


@contextlib.contextmanager
def mock_jvector():
    import build_mock
    import jvector_amd
    import jvector_amd._lib as L
    lib = C.CDLL(build_mock.build())
    for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    saved, L._lib = L._lib, lib
    saved_threads = os.environ.get("JVECTOR_HIP_HOST_THREADS")
    os.environ["JVECTOR_HIP_HOST_THREADS"] = "1"
    try:
        yield jvector_amd
    finally:
        L._lib = saved
        if saved_threads is None:
            os.environ.pop("JVECTOR_HIP_HOST_THREADS", None)
        else:
            os.environ["JVECTOR_HIP_HOST_THREADS"] = saved_threads
