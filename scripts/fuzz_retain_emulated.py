"""One-off validation aid (CPU): the robust-prune kernel body (csrc/rd_body.h) on the lane emulator against the oracle's sequential
retainDiverse on random shapes, with every form of the kernel in the draw — table look-ups / the square table / table-free, a slot's
entries split over idle lanes or not, incremental tests with random chunk sizes, wide or row-by-row staging, three lane schedules.
usage: python scripts/fuzz_retain_emulated.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_retain_diverse as TR  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
emu = TR.emu.__wrapped__() if hasattr(TR.emu, "__wrapped__") else None
if emu is None:
    import subprocess
    src = [os.path.join(ROOT, "tests", "emu", "rd_emu.cpp")]
    lib = os.path.join(ROOT, "build", "emu", "librd_emu.so")
    if not os.path.exists(lib):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src[0], "-o", lib])
    emu = C.CDLL(lib)
emu.rd_emu_run.restype = C.c_int
emu.rd_emu_run_tf.restype = C.c_int
t_end = time.time() + budget
cases = 0
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
while time.time() < t_end:
    M = int(rng.choice([3, 5, 8, 12, 16, 19, 32, 48, 64, 96, 192]))
    D = 8 * M
    N = int(rng.integers(260, 600))   # (make_case draws 256 distinct centroid rows)
    P = int(rng.integers(3, 8))
    Cn = int(rng.integers(8, 150))
    max_degree = int(rng.choice([4, 8, 16, 32, 48, 64]))
    alpha = float(rng.choice([1.0, 1.2, 1.4, 2.0]))
    vsf = int(rng.integers(0, 3))
    opq, codes, tri, cand, sc, count, before = TR.make_case(int(rng.integers(1, 1 << 30)), N, D, M, P, Cn, vsf, max_degree)
    want = TR.oracle_selection(opq, codes, tri, vsf, cand, sc, count, before, max_degree, alpha)
    form = str(rng.choice(["table", "square", "tf"]))
    env = {"RD_EMU_CHUNK": str(int(rng.choice([0, 0, 1, 2, 5, 8, 64]))), "RD_EMU_SPLIT": str(int(rng.integers(0, 2))), "RD_EMU_WIDE": str(int(rng.integers(0, 2))),
           "RD_EMU_SQUARE": "1" if form == "square" else "0", "EMU_LANE_ORDER": str(rng.choice(["", "reverse", "random:%d" % int(rng.integers(1, 99))]))}
    os.environ.update(env)
    sel = np.full((P, max_degree), -7, np.int32)
    nsel = np.full(P, -7, np.int32)
    se = np.zeros(P, np.float32)
    if form == "tf":
        cb = np.ascontiguousarray(opq.codebooks, np.float32)
        emu.rd_emu_run_tf(p(tri), p(cb), p(codes), C.c_int64(N), p(cand), p(sc), p(count), p(before), P, Cn, M, 256, vsf, max_degree, C.c_float(alpha), p(sel), p(nsel), p(se))
    else:
        emu.rd_emu_run(p(tri), p(codes), C.c_int64(N), p(cand), p(sc), p(count), p(before), P, Cn, M, 256, vsf, max_degree, C.c_float(alpha), p(sel), p(nsel), p(se))
    if not (np.array_equal(sel, want[0]) and np.array_equal(nsel, want[1]) and np.array_equal(se, want[2], equal_nan=True)):
        print("MISMATCH", dict(M=M, N=N, P=P, Cn=Cn, max_degree=max_degree, alpha=alpha, vsf=vsf, form=form, env=env))
        sys.exit(1)
    cases += 1
print(f"fuzz_retain_emulated: {cases} random cases, every kernel form, all identical to the oracle (seed {seed}, {budget:.0f} s)")
