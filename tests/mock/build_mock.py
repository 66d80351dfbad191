"""Builds build/mock/libjvector_hip_mock.so: the library's HOST sources compiled with g++ against a CPU mock of the HIP
runtime (mock_hip.cpp) and CPU kernel launchers (mock_kernels.cpp, oracle arithmetic / shared kernel bodies on the lane
emulator).  TEST HARNESS — lets the CPU suite exercise the C ABI's host logic; the product never loads it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "build", "mock")
LIB = os.path.join(OUT, "libjvector_hip_mock.so")
HOST = ["cabi", "graph_search", "sharded", "build_score", "builder", "nvq", "pq_train", "formats", "compat_host"]
CXXF = ["-std=c++17", "-O2", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-include",
        os.path.join(ROOT, "tests", "mock", "mock_prefix.h"), "-fPIC", "-Wno-unknown-pragmas", "-Wno-unused-function"]


def _sources():
    srcs = [os.path.join(ROOT, "jvector_amd", "csrc", f + ".cpp") for f in HOST]
    srcs += [os.path.join(ROOT, "tests", "mock", f) for f in ("mock_hip.cpp", "mock_kernels.cpp")]
    srcs += [os.path.join(ROOT, "oracle", f) for f in ("jv_oracle.c", "jv_oracle_simd.c", "jv_nvq.c")]
    return srcs


def _deps():
    d = list(_sources())
    for sub in ("jvector_amd/csrc", "tests/mock", "tests/emu", "include", "oracle"):
        p = os.path.join(ROOT, sub)
        d += [os.path.join(p, f) for f in os.listdir(p) if f.endswith(".h")]
    return d


def build():
    if os.environ.get("JV_MOCK_LIBRARY"):  # a pre-built variant (scripts/asan_mock.sh)
        return os.environ["JV_MOCK_LIBRARY"]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in _deps()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        objs.append(obj)
        if src.endswith(".c"):
            cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-c", src, "-o", obj]
        else:
            cmd = ["g++"] + CXXF + ["-c", src, "-o", obj]
        procs.append((subprocess.Popen(cmd), cmd))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("mock build failed: " + " ".join(cmd))
    # -Bsymbolic: calls between the library's own entry points must not be captured by a real HIP runtime / library that
    # another module (torch) may have put in the global symbol scope
    subprocess.check_call(["g++", "-shared", "-Wl,-Bsymbolic", "-o", LIB] + objs + ["-lpthread", "-lm", "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build())
