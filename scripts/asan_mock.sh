#!/bin/bash
# AddressSanitizer pass over the library's HOST code (CPU only): builds the mock-device library (tests/mock/) with
# -fsanitize=$SAN and runs the mock tests that do not use the fiber-based lane emulator (ASan cannot follow its
# stack switches) plus the corrupted-input fuzz of the format readers.  Last run (round 4 final: + the sharded exchange over the external transport, two gloo ranks): clean.
set -eu
# SAN=undefined scripts/asan_mock.sh runs the same pass under UndefinedBehaviorSanitizer (round 2: clean as well)
SAN=${SAN:-address}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/jv_asan; mkdir -p "$OUT"; cd "$OUT"
CXXF="-std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -include $ROOT/tests/mock/mock_prefix.h -fPIC -Wno-unknown-pragmas -Wno-unused-function"
for f in cabi graph_search sharded build_score builder nvq pq_train formats compat_host; do g++ $CXXF -c "$ROOT/jvector_amd/csrc/$f.cpp" -o $f.o & done
g++ $CXXF -c "$ROOT/tests/mock/mock_hip.cpp" -o mock_hip.o &
g++ $CXXF -c "$ROOT/tests/mock/mock_kernels.cpp" -o mock_kernels.o &
for f in jv_oracle jv_oracle_simd jv_nvq; do gcc -O1 -g -fsanitize=$SAN -std=c11 -fPIC -ffp-contract=off -c "$ROOT/oracle/$f.c" -o $f.o & done
wait
g++ -shared -fsanitize=$SAN -Wl,-Bsymbolic -o libjvector_hip_mock_asan.so *.o -lpthread -lm -ldl
cd "$ROOT"
RT=libasan.so; [ "$SAN" = undefined ] && RT=libubsan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=$RT) JV_MOCK_LIBRARY="$OUT/libjvector_hip_mock_asan.so"
# (the lane emulator's fibers cannot run under ASan: AUTO graphs are sent to the host searcher for this pass, tests that pin the device
#  traversal or count device calls are left out)
export JVECTOR_HIP_GRAPH_TRAVERSAL=1
python -m pytest tests/test_mock_device.py -x -q -p no:cacheprovider \
  -k "parity_suite or search_flat or host_graph_searcher or load_index or build_score or fused_build or continuous_batching or several_host or sharded_cabi_local or training_entry or fewer_clusters or ((edge_cases or irregular or negative_scores or accept_ords or ties) and host)"
python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/mock")
import jvector_amd, jvector_amd._lib as L
lib = C.CDLL(os.environ["JV_MOCK_LIBRARY"])
for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
    for name, (res, args) in table.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
L._lib = lib
os.environ["JVECTOR_HIP_HOST_THREADS"] = "1"
import test_graph_search as T
ctx = jvector_amd.HipContext(0)
T.run_searcher_object_cases(jvector_amd, ctx, cases=2, traversal="host", n_nodes=1200, nq=8)   # GraphSearcher objects: session path of the host searcher
T.run_wide_rows(jvector_amd, ctx, N=500, nq=4, traversals=("host",))                           # rows wider than 64: multi-word masks
T.run_small_cluster_count(jvector_amd, ctx, N=600, traversals=("host",))                       # padded quantizer under the searcher
ctx.close()
print("searcher objects / wide rows / small cluster counts on the host searcher under the sanitizer: clean")
PY
python -m pytest tests/test_sharded_cabi.py -x -q -p no:cacheprovider -k "local_shards"
# round 4: the one sharded exchange (jv_hip_sharded_merge_rerank) with local shards and with two gloo ranks over the external transport
python -m pytest tests/test_sharded.py -x -q -p no:cacheprovider -k "local_shards_on_the_mock or two_ranks_gloo"
python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, "tests")
import jvector_amd._lib as L
lib = C.CDLL(os.environ["JV_MOCK_LIBRARY"])
for table in (L.SIGNATURES, L.COMPAT_SIGNATURES, L.FORMAT_SIGNATURES):
    for name, (res, args) in table.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
L._lib = lib
import test_formats_cpu as T
T.test_readers_survive_corrupted_input(); T.test_odgi_rejects_corruption(); T.test_odgi_v6_fused_multilayer(True)
print("format readers under the sanitizer: clean")
PY
