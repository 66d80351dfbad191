#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds oracle/_ref/libjvector_ref.so: the reference's OWN native kernels
#   $REF/jvector_simd_kernels.cpp (three times: JV_ISA = AVX3 / AVX2 / SSE42, as meson.build:28-66 does),
#   $REF/jvector_avx3_dl_kernels.cpp, $REF/jvector_avx3_spr_kernels.cpp (empty tiers), $REF/jvector_simd.cpp (CPUID dispatch)
# compiled UNMODIFIED from where they lie under /root/reference, against oracle/ref_build/hwy/highway.h (a scalar lane
# emulation of the Highway ops they use — the reference's Highway submodule is not vendored) plus ref_exports.cpp.
# The reference's build system (meson) is not run.  Output only under oracle/_ref/ (git-ignored; travels to the GPU box with
# the snapshot).  /root/reference does not exist on the GPU box: there this script does nothing and the prebuilt .so is used.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${JVECTOR_REFERENCE:-/root/reference}/jvector-native/src/main/native/src"
OUT="$HERE/../_ref"
if [ ! -f "$REF/jvector_simd_kernels.cpp" ]; then
    echo "[oracle/_ref] reference sources not present ($REF): keeping the prebuilt library, if any" >&2
    exit 0
fi
mkdir -p "$OUT/obj"
CXX="${CXX:-g++}"
COMMON="-std=c++17 -O2 -fPIC -fvisibility=hidden -I$HERE -I$REF -Wno-unknown-pragmas -Wno-attributes"
# -mfma for the two tiers whose -march implies it in the reference's build (haswell, skylake-avx512); the compiler's default
# contraction rule then treats the kernels' scalar tails as it does there.  The SSE4.2 tier has no fma to contract into.
$CXX $COMMON -mfma -DHWY_EMU_MAX_BYTES=64 -DJV_ISA=AVX3  -c "$REF/jvector_simd_kernels.cpp" -o "$OUT/obj/kernels_avx3.o"
$CXX $COMMON -mfma -DHWY_EMU_MAX_BYTES=32 -DJV_ISA=AVX2  -c "$REF/jvector_simd_kernels.cpp" -o "$OUT/obj/kernels_avx2.o"
$CXX $COMMON       -DHWY_EMU_MAX_BYTES=16 -DJV_ISA=SSE42 -c "$REF/jvector_simd_kernels.cpp" -o "$OUT/obj/kernels_sse42.o"
$CXX $COMMON -DHWY_EMU_MAX_BYTES=64 -c "$REF/jvector_avx3_dl_kernels.cpp"  -o "$OUT/obj/kernels_avx3_dl.o"
$CXX $COMMON -DHWY_EMU_MAX_BYTES=64 -c "$REF/jvector_avx3_spr_kernels.cpp" -o "$OUT/obj/kernels_avx3_spr.o"
$CXX $COMMON -DJVECTOR_BUILD -c "$REF/jvector_simd.cpp" -o "$OUT/obj/dispatch.o"
$CXX $COMMON -c "$HERE/ref_exports.cpp" -o "$OUT/obj/ref_exports.o"
$CXX -shared -o "$OUT/libjvector_ref.so" "$OUT"/obj/*.o -lm
rm -rf "$OUT/obj"
echo "[oracle/_ref] built $OUT/libjvector_ref.so" >&2
