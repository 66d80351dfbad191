"""TEST INFRASTRUCTURE — ctypes loader for oracle/_ref/libjvector_ref.so: the reference's OWN native kernels
(/root/reference/jvector-native/src/main/native/src/jvector_simd_kernels.cpp + jvector_simd.cpp, compiled unmodified by
oracle/ref_build/build.sh against a scalar lane emulation of the Highway ops they use).  Reference-EXECUTED outputs at the C
boundary (the 22 + 2 symbols of jvector_simd_kernel_list.h:36-62 / jvector_simd.h:47,53), per ISA tier:

    R = ref.lib()                       # None when neither the library nor /root/reference is present
    R.fn("avx3", "dot_product_f32")(...)   # 16 f32 lanes + fma    (the AVX-512 tiers)
    R.fn("avx2", ...)                      # 8 lanes + fma
    R.fn("sse42", ...)                     # 4 lanes, no fma
    R.fn(None, ...)                        # the library's own CPUID-dispatched export

Only tests/ may import this (the product, bench.py's timed region and smoke() never do).  On the GPU box /root/reference does
not exist: the prebuilt library travels with the snapshot; when it is missing the tests that need it SKIP and say so."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libjvector_ref.so")
_BUILD = os.path.join(_HERE, "ref_build", "build.sh")
REFERENCE_SRC = os.path.join(os.environ.get("JVECTOR_REFERENCE", "/root/reference"), "jvector-native", "src", "main", "native", "src")
TIERS = ("avx3", "avx2", "sse42")
LANES = {"avx3": 16, "avx2": 8, "sse42": 4}

_fp, _u8p, _F, _I, _Z = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_float, C.c_int, C.c_size_t
# name -> (restype, argtypes): jvector_simd_kernel_list.h:36-62
SIGNATURES = {
    "cosine_f32": (_F, [_fp, _Z, _fp, _Z, _Z]),
    "dot_product_f32": (_F, [_fp, _Z, _fp, _Z, _Z]),
    "euclidean_f32": (_F, [_fp, _Z, _fp, _Z, _Z]),
    "add_in_place_f32": (None, [_fp, _fp, _Z]),
    "add_scalar_in_place_f32": (None, [_fp, _F, _Z]),
    "sub_in_place_f32": (None, [_fp, _fp, _Z]),
    "sub_scalar_in_place_f32": (None, [_fp, _F, _Z]),
    "max_f32": (_F, [_fp, _Z]),
    "min_in_place_f32": (None, [_fp, _fp, _Z]),
    "assemble_and_sum_f32": (_F, [_fp, _I, _u8p, _I, _Z]),
    "assemble_and_sum_pq_f32": (_F, [_fp, _Z, _u8p, _I, _u8p, _I, _I]),
    "pq_decoded_cosine_similarity_f32": (_F, [_u8p, _I, _Z, _I, _fp, _fp, _F]),
    "calculate_partial_sums_dot_f32": (None, [_fp, _I, _Z, _I, _fp, _I, _fp]),
    "calculate_partial_sums_euclidean_f32": (None, [_fp, _I, _Z, _I, _fp, _I, _fp]),
    "calculate_partial_sums_self_magnitude_f32": (None, [_fp, _I, _Z, _I, _fp]),
    "nvq_quantize_8bit": (None, [_fp, _Z, _F, _F, _F, _F, _u8p]),
    "nvq_loss": (_F, [_fp, _Z, _F, _F, _F, _F, _I]),
    "nvq_uniform_loss": (_F, [_fp, _Z, _F, _F, _I]),
    "nvq_square_l2_distance_8bit": (_F, [_fp, _u8p, _Z, _F, _F, _F, _F]),
    "nvq_dot_product_8bit": (_F, [_fp, _u8p, _Z, _F, _F, _F, _F]),
    "nvq_cosine_8bit_packed": (C.c_int64, [_fp, _u8p, _Z, _F, _F, _F, _F, _fp]),
    "nvq_shuffle_query_in_place_8bit": (None, [_fp, _Z]),
}


def build(force=False):
    """Runs the committed recipe when the reference sources are present (this container); returns the library path or None."""
    have_src = os.path.isfile(os.path.join(REFERENCE_SRC, "jvector_simd_kernels.cpp"))
    if have_src and (force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(os.path.join(_HERE, "ref_build", "hwy", "highway.h")), os.path.getmtime(_BUILD),
            os.path.getmtime(os.path.join(_HERE, "ref_build", "ref_exports.cpp")))):
        subprocess.check_call(["bash", _BUILD])
    return _SO if os.path.isfile(_SO) else None


class RefLib:
    def __init__(self, path):
        self.path = path
        self.dll = C.CDLL(path)
        self.dll.jvref_build_info.restype = C.c_char_p
        self.dll.jvector_simd_get_active_isa.restype = C.c_char_p
        self.dll.jvector_simd_get_max_isa_env.restype = C.c_char_p
        self._cache = {}

    def fn(self, tier, name):
        key = (tier, name)
        if key not in self._cache:
            f = getattr(self.dll, name if tier is None else f"jvref_{tier}_{name}")
            f.restype, f.argtypes = SIGNATURES[name]
            self._cache[key] = f
        return self._cache[key]

    def info(self):
        return self.dll.jvref_build_info().decode()

    def active_isa(self):
        return self.dll.jvector_simd_get_active_isa().decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            return None
        _lib = RefLib(path)
    return _lib


def fp(a):
    return a.ctypes.data_as(_fp)


def u8(a):
    return a.ctypes.data_as(_u8p)


def unpack_cosine(packed):
    """nvq_cosine_8bit_packed's int64: low 32 bits = float bits of sum, high 32 = float bits of bMagnitude (kernels.cpp:1637-1641)"""
    import numpy as np
    packed &= (1 << 64) - 1
    return tuple(float(x) for x in np.array([packed & 0xFFFFFFFF, packed >> 32], np.uint32).view(np.float32))
