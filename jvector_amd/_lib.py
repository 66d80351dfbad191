"""ctypes loader for libjvector_hip.so (the C ABI declared in include/jvector_hip.h).

There is no Python/CPU implementation behind this module: if the shared library is missing or no gfx950
device is usable, the calls raise.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C jvector_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (JVECTOR_HIP_LIBRARY: developer aid — an A/B build of the same C ABI, scripts/build_variant.sh)
LIB_PATH = os.environ.get("JVECTOR_HIP_LIBRARY") or os.path.join(_HERE, "libjvector_hip.so")

JV_OK, JV_ERR_INVALID, JV_ERR_NO_DEVICE, JV_ERR_HIP, JV_ERR_OOM, JV_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class JVectorHipError(RuntimeError):
    """A HIP runtime failure inside libjvector_hip (JV_ERR_HIP / JV_ERR_OOM)."""


class NoDeviceError(JVectorHipError):
    """No usable gfx950 device: the engine has no CPU fallback."""


class UnsupportedError(JVectorHipError):
    """The reference feature exists but is outside the HIP engine's scope (UnsupportedOperationException)."""


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/jvector_hip.h one to one
SIGNATURES = {
    "jv_hip_version": (C.c_char_p, []),
    "jv_hip_last_error": (C.c_char_p, []),
    "jv_hip_device_count": (_i, []),
    "jv_hip_active_arch": (C.c_char_p, [_i]),
    "jv_hip_ctx_create": (_i, [_i, _p, C.POINTER(_p)]),
    "jv_hip_ctx_destroy": (_i, [_p]),
    "jv_hip_ctx_sync": (_i, [_p]),
    "jv_hip_ctx_stream": (_p, [_p]),
    "jv_hip_ctx_profile": (_i, [_p, _i]),
    "jv_hip_ctx_set_option": (_i, [_p, C.c_char_p, _i64]),
    "jv_hip_ctx_clear_option": (_i, [_p, C.c_char_p]),
    "jv_hip_ctx_get_stat": (_i, [_p, C.c_char_p, C.POINTER(_i64)]),
    "jv_hip_ctx_reset_stats": (_i, [_p]),
    "jv_hip_ctx_profile_read": (_i, [_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "jv_hip_pq_create": (_i, [_p, _i, _i, _i, _p, _p, _p, C.POINTER(_p)]),
    "jv_hip_pq_load": (_i, [_p, _p, _sz, C.POINTER(_sz), C.POINTER(_p)]),
    "jv_hip_pq_set_anisotropic_threshold": (_i, [_p, C.c_float]),
    "jv_hip_pq_anisotropic_threshold": (C.c_float, [_p]),
    "jv_hip_pq_train": (_i, [_p, _p, _i64, _i, _i, _i, _i, C.c_uint64, C.POINTER(_p)]),
    "jv_hip_pq_train_anisotropic": (_i, [_p, _p, _i64, _i, _i, _i, _i, C.c_float, C.c_uint64, C.POINTER(_p)]),
    "jv_hip_pq_refine": (_i, [_p, _p, _p, _i64, _i, C.c_uint64, C.POINTER(_p)]),
    "jv_hip_pq_write": (_i, [_p, _p, _i, _p, _sz, C.POINTER(_sz)]),
    "jv_hip_pq_destroy": (_i, [_p]),
    "jv_hip_pq_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "jv_hip_pq_self_magnitudes": (_i, [_p, _p, _p]),
    "jv_hip_codes_create": (_i, [_p, _p, _i64, C.POINTER(_p)]),
    "jv_hip_codes_wrap": (_i, [_p, _p, _i64, _p, C.POINTER(_p)]),
    "jv_hip_codes_upload": (_i, [_p, _p, _i64, _i64, _p]),
    "jv_hip_codes_download": (_i, [_p, _p, _i64, _i64, _p]),
    "jv_hip_codes_destroy": (_i, [_p]),
    "jv_hip_codes_count": (_i64, [_p]),
    "jv_hip_codes_device_ptr": (_p, [_p]),
    "jv_hip_vectors_create": (_i, [_p, _i64, _i, C.POINTER(_p)]),
    "jv_hip_vectors_wrap": (_i, [_p, _i64, _i, _p, C.POINTER(_p)]),
    "jv_hip_vectors_upload": (_i, [_p, _p, _i64, _i64, _p]),
    "jv_hip_vectors_invalidate": (_i, [_p]),
    "jv_hip_vectors_destroy": (_i, [_p]),
    "jv_hip_pq_encode": (_i, [_p, _p, _p, _i64, _p]),
    "jv_hip_pq_encode_into": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "jv_hip_luts_create": (_i, [_p, _p, _i, C.POINTER(_p)]),
    "jv_hip_luts_build": (_i, [_p, _p, _p, _i, _i, _i]),
    "jv_hip_luts_destroy": (_i, [_p]),
    "jv_hip_luts_download": (_i, [_p, _p, _i, _p, _p]),
    "jv_hip_luts_bound_tables": (_i, [_p, _p, _p, _p]),
    "jv_hip_adc_scan": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "jv_hip_adc_scores": (_i, [_p, _p, _p, _p, _i, _p]),
    "jv_hip_fused_create": (_i, [_p, _p, _i64, _i, C.POINTER(_p)]),
    "jv_hip_fused_upload": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "jv_hip_fused_destroy": (_i, [_p]),
    "jv_hip_fused_scores": (_i, [_p, _p, _p, _p, _p, _p]),
    "jv_hip_exact_scores": (_i, [_p, _p, _p, _i, _i, _p, _i, _p]),
    "jv_hip_exact_scan": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _p]),
    "jv_hip_exact_pair_scores": (_i, [_p, _p, _i, _p, _i, _p, _i, _p]),
    "jv_hip_exact_scan_dense": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _p]),
    "jv_hip_topk": (_i, [_p, _p, _p, _i, _i64, _i64, C.c_int32, _i, _p, _p]),
    "jv_hip_search_flat": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, C.c_int32, _p, _p]),
    "jv_hip_graph_create": (_i, [_p, _i64, _i, C.POINTER(_p)]),
    "jv_hip_graph_set_level": (_i, [_p, _p, _i, _i, _p, _p, _i]),
    "jv_hip_graph_set_entry": (_i, [_p, C.c_int32, _i]),
    "jv_hip_graph_destroy": (_i, [_p]),
    "jv_hip_graph_set_traversal": (_i, [_p, _i]),
    "jv_hip_pair_table_create": (_i, [_p, _p, _i, C.POINTER(_p)]),
    "jv_hip_pair_table_size": (_i64, [_p]),
    "jv_hip_pair_table_download": (_i, [_p, _p, _p]),
    "jv_hip_pair_table_destroy": (_i, [_p]),
    "jv_hip_code_pair_scores": (_i, [_p, _p, _p, _p, _i, _p, _i, _p]),
    "jv_hip_fused_build": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "jv_hip_fused_download": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "jv_hip_pq_decode": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "jv_hip_direct_scores": (_i, [_p, _p, _p, _i, _i, _p, _i, _p]),
    "jv_hip_graph_search": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "jv_hip_graph_search_filtered": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i64, _p, _p, _p]),
    "jv_hip_searcher_create": (_i, [_p, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    "jv_hip_searcher_search": (_i, [_p, _p, _p, _i, _i, _i, _i, C.c_float, C.c_float, _p, _i64, _p, _p, _p, _p, _p]),
    "jv_hip_searcher_resume": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "jv_hip_searcher_destroy": (_i, [_p]),
    "jv_hip_device_alloc": (_i, [_p, C.c_size_t, C.POINTER(_p)]),
    "jv_hip_device_free": (_i, [_p, _p]),
    "jv_hip_graph_set_level0_device": (_i, [_p, _p, _p, _i]),
    "jv_hip_retain_diverse": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _i, C.c_float, _p, _p, _p]),
    "jv_hip_builder_create": (_i, [_p, _p, _p, _p, _i, _i, _i, C.c_float, C.c_float, C.POINTER(_p)]),
    "jv_hip_builder_seed": (_i, [_p, _p, C.c_int32]),
    "jv_hip_builder_insert_batch": (_i, [_p, _p, _p, _i]),
    "jv_hip_builder_improve_batch": (_i, [_p, _p, _p, _i]),
    "jv_hip_builder_finish": (_i, [_p, _p, _p]),
    "jv_hip_builder_stats": (_i, [_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "jv_hip_builder_neighbors_device": (_p, [_p, C.POINTER(_i)]),
    "jv_hip_builder_working_lists": (_i, [_p, _p, _p, _p, _p]),
    "jv_hip_builder_destroy": (_i, [_p]),
    "jv_hip_build_layered": (_i, [_p, _p, _p, _p, _i, _i, _i, C.c_float, C.c_float, _i, _i, C.c_uint64, _i, C.POINTER(_p)]),
    "jv_hip_layered_info": (_i, [_p, C.POINTER(_i), C.POINTER(C.c_int32), C.POINTER(_i), C.POINTER(_i64)]),
    "jv_hip_layered_level": (_i, [_p, _p, _i, _p, _p]),
    "jv_hip_layered_level0_device": (_p, [_p]),
    "jv_hip_layered_stats": (_i, [_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "jv_hip_layered_destroy": (_i, [_p]),
    "jv_hip_comm_unique_id": (_i, [_p]),
    "jv_hip_comm_create": (_i, [_p, _p, _i, _i, _p]),
    "jv_hip_comm_create_external": (_i, [_p, _i, _i, _p, _p, _p]),
    "jv_hip_comm_destroy": (_i, [_p]),
    "jv_hip_comm_rank": (_i, [_p]),
    "jv_hip_comm_world": (_i, [_p]),
    "jv_hip_comm_count": (_i, [_p, C.POINTER(_i)]),
    "jv_hip_comm_all_gather": (_i, [_p, _p, _p, _sz, _p]),
    "jv_hip_sharded_topk": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "jv_hip_sharded_search_flat": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "jv_hip_sharded_merge_rerank": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "jv_hip_nvq_create": (_i, [_p, _i, _i, _p, C.POINTER(_p)]),
    "jv_hip_nvq_compute": (_i, [_p, _p, _i, C.POINTER(_p)]),
    "jv_hip_nvq_set_learn": (_i, [_p, _i]),
    "jv_hip_nvq_dimension": (_i, [_p]),
    "jv_hip_nvq_subvectors": (_i, [_p]),
    "jv_hip_nvq_global_mean": (_i, [_p, _p, _p]),
    "jv_hip_nvq_destroy": (_i, [_p]),
    "jv_hip_nvq_vectors_create": (_i, [_p, _p, _i64, C.POINTER(_p)]),
    "jv_hip_nvq_encode": (_i, [_p, _p, _p, _i64, _i64, _p, _i64]),
    "jv_hip_nvq_vectors_upload": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "jv_hip_nvq_vectors_download": (_i, [_p, _p, _i64, _i64, _p, _p]),
    "jv_hip_nvq_vectors_count": (_i64, [_p]),
    "jv_hip_nvq_vectors_destroy": (_i, [_p]),
    "jv_hip_nvq_scores": (_i, [_p, _p, _p, _i, _i, _p, _i, _p]),
    "jv_hip_vectors_from_nvq": (_i, [_p, _p, C.POINTER(_p)]),
}

# the reference's per-pair SPI, exported unchanged (include/jvector_simd_compat.h)
_f = C.c_float
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_ubyte)
COMPAT_SIGNATURES = {
    "cosine_f32": (_f, [_fp, _sz, _fp, _sz, _sz]),
    "dot_product_f32": (_f, [_fp, _sz, _fp, _sz, _sz]),
    "euclidean_f32": (_f, [_fp, _sz, _fp, _sz, _sz]),
    "add_in_place_f32": (None, [_fp, _fp, _sz]),
    "add_scalar_in_place_f32": (None, [_fp, _f, _sz]),
    "sub_in_place_f32": (None, [_fp, _fp, _sz]),
    "sub_scalar_in_place_f32": (None, [_fp, _f, _sz]),
    "max_f32": (_f, [_fp, _sz]),
    "min_in_place_f32": (None, [_fp, _fp, _sz]),
    "assemble_and_sum_f32": (_f, [_fp, _i, _u8p, _i, _sz]),
    "assemble_and_sum_pq_f32": (_f, [_fp, _sz, _u8p, _i, _u8p, _i, _i]),
    "pq_decoded_cosine_similarity_f32": (_f, [_u8p, _i, _sz, _i, _fp, _fp, _f]),
    "calculate_partial_sums_dot_f32": (None, [_fp, _i, _sz, _i, _fp, _i, _fp]),
    "calculate_partial_sums_euclidean_f32": (None, [_fp, _i, _sz, _i, _fp, _i, _fp]),
    "calculate_partial_sums_self_magnitude_f32": (None, [_fp, _i, _sz, _i, _fp]),
    "nvq_quantize_8bit": (None, [_fp, _sz, _f, _f, _f, _f, _u8p]),
    "nvq_loss": (_f, [_fp, _sz, _f, _f, _f, _f, _i]),
    "nvq_uniform_loss": (_f, [_fp, _sz, _f, _f, _i]),
    "nvq_square_l2_distance_8bit": (_f, [_fp, _u8p, _sz, _f, _f, _f, _f]),
    "nvq_dot_product_8bit": (_f, [_fp, _u8p, _sz, _f, _f, _f, _f]),
    "nvq_cosine_8bit_packed": (C.c_int64, [_fp, _u8p, _sz, _f, _f, _f, _f, _fp]),
    "nvq_shuffle_query_in_place_8bit": (None, [_fp, _sz]),
    "jvector_simd_get_active_isa": (C.c_char_p, []),
    "jvector_simd_get_max_isa_env": (C.c_char_p, []),
}

# host-side format readers (include/jvector_formats.h)
JV_ODGI_MAX_LAYERS = 32
JV_ODGI_MAX_FEATURES = 8


class OdgiInfo(C.Structure):
    """struct jv_odgi_info (include/jvector_formats.h)."""
    _fields_ = [
        ("version", C.c_int32), ("dimension", C.c_int32), ("entry_node", C.c_int32), ("entry_level", C.c_int32),
        ("id_upper_bound", C.c_int32), ("n_layers", C.c_int32),
        ("layer_size", C.c_int32 * JV_ODGI_MAX_LAYERS), ("layer_degree", C.c_int32 * JV_ODGI_MAX_LAYERS),
        ("n_features", C.c_int32), ("feature_id", C.c_int32 * JV_ODGI_MAX_FEATURES),
        ("header_off", C.c_int64), ("l0_off", C.c_int64), ("record_stride", C.c_int64),
        ("inline_vectors_off", C.c_int64), ("fused_off", C.c_int64), ("neighbors_off", C.c_int64),
        ("pq_off", C.c_int64), ("pq_len", C.c_int64), ("pq_M", C.c_int32),
        ("upper_off", C.c_int64), ("hierarchy_off", C.c_int64), ("hierarchy_count", C.c_int32),
        ("separated_vectors_off", C.c_int64),
        ("nvq_off", C.c_int64), ("nvq_len", C.c_int64), ("nvq_S", C.c_int32), ("reserved0", C.c_int32),
        ("nvq_stride", C.c_int64), ("nvq_inline_off", C.c_int64), ("separated_nvq_off", C.c_int64),
    ]


_ip = C.POINTER(_i)
FORMAT_SIGNATURES = {
    "jv_fmt_pq_describe": (_i, [_p, _sz, C.POINTER(_sz), _ip, _ip, _ip, _ip, _ip, C.POINTER(C.c_float)]),
    "jv_fmt_pqvectors_describe": (_i, [_p, _sz, C.POINTER(_sz), C.POINTER(_i64), _ip, C.POINTER(_sz)]),
    "jv_fmt_odgi_describe": (_i, [_p, _sz, C.POINTER(OdgiInfo)]),
    "jv_fmt_odgi_read_l0": (_i, [_p, _sz, C.POINTER(OdgiInfo), _p, _p, _p]),
    "jv_fmt_odgi_read_level": (_i, [_p, _sz, C.POINTER(OdgiInfo), _i, _p, _p]),
    "jv_fmt_odgi_read_hierarchy_codes": (_i, [_p, _sz, C.POINTER(OdgiInfo), _p, _p]),
    "jv_fmt_nvq_describe": (_i, [_p, _sz, C.POINTER(_sz), _ip, _ip, _ip, C.POINTER(_sz), C.POINTER(_i64)]),
    "jv_fmt_nvq_read_mean": (_i, [_p, _sz, _p]),
    "jv_fmt_nvqvectors_describe": (_i, [_p, _sz, C.POINTER(_sz), C.POINTER(_i64), C.POINTER(_sz), C.POINTER(_i64)]),
    "jv_fmt_nvq_unpack": (_i, [_p, _sz, _i64, _i64, _i, _i, _p, _p]),
    "jv_fmt_odgi_read_nvq": (_i, [_p, _sz, C.POINTER(OdgiInfo), _p, _p]),
    "jv_fmt_xvecs_describe": (_i, [_p, _sz, C.POINTER(_i64), _ip]),
    "jv_fmt_xvecs_read": (_i, [_p, _sz, _p]),
}

_lib = None


def load():
    """Loads libjvector_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JVectorHipError(
                f"{LIB_PATH} not found: build it with `make -C jvector_amd/csrc` (there is no CPU fallback)")
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 and
        # dlopen them by absolute path, so loading /opt/rocm's copy first leaves torch with a second, GPU-less
        # runtime ("No HIP GPUs are available").  Importing torch first makes our NEEDED libamdhip64.so.7 resolve
        # (by soname) to the runtime torch already loaded.  Pure plumbing; set JVECTOR_HIP_NO_TORCH=1 to skip.
        if os.environ.get("JVECTOR_HIP_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for table in (SIGNATURES, COMPAT_SIGNATURES, FORMAT_SIGNATURES):
            for name, (res, args) in table.items():
                fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
                fn.restype = res
                fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    return load().jv_hip_last_error().decode("utf-8", "replace")


def check(status: int):
    if status == JV_OK:
        return
    msg = last_error() or f"libjvector_hip status {status}"
    if status == JV_ERR_INVALID:
        raise ValueError(msg)  # IllegalArgumentException / IndexOutOfBoundsException in the reference
    if status == JV_ERR_NO_DEVICE:
        raise NoDeviceError(msg)
    if status == JV_ERR_UNSUPPORTED:
        raise UnsupportedError(msg)
    if status == JV_ERR_OOM:
        raise MemoryError(msg)
    raise JVectorHipError(msg)
