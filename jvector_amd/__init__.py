"""jvector_amd — MI355X-native distance / quantization engine for JVector's hot path.

Product code: HIP kernels + C ABI in jvector_amd/csrc (built into jvector_amd/libjvector_hip.so) and the thin
ctypes host mirror in jvector_amd/engine.py.  Nothing here imports oracle/ and nothing computes on the CPU.
"""
from ._lib import JVectorHipError, NoDeviceError, UnsupportedError, LIB_PATH, load  # noqa: F401
from .engine import (  # noqa: F401
    ApproximateScoreFunction,
    DecoderKind,
    FlatSearcher,
    FusedPQ,
    FusedScoreFunction,
    GraphIndex,
    GraphSearcher,
    SearchResult,
    HipContext,
    NVQuantization,
    NVQVectors,
    PQBuildScoreProvider,
    PQVectors,
    ProductQuantization,
    QueryTables,
    VectorSet,
    VectorSimilarityFunction,
    device_count,
    topk,
)
