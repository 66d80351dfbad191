"""CPU-only boundary checks: the C-ABI library loads, exports every symbol include/*.h declares, and
fails loudly (no fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header, macro):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(macro + r"\s+[\w\s\*]+?\b(\w+)\s*\(", text)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    import jvector_amd
    if not os.path.exists(jvector_amd.LIB_PATH):
        g.build()
    return jvector_amd.load()


def test_every_declared_symbol_is_exported(lib):
    names = (declared_symbols("jvector_hip.h", "JV_API") + declared_symbols("jvector_simd_compat.h", "JVC_API")
             + declared_symbols("jvector_formats.h", "JV_API"))
    assert len(names) >= 68
    raw = ctypes.CDLL(os.path.join(ROOT, "jvector_amd", "libjvector_hip.so"))
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_python_signature_table_matches_header():
    import jvector_amd._lib as L
    hdr = set(declared_symbols("jvector_hip.h", "JV_API"))
    assert hdr == set(L.SIGNATURES), hdr ^ set(L.SIGNATURES)
    compat = set(declared_symbols("jvector_simd_compat.h", "JVC_API"))
    assert compat == set(L.COMPAT_SIGNATURES), compat ^ set(L.COMPAT_SIGNATURES)
    fmt = set(declared_symbols("jvector_formats.h", "JV_API"))
    assert fmt == set(L.FORMAT_SIGNATURES) and len(fmt) == 13, fmt ^ set(L.FORMAT_SIGNATURES)
    # the reference's own kernel list: 22 kernels + 2 getters (jvector_simd_kernel_list.h:36-62, jvector_simd.h:47,53)
    assert len(compat) == 24


def test_no_gpu_means_loud_failure(lib):
    import jvector_amd as J
    if J.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(J.NoDeviceError):
        J.HipContext(0)
    assert "no CPU fallback" in lib.jv_hip_last_error().decode()


def test_odgi_info_struct_matches_the_header():
    """ctypes mirror of struct jv_odgi_info: same field names in the same order as include/jvector_formats.h."""
    import jvector_amd._lib as L
    text = open(os.path.join(ROOT, "include", "jvector_formats.h")).read()
    body = re.search(r"typedef struct jv_odgi_info \{(.*?)\} jv_odgi_info;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        m = re.match(r"\s*(int32_t|int64_t)\s+(.*)", decl.strip(), flags=re.S)
        if m:
            for name in m.group(2).split(","):
                fields.append((re.sub(r"\[.*", "", name.strip()), m.group(1)))
    got = [(n, "int64_t" if (t is ctypes.c_int64) else "int32_t") for n, t in
           [(n, t._type_ if issubclass(t, ctypes.Array) else t) for n, t in L.OdgiInfo._fields_]]
    assert fields == got


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "jvector_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("nothing here imports oracle/", "").replace(
                    "imports oracle", ""), f


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's appendix is the complete list: a symbol added to the headers must show up there too"""
    import re
    text = ""
    for f in ("jvector_hip.h", "jvector_formats.h"):
        text += re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", f)).read(), flags=re.S)
    names = re.findall(r"JV_API\s+[\w\s\*]+?\b(jv_\w+)\s*\(", text)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in names if f"`{n}`" not in doc and n not in doc]
    assert len(names) > 80 and not missing, missing
