"""CPU checks of the C-ABI boundary: the library loads and exports every symbol include/ape_hip.h declares,
and the ctypes table mirrors the header (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _header_text():
    text = open(os.path.join(REPO, "include", "ape_hip.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _header_symbols():
    return sorted(set(re.findall(r"\b(ape_[a-z0-9_]+)\s*\(", _header_text())))


def test_library_exports_every_declared_symbol():
    from autoposeestimation_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: __graft_entry__.build()"
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert "ape_knn_f32" in syms
    for s in syms:
        assert hasattr(h, s), "libape_hip.so lacks %s declared in include/ape_hip.h" % s


def _ctype_of(decl):
    """ctypes type of one C parameter declaration of include/ape_hip.h"""
    d = re.sub(r"\b(const|restrict|__restrict__)\b", " ", decl).strip()
    if "*" in d or "[" in d:
        return ctypes.c_void_p
    base = " ".join(d.split()[:-1]) if len(d.split()) > 1 else d      # drop the parameter name
    table = {"int": ctypes.c_int, "int32_t": ctypes.c_int, "unsigned int": ctypes.c_uint, "unsigned": ctypes.c_uint, "long": ctypes.c_long,
             "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float, "double": ctypes.c_double}
    assert base in table, "unknown C type %r in %r" % (base, decl)
    return table[base]


def test_ctypes_table_matches_header():
    """every prototype of the header: same symbol set, same arity and -- parameter by parameter -- the same C type as the ctypes table
    (pointers of any pointee bind as void*; int64_t and long are both 8 bytes on this ABI)"""
    from autoposeestimation_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    text = _header_text()
    same = lambda a, b: a is b or ({a, b} <= {ctypes.c_long, ctypes.c_int64})  # noqa: E731
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, text)
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), name
        for i, (decl, ct) in enumerate(zip(args, argtypes)):
            want = _ctype_of(decl)
            assert same(want, ct), "%s argument %d: header says %r (%s), ctypes table has %s" % (name, i, decl.strip(), want.__name__, ct.__name__)
    rest = {"ape_last_error": ctypes.c_char_p}
    for name, rt in _lib._RESTYPES.items():
        m = re.search(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b%s\s*\(" % name, text)
        want = rest.get(name) or _ctype_of(m.group(1).strip() + " x")
        assert same(want, rt), "%s return type: header %r vs ctypes %s" % (name, m.group(1).strip(), rt.__name__)


def test_abi_version_callable_without_gpu():
    from autoposeestimation_amd import _lib
    assert _lib.lib().ape_abi_version() >= 1


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under autoposeestimation_amd/ may reference it."""
    pkg = os.path.join(REPO, "autoposeestimation_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_reference_ffi_binding_compiles_links_and_imports():
    """The reference's one native interface (`knn_pytorch.knn`, DenseFusion/lib/knn/src/knn.h:12, vision.cpp:3-5) as a compiled pybind11
    module over the C ABI: builds with the image's g++ + torch headers + -lape_hip (no GPU), imports, exports `knn`, and refuses host
    tensors instead of computing on the CPU (compute parity: tests/test_gpu_knn.py)."""
    import torch
    from autoposeestimation_amd.DenseFusion.lib.knn import build_ext, load_compiled
    path = build_ext.build()
    assert os.path.exists(path)
    mod = load_compiled()
    assert mod is not None and callable(mod.knn)
    with pytest.raises(RuntimeError, match="GPU"):
        mod.knn(torch.zeros(1, 3, 4), torch.zeros(1, 3, 5), torch.zeros(1, 1, 5, dtype=torch.int64))
