"""CPU checks of the C-ABI boundary: the library loads and exports every symbol include/ape_hip.h declares,
and the ctypes table mirrors the header (no compute calls without a GPU)."""
import ctypes
import os
import re

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


def test_ctypes_table_matches_header():
    from autoposeestimation_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    text = _header_text()
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, text)
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), name


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
