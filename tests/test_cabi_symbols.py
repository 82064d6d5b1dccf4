"""The C-ABI library builds for gfx950, loads on a CPU-only box and exports every symbol that
include/wdbx_hip.h declares; the Python binding declares the same set.  No compute calls."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "wdbx_hip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(wdbx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from wdbx_amd import _native

    path = _native.library_path()
    if not path.exists():
        subprocess.run(["make", "-C", str(ROOT / "wdbx-py_amd" / "csrc"), "all"], check=True)
    return path


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("wdbx_index_create", "wdbx_index_add", "wdbx_index_search", "wdbx_index_search_device",
                 "wdbx_index_search_sharded_device", "wdbx_comm_unique_id", "wdbx_last_error", "wdbx_hip_version"):
        assert must in syms
    # every entry point cites the reference interface it replaces
    head = HEADER.read_text()
    assert "indexing.py:1013" in head and "vector_store.py:323-345" in head


def test_integration_doc_names_every_declared_symbol():
    """every exported symbol with the reference interface it stands in for."""
    doc = (ROOT / "INTEGRATION.md").read_text()
    table = doc[doc.index("## F. Every symbol"):]
    missing = [s for s in declared_symbols() if f"`{s}`" not in table]
    assert not missing, missing


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(str(lib_path))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_matches_header(lib_path):
    from wdbx_amd import _native

    assert sorted(_native.SIGNATURES) == declared_symbols()
    lib = _native.load_library()
    assert lib.wdbx_hip_version() == 1


def test_code_object_targets_gfx950(lib_path):
    out = subprocess.run(["strings", "-n", "6", str(lib_path)], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_gpu_means_loud_failure_not_fallback(lib_path):
    import os

    from wdbx_amd import _native

    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    assert _native.device_count() == 0
    with pytest.raises(_native.HipBackendError):
        _native.NativeIndex(8)
    from wdbx_amd import WDBX

    with pytest.raises(_native.HipBackendError):
        WDBX(vector_dimension=4, data_dir=str(ROOT / "gpurun_out" / "_nogpu"), enable_plugins=False)


def test_product_never_imports_the_oracle():
    for py in (ROOT / "wdbx-py_amd").rglob("*.py"):
        assert "wdbx_oracle" not in py.read_text(), py
    for src in (ROOT / "wdbx-py_amd" / "csrc").glob("*.hip"):
        assert "oracle" not in src.read_text().lower().replace("// oracle", ""), src
