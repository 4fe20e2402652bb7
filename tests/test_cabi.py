"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/crag_dense.h declares; argument errors come back as codes + messages, never exceptions."""
import ctypes
import re
from pathlib import Path

from cadence_rag_amd import _native

INCLUDE = Path(__file__).resolve().parent.parent / "include"
HEADER = INCLUDE / "crag_dense.h"


def _declared_functions():
    names = set()
    for header in (HEADER, INCLUDE / "crag_encoder.h"):
        text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
        names.update(re.findall(r"\b(crag_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_declares_expected_surface():
    names = _declared_functions()
    for must in ("crag_index_create", "crag_index_add", "crag_index_search", "crag_index_search_async",
                 "crag_index_destroy", "crag_last_error", "crag_merge_topk"):
        assert must in names


def test_library_exports_every_declared_symbol(native_lib):
    for name in _declared_functions():
        assert hasattr(native_lib, name), f"libcrag_dense.so does not export {name}"
    assert set(_native.SIGNATURES) == set(_declared_functions())


def test_header_constants_match_python_binding():
    text = HEADER.read_text()
    assert int(re.search(r"#define CRAG_MAX_K (\d+)", text).group(1)) == _native.CRAG_MAX_K
    assert int(re.search(r"#define CRAG_DIM (\d+)", text).group(1)) == _native.CRAG_DIM


def test_argument_errors_are_codes_not_crashes(native_lib):
    h = ctypes.c_void_p()
    assert native_lib.crag_index_create(0, 0, 10, ctypes.byref(h)) == -1  # CRAG_EINVAL: dim
    assert b"dim" in native_lib.crag_last_error()
    assert native_lib.crag_index_create(0, 1024, 0, ctypes.byref(h)) == -1  # capacity
    assert native_lib.crag_index_create(0, 1024, 10, None) == -1
    assert native_lib.crag_index_size(None) == -1
    assert native_lib.crag_index_destroy(None) == 0
    assert native_lib.crag_version().startswith(b"cadence-rag_amd")


def test_no_gpu_means_enodev_not_a_cpu_fallback(native_lib):
    if native_lib.crag_device_count() > 0:
        return  # GPU box: covered by the gpu tests
    h = ctypes.c_void_p()
    assert native_lib.crag_index_create(0, 1024, 10, ctypes.byref(h)) == -4  # CRAG_ENODEV
    assert not h


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", tmp_path / "nope.so")
    try:
        _native.load()
    except _native.NativeLibraryError as exc:
        assert "no CPU fallback" in str(exc)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_product_package_never_imports_the_oracle():
    pkg = Path(_native.__file__).resolve().parent
    for path in pkg.rglob("*.py"):
        src = path.read_text()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), path
    for path in (pkg / "csrc").glob("*"):
        if path.suffix in (".hip", ".h", ".cpp"):
            assert "oracle" not in path.read_text().lower(), path
