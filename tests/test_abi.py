"""The C-ABI library loads and exports every symbol include/fandom_search.h
declares; struct layouts on the Python side match the header.  No compute
calls: this runs without a GPU."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from fandom_search_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fandom_search.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared_functions()
    for want in ("fs_index_create", "fs_corpus_create", "fs_search_corpus", "fs_search",
                 "fs_index_destroy", "fs_corpus_destroy", "fs_version", "fs_strerror"):
        assert want in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.lib_path()):
        _lib.build()
    lib = C.CDLL(_lib.lib_path())
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == _declared_functions()


def test_version_and_strerror_without_a_device():
    L = _lib.load()
    assert L.fs_version() == 1
    assert L.fs_strerror(abi.FS_E_CAPACITY) == b"row buffer too small"
    assert L.fs_strerror(abi.FS_OK) == b"ok"


def test_struct_layouts():
    assert C.sizeof(abi.FsConfig) == 48
    assert abi.FsConfig.distance_threshold.offset == 40
    assert C.sizeof(abi.FsStats) == 64
    assert C.sizeof(abi.FsIndexInfo) == 72
    assert abi.ROW_DTYPE.itemsize == 32
    assert [abi.ROW_DTYPE.fields[n][1] for n in abi.ROW_DTYPE.names] == [0, 4, 8, 12, 16, 24]


def test_bad_config_is_rejected_before_any_device_call():
    L = _lib.load()
    cfg = abi.make_config()
    cfg.struct_size = 12
    h = C.c_void_p()
    rc = L.fs_index_create(C.byref(cfg), None, None, None, 0, None, 0, None, C.byref(h))
    assert rc == abi.FS_E_INVALID and not h.value
    assert b"ABI" in L.fs_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "_PKG", str(tmp_path))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fandom_search_amd/ may
    import, load or link it."""
    pkg = os.path.join(ROOT, "fandom_search_amd")
    bad = re.compile(r"^\s*(import|from)\s+oracle\b|liboracle|\bfo_[a-z_]+\s*\(|c_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), os.path.join(dirpath, f)
