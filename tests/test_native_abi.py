"""The C-ABI library builds, loads, and exports every symbol include/vpt_hip.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import vpt_amd
    from vpt_amd import _native
    return _native.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "vpt_hip.h")).read()
    names = re.findall(r"^(?:int|long|const char\*)\s+(vpt_\w+)\s*\(", hdr, flags=re.M)
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n


def test_ctypes_signatures_cover_header(lib):
    from vpt_amd import _native
    hdr = open(os.path.join(ROOT, "include", "vpt_hip.h")).read()
    for name, args in _native.SIGNATURES.items():
        m = re.search(r"(?:int|long)\s+" + name + r"\s*\(([^;]*)\)\s*;", hdr, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(args), name


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.vpt_version()
    assert isinstance(lib.vpt_last_error(), bytes)


def test_abi_number_matches_header_and_binding(lib, monkeypatch):
    """vpt_abi_version() == VPT_HIP_ABI of the header == the number _native.py was written against; a library that reports another
    number (a stale build: argument lists differ) is refused at load time, before any call."""
    from vpt_amd import _native
    hdr = open(os.path.join(ROOT, "include", "vpt_hip.h")).read()
    abi_hdr = int(re.search(r"#define\s+VPT_HIP_ABI\s+(\d+)", hdr).group(1))
    assert lib.vpt_abi_version() == abi_hdr == _native.ABI_VERSION
    monkeypatch.setattr(_native, "_libs", {})
    monkeypatch.setattr(_native, "ABI_VERSION", abi_hdr + 1)
    with pytest.raises(_native.NativeLibraryError, match="C-ABI version"):
        _native.load("bf16")


def test_missing_library_fails_loudly(monkeypatch):
    import vpt_amd  # noqa: F401
    from vpt_amd import _native
    monkeypatch.setattr(_native, "_libs", {})
    monkeypatch.setattr(_native, "_LIB_PATHS", {"bf16": "/nonexistent/libvpt_hip.so", "fp16": "/nonexistent/libvpt_hip_f16.so"})
    for fmt in ("bf16", "fp16"):
        with pytest.raises(_native.NativeLibraryError):
            _native.load(fmt)


def test_both_operand_formats_built_with_the_same_abi(lib):
    """libvpt_hip_f16.so (precision="fp16") is the same sources built with -DVPT_OPERAND_F16: identical export list."""
    from vpt_amd import _native
    f16 = _native.load("fp16")
    assert lib.vpt_operand_format() == b"bf16" and f16.vpt_operand_format() == b"fp16"
    for name in _native.SIGNATURES:
        assert hasattr(f16, name), name
