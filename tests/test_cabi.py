"""C-ABI surface: the shared library builds for gfx950 without a GPU, loads, exports every symbol declared in
include/hific_hip.h, the ctypes signature table covers exactly those symbols, and the product path refuses CPU
tensors loudly (no CPU / PyTorch fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "hific_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hific_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hific_hip.h but not exported"
    assert lib.hific_version() >= 100


def test_binding_table_matches_header(hific):
    from hific_amd import lib
    assert sorted(lib.SIGNATURES) == _header_symbols()


def test_ops_refuse_cpu_tensors(hific):
    from hific_amd import ops, lib
    x = torch.zeros(1, 4, 8, 8)
    w = torch.zeros(4, 4, 3, 3)
    with pytest.raises(lib.HificError):
        ops.conv2d(x, w, None, 1, (1, 1, 1, 1))
    with pytest.raises(lib.HificError):
        ops.channel_norm(x, torch.ones(1, 4, 1, 1), torch.zeros(1, 4, 1, 1))


def test_missing_library_is_loud(hific, monkeypatch):
    from hific_amd import lib
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libhific_hip.so")
    with pytest.raises(ImportError):
        lib._load()


def test_workspace_queries(built_lib):
    lib = ctypes.CDLL(built_lib)
    lib.hific_conv2d_ws_bytes.restype = ctypes.c_size_t
    b = lib.hific_conv2d_ws_bytes(16, 960, 16, 16, 960, 3, 3, 1, 1, 1, 1, 1, 1)
    assert 16 * 2 ** 20 < b < 2 ** 31


def test_host_library_exports_every_declared_symbol(built_lib):
    """libhific_host.so (CPU-side table construction, include/hific_host.h) is built next to the device library."""
    path = os.path.join(os.path.dirname(built_lib), "libhific_host.so")
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hific_host.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(hific_[a-z0-9_]+)\s*\(", txt)))
    assert syms == ["hific_build_cdf_rows", "hific_host_version", "hific_pmf_to_quantized_cdf", "hific_rans_decode",
                    "hific_rans_decode_vec", "hific_rans_encode", "hific_rans_encode_vec"]
    for s in syms:
        assert hasattr(lib, s)
