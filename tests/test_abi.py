"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/kmcp_gpu.h declares, and refuses to compute without a GPU (no silent fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    from kmcp_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        g.build()
    return lib


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "kmcp_gpu.h")).read()
    declared = set(re.findall(r"\b(kmcpg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.load()
    for s in declared:
        assert getattr(lib, s) is not None


def test_struct_layouts_match_header(L):
    import ctypes as C
    assert C.sizeof(L.Hit) == 12
    assert C.sizeof(L.Match) == 56
    assert C.sizeof(L.Opts) == 16
    assert C.sizeof(L.Params) == 64
    assert C.sizeof(L.SynthSpec) == 64


def test_no_cpu_fallback(L, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.KmcpGpuError) as e:
        L.Database.open(str(tmp_path))
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kmcp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("parity oracle", ""), f
