"""The C-ABI shared library loads and exports every symbol include/bwtgpu.h declares, and
(on a machine without a GPU) every compute entry point fails loudly instead of falling
back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

from bwt_amd import api, build
from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return api.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "bwtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bwtgpu_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(api.SYMBOLS) == names  # the Python binding covers the whole header


def test_size_helpers_need_no_gpu(lib):
    assert lib.bwtgpu_num_blocks(10, 3) == 4
    assert lib.bwtgpu_num_blocks(0, 3) == 0
    assert lib.bwtgpu_forward_output_bytes(152089, 131072) == 152091
    m = (ctypes.c_uint32 * 8)(0xFFFFFFFF, 0x7)
    assert lib.bwtgpu_lowest_clear_bit(m) == 35
    full = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    assert lib.bwtgpu_lowest_clear_bit(full) == api.ERR_NO_FREE_BYTE


def _no_gpu():
    try:
        import torch

        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_no_cpu_fallback(lib):
    """Without a device the product path must refuse, not compute on the host."""
    with pytest.raises(api.BwtGpuError) as e:
        api.forward_batch(b"banana", 6, ord("~"))
    assert e.value.code == api.ERR_NO_DEVICE
    with pytest.raises(api.BwtGpuError):
        api.inverse_batch(b"bnn~aaa", 6, ord("~"))
    with pytest.raises(api.BwtGpuError):
        api.find_unused_byte(b"abc")
    assert api.bwt_forward(b"", ord("~")) == b"~"  # the empty block needs no arithmetic at all


def test_product_never_imports_the_oracle():
    """bwt_amd (and the C++ host layer) must not reference oracle/ in any form."""
    for base in ("bwt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    text = open(os.path.join(dirpath, fn), errors="replace").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, fn)
