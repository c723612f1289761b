"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/dpf.h declares (no compute is attempted here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from similaritysearchbyrdf_amd import _lib
    names = declared_functions("dpf.h")
    assert len(names) >= 18
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names          # the Python binding covers the whole ABI


def test_struct_layouts_match_header():
    from similaritysearchbyrdf_amd import _lib
    assert C.sizeof(_lib.DpfConfig) == 20 * 4 + 2 * 8
    assert C.sizeof(_lib.DpfStats) == 8 * 8 + 5 * 8 + 6 * 8 + 3 * 8 + 18 * 8


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    from similaritysearchbyrdf_amd._lib import DpfIndex, DpfError
    from similaritysearchbyrdf_amd import synthetic
    with pytest.raises(DpfError) as ei:
        DpfIndex(synthetic.make_gaussian_family(8, 1))
    assert "no HIP device" in str(ei.value) or "hip" in str(ei.value).lower()


def test_bad_configurations_are_rejected_before_touching_the_device():
    from similaritysearchbyrdf_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    base = dict(family=0, dim=8, num_functions=32, num_tables=1, chain_length=32, partition_bits=3, bucket_bits=28,
                dir_node_size=32, bucket_overflow=500, type_of_index=0, exclude_self=0, metric=0, top_k=10, dtype=0,
                device=0, flags=0, row_begin=0, row_end=-1)
    for bad in (dict(dir_node_size=48), dict(chain_length=16), dict(top_k=0), dict(top_k=65), dict(family=7),
                dict(dtype=3), dict(bucket_bits=40), dict(partition_bits=9), dict(bucket_overflow=0), dict(dim=0)):
        cfg = _lib.DpfConfig(**dict(base, **bad))
        assert L.dpf_create(C.byref(cfg), C.byref(h)) == _lib.DPF_EINVAL, bad
        assert L.dpf_last_error(None)
    assert L.dpf_create(None, C.byref(h)) == _lib.DPF_EINVAL
