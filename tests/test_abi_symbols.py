"""The C-ABI library loads and exports every symbol include/isscabac.h declares (no GPU needed), and the
host-side pure functions behave like the reference's init code."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from isscabac_amd import build
    return ctypes.CDLL(build.build_lib())


def test_all_declared_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "isscabac.h")).read()
    names = sorted(set(re.findall(r"\b(isscabac_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    lib.isscabac_abi_version.restype = ctypes.c_int
    assert lib.isscabac_abi_version() == 1


def test_init_maps_match_oracle(oracle):
    import isscabac_amd as ia
    p0 = np.arange(256) / 255.0
    assert ia.init_by_prob(p0).tolist() == oracle.init_by_prob(p0).tolist()
    t = [0, 1, 20, 5, 0, 62, 9, 1, 0]
    assert ia.init_by_state(t).tolist() == oracle.init_by_state(t).tolist()
    with pytest.raises(ia.EngineError):
        ia.init_by_state([0, 0, 63])  # the reference's own decoder indexes out of bounds for state 63
    with pytest.raises(ia.EngineError):
        ia.init_by_prob([1.5])
    assert ia.stream_bound(0) == 2 and ia.stream_bound(4096) == (6 * 4096 + 9 + 7) // 8


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import isscabac_amd as ia
    with pytest.raises(ia.EngineError):
        ia.Engine(0)
