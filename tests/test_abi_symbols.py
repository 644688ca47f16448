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
    import torch  # noqa: F401  (torch's bundled HIP runtime first: a later GPU test in the same process must share it)
    from isscabac_amd import build
    return ctypes.CDLL(build.build_lib())


def test_all_declared_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "isscabac.h")).read()
    names = sorted(set(re.findall(r"\b(isscabac_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    lib.isscabac_abi_version.restype = ctypes.c_int
    assert lib.isscabac_abi_version() == 2


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


def test_stream_file_container_round_trip(oracle, tmp_path):
    """one-file-per-stream <-> packed payload+offsets; the files are what the reference writes"""
    from isscabac_amd.wrapper import read_stream_files, write_stream_files
    bins, ctx = oracle.synth(0xCABAC002, 0, 5, 300, 3)
    payload, offsets = oracle.encode_batch(bins, ctx, np.zeros(3, np.uint8))
    paths = [str(tmp_path / ("s%d.bin" % i)) for i in range(5)]
    sizes = write_stream_files(payload, offsets, paths)
    assert sizes == np.diff(offsets.astype(np.int64)).tolist()
    for i, p in enumerate(paths):
        assert open(p, "rb").read() == oracle.encode_stream(bins[i], ctx[i], np.zeros(3, np.uint8)).tobytes()
    p2, o2 = read_stream_files(paths)
    assert p2.tobytes() == payload.tobytes() and (o2 == offsets).all()
