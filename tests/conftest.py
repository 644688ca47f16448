import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "cabac_golden.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import pyoracle
    if not (pyoracle.have_reference() or os.path.isdir("/root/reference/CABAC")):
        pytest.skip("reference build (oracle/_ref) not available here")
    return pyoracle.Reference()


def unhex(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8)


def unbits(s, n):
    return np.unpackbits(unhex(s))[:n] if n else np.zeros(0, np.uint8)


@pytest.fixture(scope="session")
def golden():
    with open(GOLDEN) as f:
        doc = json.load(f)
    for s in doc["streams"]:
        s["bins_a"] = unbits(s["bins"], s["n"])
        s["ctx_a"] = unhex(s["ctx"])
        s["ctx_init_a"] = unhex(s["ctx_init"])
        s["payload_a"] = unhex(s["payload"])
        s["ctx_final_a"] = unhex(s["ctx_final"])
    return doc


@pytest.fixture(scope="session", params=["throughput", "latency", "latency_single"])
def engine(request):
    """The HIP engine through its C ABI (GPU tests only), once per kernel instance: the test batches are small, so under
    automatic tuning every one of them would run the wave-pair kernels only.  throughput: k_encode / k_decode<CM,false,*>;
    latency: k_encode_pair / k_decode_pair (<= 64 contexts: what AUTO runs up to 128 Ki streams -- an 8-GPU job's per-GPU
    share of config 3, config 2's shape), k_*<CM,true,false> beyond; latency_single: k_encode / k_decode<CM,true,false> for
    every context count."""
    import torch  # noqa: F401  (load torch's HIP runtime first so both share one libamdhip64)
    from isscabac_amd import Engine
    eng = Engine(device=0)
    eng.set_tuning(request.param)
    return eng
