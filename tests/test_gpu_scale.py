"""Scale checks of the batch path on one MI355X ("offsets must be 64-bit"; BASELINE config 5's
per-GPU share at full size).  Device round trips over every stream plus sampled streams byte for byte against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _round_trip_and_sample(S, B, C, seed, ep, sample, cap_per_stream):
    torch = _torch()
    from isscabac_amd import Engine
    from oracle.pyoracle import Oracle
    dev = torch.device("cuda:0")
    eng, orc = Engine(0), Oracle()
    bins = torch.empty((S, B), dtype=torch.uint8, device=dev)
    ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
    eng.synth_dev(seed, 0, S, B, B, C, ep, bins.data_ptr(), ctx.data_ptr())
    init = np.zeros(C, np.uint8)
    cap = S * cap_per_stream
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
    total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
    off = offsets.cpu().numpy().astype(np.uint64)
    assert off[0] == 0 and off[-1] == total and (np.diff(off.astype(np.int64)) > 0).all()
    out = torch.empty((S, B), dtype=torch.uint8, device=dev)
    status = torch.empty((S,), dtype=torch.int32, device=dev)
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
    assert int(status.abs().sum()) == 0
    assert torch.equal(out, bins)
    del out, status
    for s in sample:
        hb, hc = orc.synth(seed, s, 1, B, C, ep)
        want = orc.encode_stream(hb[0], hc[0], init)
        got = payload[int(off[s]):int(off[s + 1])].cpu().numpy()
        assert got.tobytes() == want.tobytes(), s
    eng.close()
    return total, off


def test_config5_per_gpu_share_at_full_size():
    """BASELINE.json config 5 on one of its 8 GPUs: 1 Mi streams x 8 Ki bins, 50 % bypass bins -- the mixed-chunk path of
    both kernels at full occupancy.  Device round trip over all 8.6 G bins, 133 strided streams against the oracle."""
    S = 1 << 20
    sample = list(range(0, S, S // 128)) + [S - 1, S - 2, 63, 64, 65]
    total, _off = _round_trip_and_sample(S, 8192, 64, 0xCABAC005, True, sample, 8192 // 8 + 8192 // 16 + 64)
    assert 0.11 * S * 8192 < total < 0.14 * S * 8192  # ~1.0 bit per bin: half of them bypass bins


def test_payload_beyond_4_gib_offsets_are_64_bit():
    """5 Mi streams x 8 Ki bins with bypass bins: ~5.4 GB of payload, so streams start beyond 2^32 -- offsets, the scan,
    the gather and the decoder's payload addressing all have to be 64-bit.  Needs ~175 GB of free HBM."""
    torch = _torch()
    free, _tot = torch.cuda.mem_get_info(0)
    if free < 180e9:
        pytest.skip("needs 180 GB of free HBM, %.0f GB are free" % (free / 1e9))
    S = 5 << 20
    # sampled: the batch's ends and 64 streams from the tail, which lies beyond the 4 GiB mark (checked below)
    total, off = _round_trip_and_sample(S, 8192, 64, 0xCABAC005, True,
                                        [0, 1, S // 2, S - 1] + list(range(S - 100_000, S - 100_000 + 64)), 1100)
    assert total > (1 << 32)
    assert int(off[S - 100_000]) > (1 << 32)  # the sampled run does start beyond 4 GiB


def test_streams_of_the_maximum_length():
    """ISSCABAC_MAX_PITCH: 64 streams (one full wave) of 2^25 bins each -- the longest rows the batch calls take; the lanes' 32-bit
    row offsets reach 63 * 2^25 + 2^25 - 16 < 2^32, a stream's payload ~3 MB, its slot ~25 MB.  Device round trip over all
    2.1 G bins, the first and the last stream byte for byte against the oracle (33 M bins each)."""
    S, B = 64, 1 << 25
    total, off = _round_trip_and_sample(S, B, 64, 0xCABAC003, False, [0, S - 1], B // 8 + 64)
    assert 0.09 * S * B < total < 0.11 * S * B


@pytest.mark.parametrize("S,B,C,seed,tuning,enc_inst,dec_inst", [
    # config 3's per-GPU share of an 8-GPU strong-scaled job: what `bench.py --gpus 8` runs on every rank
    (131072, 4096, 64, 0xCABAC003, "auto", "k_encode_pair", "k_decode_pair"),
    # config 2's shape
    (65536, 1024, 3, 0xCABAC002, "auto", "k_encode_pair", "k_decode_pair"),
    # the one-wave latency-oriented instances at the share's size (AUTO's decoder there until the pairs' waves were pinned to SIMDs)
    (131072, 4096, 64, 0xCABAC003, "latency_single", "k_encode<1,true,false>", "k_decode<1,true,false>"),
])
def test_per_gpu_share_under_auto_tuning_against_the_oracle(S, B, C, seed, tuning, enc_inst, dec_inst):
    """AUTO's own kernel choice up to 128 Ki streams -- the wave pairs, four waves started per pair and two kept on the SIMDs a
    per-CU ticket names -- and the one-wave latency-oriented instances, at the real launch sizes: device round trip over every
    stream, 133 sampled streams byte for byte against the oracle, and 320 corrupted streams (random bit flips; 64 of them made
    to start with 0xFF, the reference's register-overflow case) whose decoded bins AND status words must equal the oracle's
    (ref: CABAC_ArithmeticDecoder.cpp:73-190)."""
    torch = _torch()
    from isscabac_amd import Engine
    from oracle.pyoracle import Oracle
    dev = torch.device("cuda:0")
    eng, orc = Engine(0), Oracle()
    eng.set_tuning(tuning)
    bins = torch.empty((S, B), dtype=torch.uint8, device=dev)
    ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
    eng.synth_dev(seed, 0, S, B, B, C, False, bins.data_ptr(), ctx.data_ptr())
    init = np.zeros(C, np.uint8)
    cap = S * (B // 8 + 64)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
    total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
    assert eng.last_instance("encode") == enc_inst
    off = offsets.cpu().numpy().astype(np.uint64)
    assert off[0] == 0 and off[-1] == total and (np.diff(off.astype(np.int64)) > 0).all()
    out = torch.empty((S, B), dtype=torch.uint8, device=dev)
    status = torch.empty((S,), dtype=torch.int32, device=dev)
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
    assert eng.last_instance("decode") == dec_inst
    assert int(status.abs().sum()) == 0
    assert torch.equal(out, bins)
    sample = list(range(0, S, S // 128)) + [S - 1, S - 2, 63, 64, 65]
    for s in sample:
        hb, hc = orc.synth(seed, s, 1, B, C)
        want = orc.encode_stream(hb[0], hc[0], init)
        got = payload[int(off[s]):int(off[s + 1])].cpu().numpy()
        assert got.tobytes() == want.tobytes(), s
    # corrupted streams, spread over the launch (every wave position: 320 streams at a stride coprime to 64)
    rng = np.random.default_rng(6)
    host = payload[:total].cpu().numpy().copy()
    bad_ids = [(7 + 409 * k) % S for k in range(320)]
    for k, s in enumerate(bad_ids):
        a, b = int(off[s]), int(off[s + 1])
        for _ in range(1 + k % 3):
            host[a + int(rng.integers(0, b - a))] ^= 1 << int(rng.integers(0, 8))
        if k % 5 == 0:
            host[a] = 0xFF
    payload[:total] = torch.from_numpy(host).to(dev)
    out.fill_(7)
    status.fill_(9)
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
    assert eng.last_instance("decode") == dec_inst
    st = status.cpu().numpy().astype(np.uint32)
    flagged = 0
    for s in bad_ids:
        hb, hc = orc.synth(seed, s, 1, B, C)
        wbins, wst = orc.decode_stream(host[int(off[s]):int(off[s + 1])], hc[0], init)
        assert st[s] == wst, ("status", s)
        assert (out[s].cpu().numpy() == wbins).all(), ("bins", s)
        flagged += wst != 0
    assert flagged > 100
    # ... and nobody else was touched
    ok = np.ones(S, bool)
    ok[bad_ids] = False
    assert not st[ok].any()
    good = torch.from_numpy(np.flatnonzero(ok)).to(dev)
    assert torch.equal(out[good], bins[good])
    eng.close()


def test_config5_as_one_job_on_one_gpu():
    """BASELINE.json config 5 as specified -- 8 Mi streams x 8 Ki bins, 50 % bypass bins -- run as ONE job on this one GPU:
    bench.py codes it in 8 launches of 1 Mi streams per step (~220 GB of HBM), decodes everything back and compares every bin
    (verified_bit_exact) and 4 096 streams byte for byte with the CPU coder.  What 8 GPUs would split, one GPU does in turn."""
    import json
    import os
    import subprocess
    import sys
    torch = _torch()
    torch.cuda.empty_cache()  # (this process's cached blocks of the tests above would not leave room for the child)
    free, _tot = torch.cuda.mem_get_info(0)
    if free < 235e9:
        pytest.skip("needs 235 GB of free HBM, %.0f GB are free" % (free / 1e9))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config5", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, cwd=root, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["verified_bit_exact"] is True and d["cpu_baseline"]["gpu_streams_checked_byte_exact"] == 4096
    assert d["config"]["streams_total"] == 8 << 20 and d["config"]["launches_per_step"] == 8 and d["config"]["bins_per_stream"] == 8192
    assert "8388608 streams x 8192 bins, 64 contexts, 50% bypass" in d["metric"]
    assert 0.8 < d["config"]["bits_per_bin"] < 1.15  # half the bins cost one bit each, the context-coded half ~0.72
