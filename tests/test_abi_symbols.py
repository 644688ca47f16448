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


def test_native_programs_compile_against_the_header_and_fail_loudly_without_gpu(tmp_path):
    """examples/simple_cabac.cpp and tests/perf/per_bin_native.cpp use the C ABI from compiled code: they must build against
    include/isscabac.h + the library as they are, and -- the product has no CPU coding path -- stop with the engine's error
    where there is no GPU (on a GPU box tests/test_gpu_parity.py runs the example to its known answer)"""
    import subprocess
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "isscabac_amd", "lib")
    for src in (os.path.join(root, "examples", "simple_cabac.cpp"), os.path.join(root, "tests", "perf", "per_bin_native.cpp")):
        exe = str(tmp_path / os.path.basename(src)[:-4])
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), src, "-o", exe, "-L", lib,
                               "-lisscabac_hip", "-Wl,-rpath," + lib, "-Wl,--allow-shlib-undefined"])
        if not torch.cuda.is_available():
            env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
            out = subprocess.run([exe, "100"], capture_output=True, text=True, timeout=120, env=env)
            assert out.returncode != 0 and "isscabac_create" in out.stderr, out.stdout + out.stderr


def test_stream_file_container_round_trip(oracle, tmp_path):
    """one-file-per-stream <-> packed payload+offsets through the C ABI; the files are what the reference
    writes.  Host-only calls: they run without a GPU."""
    import isscabac_amd.engine as eng
    bins, ctx = oracle.synth(0xCABAC002, 0, 5, 300, 3)
    payload, offsets = oracle.encode_batch(bins, ctx, np.zeros(3, np.uint8))
    paths = [eng.stream_file_path(str(tmp_path / "s"), i) for i in range(5)]
    assert paths[3].endswith("s000003.bin")
    eng.write_stream_files(payload, offsets, paths)
    assert [os.path.getsize(p) for p in paths] == np.diff(offsets.astype(np.int64)).tolist()
    for i, p in enumerate(paths):
        assert open(p, "rb").read() == oracle.encode_stream(bins[i], ctx[i], np.zeros(3, np.uint8)).tobytes()
    p2, o2 = eng.read_stream_files(paths)
    assert p2.tobytes() == payload.tobytes() and (o2 == offsets).all()
    with pytest.raises(eng.EngineError, match="cannot read"):
        eng.read_stream_files(paths + [str(tmp_path / "missing.bin")])
    with pytest.raises(eng.EngineError, match="cannot write"):
        eng.write_stream_files(payload, offsets, [str(tmp_path / "no_dir" / "x.bin")] * 5)
    # an empty stream in the middle (zero-length file) survives the round trip
    o3 = np.array([0, 4, 4, 9], np.uint64)
    eng.write_stream_files(payload[:9], o3, paths[:3])
    p4, o4 = eng.read_stream_files(paths[:3])
    assert p4.tobytes() == payload[:9].tobytes() and (o4 == o3).all()
    # nothing but empty files, and no files at all: a NULL payload buffer is fine when no byte has to go there (what a
    # MEX caller's 0x1 matrix hands over)
    import ctypes as C
    L = eng._load()
    for k in (2, 0):
        for q in paths[:k]:
            open(q, "wb").close()
        arr = (C.c_char_p * max(1, k))(*[q.encode() for q in paths[:k]])
        off = (C.c_uint64 * (k + 1))(*([7] * (k + 1)))
        tot = C.c_uint64(99)
        assert L.isscabac_read_stream_files(arr, k, None, 0, off, C.byref(tot)) == 0
        assert list(off) == [0] * (k + 1) and tot.value == 0


def test_stream_files_interoperate_with_the_reference_decoder(oracle, reference, tmp_path):
    """files written by isscabac_write_stream_files are decoded by the REAL reference classes (oracle/_ref), and files the
    reference's encoder wrote come back through isscabac_read_stream_files as the batch the oracle produces"""
    import isscabac_amd.engine as eng
    S, B, C = 6, 700, 5
    bins, ctx = oracle.synth(0xCABAC003, 3, S, B, C)
    init = np.arange(C, dtype=np.uint8) * 9
    payload, offsets = oracle.encode_batch(bins, ctx, init)
    paths = [eng.stream_file_path(str(tmp_path / "w"), i) for i in range(S)]
    eng.write_stream_files(payload, offsets, paths)
    for s in range(S):  # the reference reads our files
        got, status, _ = reference.decode_stream(np.fromfile(paths[s], np.uint8), ctx[s], init)
        assert status == 0 and (got == bins[s]).all()
    for s in range(S):  # ... and we read the reference's
        p_ref, _ = reference.encode_stream(bins[s], ctx[s], init)
        p_ref.tofile(paths[s])
    p2, o2 = eng.read_stream_files(paths)
    assert p2.tobytes() == payload.tobytes() and (o2 == offsets).all()


def test_iss_side_info_container_round_trip(tmp_path):
    import isscabac_amd.engine as eng
    rng = np.random.default_rng(8)
    cW, cH, Q = rng.random(8), rng.random(6), rng.normal(size=(2, 3))
    xw, xh = rng.integers(0, 256, 23).astype(np.uint8), rng.integers(0, 256, 23).astype(np.uint8)
    w, h = rng.integers(0, 256, 901).astype(np.uint8), rng.integers(0, 256, 77).astype(np.uint8)
    fn = str(tmp_path / "iss.bin")
    eng.iss_container_write(fn, cW, cH, Q, xw, xh, w, h)
    assert os.path.getsize(fn) == 8 + 32 + 16 + 8 * (8 + 6 + 6) + 46 + 901 + 77
    d = eng.iss_container_read(fn)
    assert (d["cW"] == cW).all() and (d["cH"] == cH).all() and (d["Q"] == Q).all()
    assert (d["xw"] == xw).all() and (d["xh"] == xh).all() and (d["w_stream"] == w).all() and (d["h_stream"] == h).all()
    # equalProb: no context initialisations are transmitted (ISS.m:198-200)
    eng.iss_container_write(fn, cW, cH, Q, [], [], w, [])
    d = eng.iss_container_read(fn)
    assert d["xw"].size == 0 and d["xh"].size == 0 and d["h_stream"].size == 0 and (d["w_stream"] == w).all()
    open(fn, "wb").write(b"not a container at all, really not" * 3)
    with pytest.raises(eng.EngineError, match="not an ISS container"):
        eng.iss_container_read(fn)


def test_no_instruction_touches_an_lds_read_in_flight():
    """tools/check_inflight_regs.py on the ISA of every .hip file: (A) the latency-oriented encoder issues its state pass in
    inline assembly with the waits placed by hand -- a copy or spill the compiler put between such a read and its wait
    would be a race no parity test is sure to catch; (B) the decode kernels issue their wide output stores by hand, and a
    VALU write of the data registers within two wait states once produced wrong bytes with clean status words; (C) a VALU
    result written with a destination select must not be read by the very next VALU instruction.  Negative controls: the
    checker does flag a hand-made violation of each kind."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_regs as chk
    bad, counts = chk.check("_Zfoo:\n\tds_read_b64 v[4:5], v1 offset:768\n\tv_mov_b32_e32 v9, v5\n\ts_waitcnt lgkmcnt(0)\n")
    assert len(bad) == 1 and bad[0].startswith("A ") and counts == {"_Zfoo": 1}
    bad, _ = chk.check("_Zfoo:\n\tds_read_b64 v[4:5], v1 offset:768\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v9, v5\n")
    assert not bad
    # B: the data of a 128-bit store rewritten one wait state later / two wait states later / an LDS store (no such hazard)
    st = "_Zfoo:\n\tglobal_store_dwordx4 v1, v[4:7], s[2:3] nt\n"
    bad, _ = chk.check(st + "\tv_and_b32_e32 v6, 1, v9\n")
    assert len(bad) == 1 and bad[0].startswith("B ")
    bad, _ = chk.check(st + "\ts_nop 0\n\tv_and_b32_e32 v6, 1, v9\n")
    assert len(bad) == 1 and bad[0].startswith("B ")
    assert not chk.check(st + "\ts_nop 1\n\tv_and_b32_e32 v6, 1, v9\n")[0]
    assert not chk.check(st + "\tv_and_b32_e32 v8, 1, v6\n\tv_mov_b32_e32 v10, v6\n\tv_mov_b32_e32 v6, v1\n")[0]  # reads are fine
    assert not chk.check("_Zfoo:\n\tglobal_store_dwordx2 v1, v[4:5], s[2:3]\n\tv_mov_b32_e32 v4, v1\n")[0]      # 64 bits: no hazard
    assert not chk.check("_Zfoo:\n\tds_write_b128 v1, v[4:7]\n\tv_mov_b32_e32 v4, v1\n")[0]
    assert chk.check.sites["_Zfoo"][1] == 0
    # C: the byte written by an SDWA move read by the next VALU instruction / one instruction later
    pw = "_Zfoo:\n\tv_lshrrev_b32_sdwa v3, v8, v9 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
    bad, _ = chk.check(pw + "\tv_cndmask_b32_e64 v4, v5, v3, s[0:1]\n")
    assert len(bad) == 1 and bad[0].startswith("C ")
    assert not chk.check(pw + "\ts_nop 0\n\tv_cndmask_b32_e64 v4, v5, v3, s[0:1]\n")[0]
    assert not chk.check(pw + "\tds_write_b8 v1, v3\n")[0]  # (an LDS instruction is no VALU reader)
    pres = "_Zfoo:\n\tv_mov_b32_sdwa v3, v9 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0\n\tv_mov_b32_sdwa v3, v8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n"
    bad, _ = chk.check(pres)  # UNUSED_PRESERVE reads its own destination
    assert len(bad) == 1 and bad[0].startswith("C ")
    assert not chk.check("_Zfoo:\n\tv_add_u32_sdwa v3, v8, v9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\tv_mov_b32_e32 v4, v3\n")[0]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    # every coder kernel was among the ones checked: the latency-oriented encoder (A), the decode kernels' hand-issued stores
    # (B: 8 per store pass), the wave-pair kernels, the symbol and handle kernels
    for name in ("k_encodeILi1ELb1ELb0E", "k_decodeILi1ELb0ELb0E", "k_encode_pair", "k_decode_pair", "k_encode_symbols_trie", "k_stream_enc_bins"):
        assert name in out.stdout, name
    line = [l for l in out.stdout.splitlines() if "k_decodeILi1ELb0ELb0E" in l][0]
    assert int(line.split("LDS reads,")[1].split("wide stores")[0]) >= 8
