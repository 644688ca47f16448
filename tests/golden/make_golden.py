#!/usr/bin/env python3
"""Generate tests/golden/cabac_golden.json from the REAL reference (oracle/_ref/libcabac_ref.so).

Run in the build container (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden.py

Every expected payload / decoded bin / final context state in the fixture comes from the reference
classes (CABAC_ArithmeticEncoder/Decoder, ContextModel, CABAC_BitstreamFile) driven by
oracle/ref_shim.cpp.  The C restatement is used here only to *select* interesting inputs (streams
that exercise 0xFF runs, carries and the carry branch of finish()), never to produce expectations.
The fixture is data only: inputs, expected outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabac_golden.json")


def hx(a):
    return np.ascontiguousarray(a, np.uint8).tobytes().hex()


def bits_hex(b):
    return np.packbits(np.asarray(b, np.uint8)).tobytes().hex()


def main():
    orc, ref = Oracle(), Reference()
    rng = np.random.default_rng(20261003)
    streams, ops, synth = [], [], []

    def add_stream(name, bins, ctx, ctx_init, note=""):
        bins = np.asarray(bins, np.uint8)
        ctx = np.asarray(ctx, np.uint8)
        ctx_init = np.asarray(ctx_init, np.uint8)
        kind = (ctx == 0xFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx).astype(np.uint32)
        payload, fin, _trace, bits = ref.encode_ops(kind, bins.astype(np.uint32), arg, ctx_init, want_trace=True)
        dec, status, dfin = ref.decode_stream(payload, ctx, ctx_init)
        assert status == 0 and (dec == bins).all() and (dfin == fin).all(), name
        orc.counters(reset=True)
        orc.encode_stream(bins, ctx, ctx_init)
        ev = orc.counters()
        streams.append(dict(name=name, note=note, n=int(bins.size), nctx=int(ctx_init.size), ctx_init=hx(ctx_init),
                            bins=bits_hex(bins), ctx=hx(ctx), payload=hx(payload),
                            bits_before_finish=int(bits[-1]) if bins.size else 0, ctx_final=hx(fin), events=ev))
        return ev

    # ---- known answers quoted in SURVEY.md section 8(c) ---------------------------------------
    kind = [0] * 6 + [1] * 5 + [2] + [0] * 6
    val = [0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 18, 1, 1, 0, 1, 1, 1]
    arg = [0] * 6 + [0] * 5 + [5] + [1] * 6
    init = np.array([(20 << 1) | 0, (0 << 1) | 1], np.uint8)  # SimpleCABAC.cpp:63-64 init(0,20); ctx1 default (1,0)
    payload, fin = ref.encode_ops(kind, val, arg, init)
    res, st, _ = ref.decode_ops(payload, kind, arg, init)
    assert st == 0 and list(res) == val
    ops.append(dict(name="simplecabac_main", note="CABAC/SimpleCABAC.cpp:48-116 script", kind=kind, val=val, arg=arg,
                    ctx_init=hx(init), payload=hx(payload), ctx_final=hx(fin)))
    add_stream("empty", [], [], [0], "start(); finish()")
    x, b, c = 12345, [], []
    for _ in range(200):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        b.append((x >> 16) & 1)
        c.append((x >> 20) % 3)
    add_stream("lcg200_3ctx", b, c, [0, 0, 0], "SURVEY 8(c) known answer 3")
    add_stream("mps_20000", np.zeros(20000, np.uint8), np.zeros(20000, np.uint8), [0], "all-MPS, state saturates at 62")

    # ---- single bins, every initial packed state ------------------------------------------------
    for p in (0, 1, 2, 61, 124, 125):
        for bin_ in (0, 1):
            add_stream("single_p%d_b%d" % (p, bin_), [bin_], [0], [p])

    # ---- all-LPS adversarial (bin always != mps), several contexts ------------------------------
    n = 1500
    cm = [orc.lib.orc_ctx_pack(0, 62)] * 4
    bins, ctx = [], []
    for i in range(n):
        cidx = i % 4
        bin_ = 1 - (cm[cidx] & 1)
        bins.append(bin_)
        ctx.append(cidx)
        cm[cidx] = orc.lib.orc_ctx_next_lps(cm[cidx])
    add_stream("all_lps_4ctx", bins, ctx, [124] * 4, "every bin is the LPS, starts at state 62")

    # ---- state-62 saturation then LPS bursts ---------------------------------------------------
    bins = np.zeros(3000, np.uint8)
    bins[500::97] = 1
    add_stream("sat62_bursts", bins, np.zeros(3000, np.uint8), [124], "mps=0,state=62 with sparse LPS")

    # ---- EP only ---------------------------------------------------------------------------------
    for n in (1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 40, 1000):
        add_stream("ep_only_%d" % n, rng.integers(0, 2, n), np.full(n, 0xFF), [0])
    add_stream("ep_all_ones_300", np.ones(300, np.uint8), np.full(300, 0xFF), [0], "0xFF runs from bypass ones")
    add_stream("ep_all_zeros_300", np.zeros(300, np.uint8), np.full(300, 0xFF), [0])

    # ---- multi-bin EP op scripts (encodeBinsEP / decodeBinsEP) ----------------------------------
    for t in range(6):
        k, v, a = [], [], []
        for _ in range(60):
            r = rng.integers(0, 3)
            if r == 0:
                k.append(0); v.append(int(rng.integers(0, 2))); a.append(int(rng.integers(0, 5)))
            elif r == 1:
                k.append(1); v.append(int(rng.integers(0, 2))); a.append(0)
            else:
                nb = int(rng.integers(1, 33))
                k.append(2); v.append(int(rng.integers(0, 2 ** nb))); a.append(nb)
        init5 = rng.integers(0, 126, 5).astype(np.uint8)
        payload, fin = ref.encode_ops(k, v, a, init5)
        res, st, dfin = ref.decode_ops(payload, k, a, init5)
        assert st == 0 and list(res) == v and (fin == dfin).all()
        ops.append(dict(name="mixed_ops_%d" % t, note="ctx / EP / multi-EP mix", kind=k, val=v, arg=a,
                        ctx_init=hx(init5), payload=hx(payload), ctx_final=hx(fin)))

    # ---- random streams: skewed and uniform, ragged lengths, random inits -----------------------
    for t, (n, C) in enumerate([(1, 1), (2, 2), (3, 3), (17, 3), (64, 8), (255, 16), (1000, 64), (1024, 3),
                                (2500, 64), (4096, 64), (4096, 200), (777, 255)]):
        init_c = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, n).astype(np.uint8)
        p1 = rng.random(C)
        bins = (rng.random(n) < p1[ctx]).astype(np.uint8)
        add_stream("rand_%d_n%d_c%d" % (t, n, C), bins, ctx, init_c)
    # with bypass bins mixed in
    for t, n in enumerate([50, 2048]):
        C = 64
        init_c = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, n).astype(np.uint8)
        ctx[rng.random(n) < 0.5] = 0xFF
        bins = rng.integers(0, 2, n).astype(np.uint8)
        add_stream("rand_ep_mix_%d_n%d" % (t, n), bins, ctx, init_c, "50% bypass")

    # ---- searched: long 0xFF runs, carries in writeOut, carry branch of finish() ------------------
    want = dict(ff_run=3, carry=3, finish_carry=3)
    seed = 0
    while any(v > 0 for v in want.values()) and seed < 200000:
        seed += 1
        r2 = np.random.default_rng(seed)
        n = int(r2.integers(30, 600))
        C = int(r2.choice([1, 3, 8]))
        init_c = r2.integers(0, 126, C).astype(np.uint8)
        ctx = r2.integers(0, C, n).astype(np.uint8)
        if r2.random() < 0.5:
            ctx[r2.random(n) < 0.6] = 0xFF
        bins = (r2.random(n) < r2.random()).astype(np.uint8)
        orc.counters(reset=True)
        orc.encode_stream(bins, ctx, init_c)
        ev = orc.counters()
        tag = None
        if ev["finish_carries"] and want["finish_carry"] > 0:
            tag = "finish_carry"
        elif ev["carry_resolves"] and ev["max_num_buffered"] >= 2 and want["carry"] > 0:
            tag = "carry"
        elif ev["max_num_buffered"] >= 4 and want["ff_run"] > 0:
            tag = "ff_run"
        if tag:
            want[tag] -= 1
            add_stream("searched_%s_seed%d" % (tag, seed), bins, ctx, init_c, "selected for %s" % tag)
    assert all(v == 0 for v in want.values()), want

    # ---- synthetic generator of SURVEY 8(d): pins the generator AND the coder ---------------------
    for seed_, s0, nS, B, C, ep in [(0xCABAC002, 0, 4, 1024, 3, 0), (0xCABAC002, 65532, 4, 1024, 3, 0),
                                    (0xCABAC003, 0, 2, 4096, 64, 0), (0xCABAC003, 1048574, 2, 4096, 64, 0),
                                    (0xCABAC005, 0, 2, 8192, 64, 1), (0xCABAC005, 8388606, 2, 8192, 64, 1)]:
        bins, ctx = orc.synth(seed_, s0, nS, B, C, bool(ep))
        init_c = np.zeros(C, np.uint8)  # (mps 0, state 0) == initByProb with p0 byte 128
        payloads = []
        for s in range(nS):
            payload, fin = ref.encode_stream(bins[s], ctx[s], init_c)
            dec, st, _ = ref.decode_stream(payload, ctx[s], init_c)
            assert st == 0 and (dec == bins[s]).all()
            payloads.append(hx(payload))
        synth.append(dict(seed=seed_, s0=s0, nS=nS, B=B, C=C, ep_half=ep,
                          first_bins=bits_hex(bins[0][:64]), first_ctx=hx(ctx[0][:64]),
                          sha256_bins=hashlib.sha256(bins.tobytes()).hexdigest(),
                          sha256_ctx=hashlib.sha256(ctx.tobytes()).hexdigest(), payloads=payloads))

    # ---- exhaustive state-transition table from the reference ContextModel ------------------------
    next_mps = [ref.next_state(p, 0) for p in range(128)]
    next_lps = [ref.next_state(p, 1) for p in range(128)]

    doc = dict(
        meta=dict(generator="tests/golden/make_golden.py", source="oracle/_ref/libcabac_ref.so built from "
                  "/root/reference/CABAC/{CABAC_ArithmeticEncoder,CABAC_ArithmeticDecoder,CABAC_BitstreamFile,"
                  "ContextModel}.cpp", bins_encoding="numpy.packbits hex (MSB first)", ctx_encoding="hex, 1 byte/bin, "
                  "0xff = bypass", ctx_init_encoding="hex, packed (state<<1)|mps"),
        streams=streams, ops=ops, synth=synth, next_state_mps=next_mps, next_state_lps=next_lps)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(streams), "streams,", len(ops), "op scripts,",
          len(synth), "synthetic groups")


if __name__ == "__main__":
    main()
