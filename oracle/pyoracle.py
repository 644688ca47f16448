"""ctypes bindings for the oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`Oracle()` wraps oracle/libcabac_oracle.so (the plain-C restatement, oracle/cabac_oracle.c);
`Reference()` wraps oracle/_ref/libcabac_ref.so (the real reference classes, built by
oracle/Makefile from /root/reference -- present only where that build ran).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libcabac_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libcabac_ref.so")
# the same sources with the reference's two compile-time switches on (state traces, fixed-probability bins)
REF_FULL_SO = os.path.join(HERE, "_ref", "libcabac_ref_full.so")
PROB_BASE = 155  # ctx byte of a fixed-probability bin = PROB_BASE + P(1) in percent (batches of <= 128 contexts)

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


_METHOD = {"TU": 0, "EG": 1, "TR": 2, "FL32": 3}  # cabacBinarizer.m:10-27 (TR / FL32: encode side only)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def build(force=False):
    """Compile the C restatement (and, when /root/reference exists, the reference shim) if they are missing.
    Never rebuilds an existing library implicitly: several bench ranks may import this at once."""
    if force or not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", HERE, "libcabac_oracle.so"], stdout=subprocess.DEVNULL)
    if (force or not os.path.exists(REF_SO) or not os.path.exists(REF_FULL_SO)) and os.path.isdir("/root/reference/CABAC"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def have_reference():
    return os.path.exists(REF_SO)


def have_reference_full():
    return os.path.exists(REF_FULL_SO)


def scratch_path(tag="s"):
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    return os.path.join(d, "isscabac_%s_%d.bin" % (tag, os.getpid()))


class Oracle:
    def __init__(self):
        build()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.orc_encode_stream.restype = C.c_size_t
        L.orc_encode_stream.argtypes = [u8p, u8p, C.c_uint32, u8p, C.c_int, u8p, C.c_size_t, u64p, u8p]
        L.orc_decode_stream.restype = C.c_uint32
        L.orc_decode_stream.argtypes = [u8p, C.c_size_t, u8p, C.c_uint32, u8p, C.c_int, u8p, u8p]
        L.orc_encode_batch.restype = C.c_uint64
        L.orc_encode_batch.argtypes = [u8p, u8p, u32p, C.c_uint64, C.c_uint32, u8p, C.c_int, u8p, C.c_uint64, u64p]
        L.orc_decode_batch.restype = None
        L.orc_decode_batch.argtypes = [u8p, u64p, u8p, u32p, C.c_uint64, C.c_uint32, u8p, C.c_int, u8p, u32p]
        L.orc_p0_to_packed.restype = C.c_uint8
        L.orc_p0_to_packed.argtypes = [C.c_double]
        L.orc_init_by_state.argtypes = [f64p, C.c_int, u8p]
        L.orc_init_by_prob.argtypes = [f64p, C.c_int, u8p]
        L.orc_ctx_pack.restype = C.c_uint8
        L.orc_ctx_pack.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_ctx_next_mps.restype = C.c_uint8
        L.orc_ctx_next_mps.argtypes = [C.c_uint8]
        L.orc_ctx_next_lps.restype = C.c_uint8
        L.orc_ctx_next_lps.argtypes = [C.c_uint8]
        L.orc_lps_table.restype = C.c_uint8
        L.orc_lps_table.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_renorm_table.restype = C.c_uint8
        L.orc_renorm_table.argtypes = [C.c_uint32]
        L.orc_binarize_tu.restype = C.c_int
        L.orc_binarize_tu.argtypes = [C.c_uint32, C.c_uint32, u8p]
        L.orc_binarize_egk.restype = C.c_int
        L.orc_binarize_egk.argtypes = [C.c_uint32, C.c_uint32, u8p]
        L.orc_debinarize_tu.restype = C.c_uint32
        L.orc_debinarize_tu.argtypes = [u8p, C.c_int, C.c_uint32]
        L.orc_debinarize_egk.restype = C.c_uint32
        L.orc_debinarize_egk.argtypes = [u8p, C.c_int, C.c_uint32]
        L.orc_iss_ctx.restype = C.c_int
        L.orc_iss_ctx.argtypes = [C.c_int, u8p, u8p, C.c_int, C.c_uint, C.c_int]
        L.orc_iss_encode_column.restype = C.c_size_t
        L.orc_iss_encode_column.argtypes = [u32p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint, C.c_int,
                                            u8p, C.c_int, u8p, C.c_size_t, u32p]
        L.orc_iss_decode_column.restype = C.c_uint32
        L.orc_iss_decode_column.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint,
                                            C.c_int, u8p, C.c_int, u32p]
        L.orc_iss_encode_matrix.restype = C.c_size_t
        L.orc_iss_encode_matrix.argtypes = [u32p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint, C.c_int,
                                            u8p, C.c_int, u8p, C.c_size_t, u32p]
        L.orc_iss_decode_matrix.restype = C.c_uint32
        L.orc_iss_decode_matrix.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint,
                                            C.c_int, u8p, C.c_int, u32p]
        L.orc_iss_init_context_model.restype = None
        L.orc_iss_init_context_model.argtypes = [u32p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint,
                                                 C.c_int, f64p]
        L.orc_matlab_uint8_of_p.restype = C.c_uint8
        L.orc_matlab_uint8_of_p.argtypes = [C.c_double]
        L.orc_demo_encode.restype = C.c_size_t
        L.orc_demo_encode.argtypes = [u32p, C.c_int, C.c_uint32, u8p, u8p, C.c_size_t, u32p]
        L.orc_demo_decode.restype = C.c_uint32
        L.orc_demo_decode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u8p, u32p]
        L.orc_synth_fill.restype = None
        L.orc_synth_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, u8p, u8p]
        # incremental encoder/decoder for op scripts
        L.orc_enc_start.argtypes = [C.c_void_p, u8p, C.c_size_t]
        L.orc_enc_bin.argtypes = [C.c_void_p, C.c_uint32, u8p]
        L.orc_enc_bin_ep.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_enc_bins_ep.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.orc_enc_finish.argtypes = [C.c_void_p]
        L.orc_enc_bytes.restype = C.c_size_t
        L.orc_enc_bytes.argtypes = [C.c_void_p]
        L.orc_enc_bits.restype = C.c_uint64
        L.orc_enc_bits.argtypes = [C.c_void_p]
        L.orc_dec_start.argtypes = [C.c_void_p, u8p, C.c_size_t]
        L.orc_dec_bin.restype = C.c_uint32
        L.orc_dec_bin.argtypes = [C.c_void_p, u8p]
        L.orc_dec_bin_ep.restype = C.c_uint32
        L.orc_dec_bin_ep.argtypes = [C.c_void_p]
        L.orc_dec_bins_ep.restype = C.c_uint32
        L.orc_dec_bins_ep.argtypes = [C.c_void_p, C.c_int]
        L.orc_dec_finish.restype = C.c_uint32
        L.orc_dec_finish.argtypes = [C.c_void_p]
        L.orc_enc_bin_prob.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_dec_bin_prob.restype = C.c_uint32
        L.orc_dec_bin_prob.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_prob_table.restype = C.c_uint8
        L.orc_prob_table.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_binarize_trk.restype = C.c_int
        L.orc_binarize_trk.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.orc_binarize_fl.restype = C.c_int
        L.orc_binarize_fl.argtypes = [C.c_uint32, C.c_int, u8p]
        L.orc_trace_state.restype = C.c_uint8
        L.orc_trace_state.argtypes = [C.c_uint8]
        L.orc_trace_stats.restype = C.c_uint64
        L.orc_trace_stats.argtypes = [u8p, u8p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_uint64, u32p]

    def counters(self, reset=False):
        class Cn(C.Structure):
            _fields_ = [(n, C.c_uint64) for n in ("writeouts", "ff_leads", "carry_resolves", "finish_carries",
                                                  "max_num_buffered")]
        self.lib.orc_counters_get.restype = C.POINTER(Cn)
        c = self.lib.orc_counters_get().contents
        d = {n: int(getattr(c, n)) for n, _ in Cn._fields_}
        if reset:
            self.lib.orc_counters_reset()
        return d

    # ---- whole streams -------------------------------------------------------------------
    @staticmethod
    def stream_cap(nbins):
        return (6 * int(nbins) + 9 + 7) // 8 + 8

    def encode_stream(self, bins, ctx, ctx_init, want_final=False):
        bins = np.ascontiguousarray(bins, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        out = np.zeros(self.stream_cap(bins.size), np.uint8)
        bits = C.c_uint64(0)
        fin = np.zeros(max(1, ctx_init.size), np.uint8)
        n = self.lib.orc_encode_stream(_p(bins, u8p), _p(ctx, u8p), bins.size, _p(ctx_init, u8p), ctx_init.size,
                                       _p(out, u8p), out.size, C.byref(bits), _p(fin, u8p))
        assert n > 0
        if want_final:
            return out[:n].copy(), int(bits.value), fin[:ctx_init.size]
        return out[:n].copy()

    def decode_stream(self, payload, ctx, ctx_init):
        payload = np.ascontiguousarray(payload, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        bins = np.zeros(ctx.size, np.uint8)
        st = self.lib.orc_decode_stream(_p(payload, u8p), payload.size, _p(ctx, u8p), ctx.size, _p(ctx_init, u8p),
                                        ctx_init.size, _p(bins, u8p), None)
        return bins, int(st)

    def encode_batch(self, bins, ctx, ctx_init, nbins=None):
        bins = np.ascontiguousarray(bins, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        S, B = bins.shape
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        nb = None if nbins is None else np.ascontiguousarray(nbins, np.uint32)
        cap = S * self.stream_cap(B)
        payload = np.zeros(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        tot = self.lib.orc_encode_batch(_p(bins, u8p), _p(ctx, u8p), _p(nb, u32p), S, B, _p(ctx_init, u8p),
                                        ctx_init.size, _p(payload, u8p), cap, _p(offsets, u64p))
        assert tot != 2 ** 64 - 1
        return payload[:tot].copy(), offsets

    def decode_batch(self, payload, offsets, ctx, ctx_init, nbins=None):
        payload = np.ascontiguousarray(payload, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        S, B = ctx.shape
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        nb = None if nbins is None else np.ascontiguousarray(nbins, np.uint32)
        bins = np.zeros((S, B), np.uint8)
        status = np.zeros(S, np.uint32)
        self.lib.orc_decode_batch(_p(payload, u8p), _p(offsets, u64p), _p(ctx, u8p), _p(nb, u32p), S, B,
                                  _p(ctx_init, u8p), ctx_init.size, _p(bins, u8p), _p(status, u32p))
        return bins, status

    # ---- state traces --------------------------------------------------------------------
    def trace_stats(self, bins, ctx, ctx_init, which_ctx, decoder=False):
        """(trace [N, 5] uint8, transitions [128, 128] uint32) of context `which_ctx`: getEncoderStats / getDecoderStats."""
        bins = np.ascontiguousarray(bins, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        trace = np.zeros((max(1, bins.size), 5), np.uint8)
        trans = np.zeros((128, 128), np.uint32)
        n = self.lib.orc_trace_stats(_p(bins, u8p), _p(ctx, u8p), bins.size, _p(ctx_init, u8p), ctx_init.size,
                                     int(which_ctx), int(bool(decoder)), _p(trace, u8p), trace.shape[0], _p(trans, u32p))
        return trace[:n].copy(), trans

    # ---- op scripts (kind 0 ctx bin, 1 EP bin, 2 multi-EP, 3 fixed-probability bin: arg = P(1) in percent) ---------
    def encode_ops(self, kind, val, arg, ctx_init):
        cm = np.array(ctx_init, np.uint8).copy()
        cap = self.stream_cap(int(np.sum(np.where(np.asarray(kind) == 2, arg, 1)))) + 64
        out = np.zeros(cap, np.uint8)
        enc = C.create_string_buffer(256)
        L = self.lib
        L.orc_enc_start(enc, _p(out, u8p), cap)
        for k, v, a in zip(kind, val, arg):
            if k == 0:
                L.orc_enc_bin(enc, int(v), C.cast(cm.ctypes.data + int(a), u8p))
            elif k == 1:
                L.orc_enc_bin_ep(enc, int(v))
            elif k == 3:
                L.orc_enc_bin_prob(enc, int(v), int(a))
            else:
                L.orc_enc_bins_ep(enc, int(v), int(a))
        L.orc_enc_finish(enc)
        length = L.orc_enc_bytes(enc)
        return out[:length].copy(), cm

    def decode_ops(self, payload, kind, arg, ctx_init):
        cm = np.array(ctx_init, np.uint8).copy()
        payload = np.ascontiguousarray(payload, np.uint8)
        dec = C.create_string_buffer(256)
        L = self.lib
        L.orc_dec_start(dec, _p(payload, u8p), payload.size)
        res = []
        for k, a in zip(kind, arg):
            if k == 0:
                res.append(L.orc_dec_bin(dec, C.cast(cm.ctypes.data + int(a), u8p)))
            elif k == 1:
                res.append(L.orc_dec_bin_ep(dec))
            elif k == 3:
                res.append(L.orc_dec_bin_prob(dec, int(a)))
            else:
                res.append(L.orc_dec_bins_ep(dec, int(a)))
        st = L.orc_dec_finish(dec)
        return np.array(res, np.uint32), int(st), cm

    # ---- init maps -------------------------------------------------------------------------
    def p0_to_packed(self, p0):
        return int(self.lib.orc_p0_to_packed(float(p0)))

    def init_by_prob(self, p0):
        p0 = np.ascontiguousarray(p0, np.float64).ravel()
        out = np.zeros(p0.size, np.uint8)
        self.lib.orc_init_by_prob(_p(p0, f64p), p0.size, _p(out, u8p))
        return out

    def init_by_state(self, triples):
        t = np.ascontiguousarray(triples, np.float64).ravel()
        n = t.size // 3
        out = np.zeros(n, np.uint8)
        self.lib.orc_init_by_state(_p(t, f64p), n, _p(out, u8p))
        return out

    # ---- binarisation / selectors ----------------------------------------------------------
    def binarize(self, v, method, nq=0, k=0):
        buf = np.zeros(128, np.uint8)
        buf = np.zeros(320, np.uint8)
        if method == "TU":
            n = self.lib.orc_binarize_tu(int(v), int(nq) - 1, _p(buf, u8p))
        elif method == "TR":
            n = self.lib.orc_binarize_trk(int(v), int(k), int(nq) - 1, _p(buf, u8p))
        elif method == "FL32":
            n = self.lib.orc_binarize_fl(int(v), 32, _p(buf, u8p))
        else:
            n = self.lib.orc_binarize_egk(int(v), int(k), _p(buf, u8p))
        return buf[:n].copy()

    def debinarize(self, bins, method, nq=0, k=0):
        b = np.ascontiguousarray(bins, np.uint8)
        if method == "TU":
            return int(self.lib.orc_debinarize_tu(_p(b, u8p), b.size, int(nq) - 1))
        return int(self.lib.orc_debinarize_egk(_p(b, u8p), b.size, int(k)))

    def iss_ctx(self, n, g_prev, g_up1, types, nlbp):
        gp = np.ascontiguousarray(list(g_prev) + [0], np.uint8)
        gu = np.ascontiguousarray(list(g_up1) + [0], np.uint8)
        return int(self.lib.orc_iss_ctx(int(n), _p(gp, u8p), _p(gu, u8p), len(g_up1), int(types), int(nlbp)))

    def iss_encode_column(self, sym, nq, method, k, types, nlbp, ctx_init):
        sym = np.ascontiguousarray(sym, np.uint32)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        cap = self.stream_cap(sym.size * 320)
        out = np.zeros(cap, np.uint8)
        nb = C.c_uint32(0)
        n = self.lib.orc_iss_encode_column(_p(sym, u32p), sym.size, nq, _METHOD[method], k, types, nlbp,
                                           _p(ctx_init, u8p), ctx_init.size, _p(out, u8p), cap, C.byref(nb))
        return out[:n].copy(), int(nb.value)

    def iss_decode_column(self, payload, rows, nq, method, k, types, nlbp, ctx_init):
        payload = np.ascontiguousarray(payload, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        sym = np.zeros(rows, np.uint32)
        st = self.lib.orc_iss_decode_column(_p(payload, u8p), payload.size, rows, nq, _METHOD[method], k,
                                            types, nlbp, _p(ctx_init, u8p), ctx_init.size, _p(sym, u32p))
        return sym, int(st)

    def iss_encode_matrix(self, G, nq, method, k, types, nlbp, ctx_init):
        """G: [rows, cols]; coded column by column as ONE stream (cabacEncode.m:45-70)."""
        G = np.asarray(G)
        rows, cols = G.shape
        sym = np.ascontiguousarray(G.T, np.uint32).ravel()
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        cap = self.stream_cap(sym.size * 320)
        out = np.zeros(cap, np.uint8)
        nb = C.c_uint32(0)
        n = self.lib.orc_iss_encode_matrix(_p(sym, u32p), rows, cols, nq, _METHOD[method], k, types, nlbp,
                                           _p(ctx_init, u8p), ctx_init.size, _p(out, u8p), cap, C.byref(nb))
        return out[:n].copy(), int(nb.value)

    def iss_encode_matrix_stats(self, G, nq, method, k, types, nlbp, ctx_init):
        """-> (payload, ctxHist[7*nlbp+2], ctxCost[...], H[rows, cols]) of cabacEncode.m:39-42,60-67"""
        G = np.asarray(G)
        rows, cols = G.shape
        sym = np.ascontiguousarray(G.T, np.uint32).ravel()
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        cap = self.stream_cap(sym.size * 320)
        out = np.zeros(cap, np.uint8)
        n = 7 * nlbp + 2
        hist, cost, H = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(rows * cols, np.uint32)
        f = self.lib.orc_iss_encode_matrix_stats
        f.restype = C.c_size_t
        f.argtypes = [u32p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint, C.c_int, u8p, C.c_int, u8p,
                      C.c_size_t, u32p, u32p, u32p]
        m = f(_p(sym, u32p), rows, cols, nq, _METHOD[method], k, types, nlbp, _p(ctx_init, u8p), ctx_init.size,
              _p(out, u8p), cap, _p(hist, u32p), _p(cost, u32p), _p(H, u32p))
        return out[:m].copy(), hist, cost, H.reshape(cols, rows).T.copy()

    def iss_decode_matrix(self, payload, rows, cols, nq, method, k, types, nlbp, ctx_init):
        payload = np.ascontiguousarray(payload, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        sym = np.zeros(rows * cols, np.uint32)
        st = self.lib.orc_iss_decode_matrix(_p(payload, u8p), payload.size, rows, cols, nq, _METHOD[method],
                                            k, types, nlbp, _p(ctx_init, u8p), ctx_init.size, _p(sym, u32p))
        return sym.reshape(cols, rows).T.copy(), int(st)

    def iss_init_context_model(self, G, nq, method, k, types, nlbp):
        """cabacInitContextModel.m on the rows x cols matrix G -> (p0 doubles, uint8(p0*255) bytes)."""
        G = np.asarray(G)
        rows, cols = G.shape
        sym = np.ascontiguousarray(G.T, np.uint32).ravel()
        out = np.zeros(7 * nlbp + 2, np.float64)
        self.lib.orc_iss_init_context_model(_p(sym, u32p), rows, cols, nq, _METHOD[method], k, types, nlbp,
                                            _p(out, f64p))
        return out, np.array([self.lib.orc_matlab_uint8_of_p(float(v)) for v in out], np.uint8)

    def demo_encode(self, sym, nq, ctx_init):
        sym = np.ascontiguousarray(sym, np.uint32)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        cap = self.stream_cap(sym.size * nq)
        out = np.zeros(cap, np.uint8)
        nb = C.c_uint32(0)
        n = self.lib.orc_demo_encode(_p(sym, u32p), sym.size, nq, _p(ctx_init, u8p), _p(out, u8p), cap, C.byref(nb))
        return out[:n].copy(), int(nb.value)

    def demo_decode(self, payload, nsym, nq, ctx_init):
        payload = np.ascontiguousarray(payload, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        sym = np.zeros(nsym, np.uint32)
        st = self.lib.orc_demo_decode(_p(payload, u8p), payload.size, nsym, nq, _p(ctx_init, u8p), _p(sym, u32p))
        return sym, int(st)

    def roundtrip_threads(self, seed, s0, B, Cn, ep_half, nthreads, seconds):
        """the restatement, in-memory sink, `nthreads` native threads for about `seconds` ->
        (streams completed, wall seconds, every round trip exact)"""
        f = self.lib.orc_roundtrip_threads
        f.restype = C.c_int
        f.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_double, u64p, f64p]
        wall, done = C.c_double(0), C.c_uint64(0)
        ok = f(int(seed), int(s0), int(B), int(Cn), int(bool(ep_half)), int(nthreads), float(seconds), C.byref(done),
               C.byref(wall))
        return int(done.value), wall.value, bool(ok)

    def synth(self, seed, s0, nS, B, Cn, ep_half=False):
        bins = np.zeros((nS, B), np.uint8)
        ctx = np.zeros((nS, B), np.uint8)
        self.lib.orc_synth_fill(seed, s0, nS, B, Cn, int(ep_half), _p(bins, u8p), _p(ctx, u8p))
        return bins, ctx


class Reference:
    """The real reference classes (oracle/_ref/libcabac_ref.so)."""

    def __init__(self, tag="ref", full=False):
        """full=True: the build with RWTH_TRACE_CABAC_STATES and RWTH_CABAC_FIXED_PROBABILITY on (op kind 3, get_trace)."""
        so = REF_FULL_SO if full else REF_SO
        if not os.path.exists(so):
            build()
        if not os.path.exists(so):
            raise RuntimeError("%s not built (needs /root/reference)" % so)
        L = self.lib = C.CDLL(so)
        self.full = full
        if full:
            assert L.ref_has_fixed_prob() == 1 and L.ref_has_trace() == 1
            L.ref_get_trace.restype = C.c_long
            L.ref_get_trace.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, u32p]
        L.ref_encode_ops.restype = C.c_long
        L.ref_encode_ops.argtypes = [u8p, u32p, u32p, C.c_uint32, u8p, C.c_int, C.c_char_p, u8p, C.c_size_t,
                                     u32p, u32p, u8p]
        L.ref_decode_ops.restype = C.c_long
        L.ref_decode_ops.argtypes = [u8p, C.c_size_t, u8p, u32p, C.c_uint32, u8p, C.c_int, C.c_char_p, u32p, u32p,
                                     u8p]
        L.ref_roundtrip_batch.restype = C.c_long
        L.ref_roundtrip_batch.argtypes = [u8p, u8p, C.c_uint64, C.c_uint32, u8p, C.c_int, C.c_char_p, C.c_int,
                                          f64p, f64p]
        L.ref_next_state.restype = C.c_uint8
        L.ref_next_state.argtypes = [C.c_uint8, C.c_int]
        self.scratch = scratch_path(tag).encode()

    def __del__(self):
        try:
            os.unlink(self.scratch.decode())
        except Exception:
            pass

    def encode_ops(self, kind, val, arg, ctx_init, want_trace=False):
        kind = np.ascontiguousarray(kind, np.uint8)
        val = np.ascontiguousarray(val, np.uint32)
        arg = np.ascontiguousarray(arg, np.uint32)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        nb = int(np.sum(np.where(kind == 2, arg, 1)))
        cap = (6 * nb + 9 + 7) // 8 + 64
        out = np.zeros(cap, np.uint8)
        trace = np.zeros((kind.size, 5), np.uint32)
        bits = np.zeros(kind.size, np.uint32)
        fin = np.zeros(max(1, ctx_init.size), np.uint8)
        n = self.lib.ref_encode_ops(_p(kind, u8p), _p(val, u32p), _p(arg, u32p), kind.size, _p(ctx_init, u8p),
                                    ctx_init.size, self.scratch, _p(out, u8p), cap, _p(trace, u32p), _p(bits, u32p),
                                    _p(fin, u8p))
        assert n >= 0, n
        if want_trace:
            return out[:n].copy(), fin[:ctx_init.size], trace, bits
        return out[:n].copy(), fin[:ctx_init.size]

    def decode_ops(self, payload, kind, arg, ctx_init):
        payload = np.ascontiguousarray(payload, np.uint8)
        kind = np.ascontiguousarray(kind, np.uint8)
        arg = np.ascontiguousarray(arg, np.uint32)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        res = np.zeros(kind.size, np.uint32)
        fin = np.zeros(max(1, ctx_init.size), np.uint8)
        st = self.lib.ref_decode_ops(_p(payload, u8p), payload.size, _p(kind, u8p), _p(arg, u32p), kind.size,
                                     _p(ctx_init, u8p), ctx_init.size, self.scratch, _p(res, u32p), None, _p(fin, u8p))
        assert st >= 0, st
        return res, int(st), fin[:ctx_init.size]

    def get_trace(self, decoder, ctx, cap=1 << 20):
        """getEncoderStats / getDecoderStats of context `ctx` after the most recent encode_ops / decode_ops (full build)."""
        trace = np.zeros((cap, 5), np.uint8)
        trans = np.zeros((128, 128), np.uint32)
        n = self.lib.ref_get_trace(int(bool(decoder)), int(ctx), _p(trace, u8p), cap, _p(trans, u32p))
        assert 0 <= n <= cap, n
        return trace[:n].copy(), trans

    def encode_stream(self, bins, ctx, ctx_init):
        """ctx byte 0xFF = EP bin (the shim maps it to encodeBinEP)."""
        bins = np.asarray(bins, np.uint8)
        ctx = np.asarray(ctx, np.uint8)
        kind = (ctx == 0xFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx).astype(np.uint32)
        return self.encode_ops(kind, bins.astype(np.uint32), arg, ctx_init)

    def decode_stream(self, payload, ctx, ctx_init):
        ctx = np.asarray(ctx, np.uint8)
        kind = (ctx == 0xFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx).astype(np.uint32)
        res, st, fin = self.decode_ops(payload, kind, arg, ctx_init)
        return res.astype(np.uint8), st, fin

    def roundtrip_batch(self, bins, ctx, ctx_init, decode=True):
        bins = np.ascontiguousarray(bins, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint8)
        ctx_init = np.ascontiguousarray(ctx_init, np.uint8)
        S, B = bins.shape
        te, td = C.c_double(0), C.c_double(0)
        tot = self.lib.ref_roundtrip_batch(_p(bins, u8p), _p(ctx, u8p), S, B, _p(ctx_init, u8p), ctx_init.size,
                                           self.scratch, 3 if decode else 1, C.byref(te), C.byref(td))
        assert tot >= 0, tot
        return int(tot), te.value, td.value

    def next_state(self, packed, lps):
        return int(self.lib.ref_next_state(int(packed), int(lps)))
