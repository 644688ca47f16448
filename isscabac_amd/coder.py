"""Python mirror of the reference's ISS coder front end, on the batch engine:

    ISS/+coder/cabacEncode.m   -> cabacEncode(G, Nq, param)            returns (nbits, ctxInit0) and writes param["fn"]
    ISS/+coder/cabacDecode.m   -> cabacDecode(Nq, param, ctxInit, siz) returns G

Same parameter names as the MATLAB `param` struct (ISS/ISS.m:49-53): binMethod ('DEC2TU', 'DEC2EG0', 'DEC2EG1',
'DEC2EG2'), cmTypes (subset of 'cond0', 'cond1', 'conds0', 'conds1', 'condbinlft'), Nlbp, equalProb, fn.  A whole matrix
is ONE bitstream, coded column after column (cabacEncode.m:45-70) -- one call instead of one MEX call per bin.  That one
stream is serial work (16 000 bins on one lane: 1.3 / 2.5 ms on the device, 0.2 ms on one host core), so by default it is coded by
the engine's host twin (isscabac_cpu_encode_symbols / _decode_symbols: the kernels' own per-lane code compiled for the host);
`serial="gpu"` -- the default when an engine is handed in -- keeps it on the device.  The empirical context initialisation
(cabacInitContextModel.m) is a histogram over the whole matrix and runs on the GPU either way; the parallel form of this
workload is columns as independent streams (Engine.encode_symbols).
"""
import numpy as np

from .engine import Engine, SymbolParams, cpu_decode_symbols, cpu_encode_symbols, init_by_prob

_TYPES = {"cond0": 1, "cond1": 2, "conds0": 4, "conds1": 8, "condbinlft": 16}
_engine = None


def _eng(engine):
    global _engine
    if engine is not None:
        return engine
    if _engine is None:
        _engine = Engine(0)
    return _engine


def _symbol_params(Nq, param, col_rows=0):
    method = param.get("binMethod", "DEC2EG0")
    if Nq <= 2:  # cabacEncode.m:15-20: no binarisation, the symbols are the bins (TU with one bin is exactly that)
        kind, k, nq = "TU", 0, 2
    elif method == "DEC2TU":
        kind, k, nq = "TU", 0, int(Nq)
    elif method in ("DEC2EG0", "DEC2EG1", "DEC2EG2"):
        kind, k, nq = "EG", int(method[-1]), int(Nq)
    else:
        raise ValueError("binMethod %r: the reference's decoder knows DEC2TU and DEC2EG0..2 only "
                         "(cabacDecodeSymbolFinished.m:12-30)" % (method,))
    types = 0
    for t in param.get("cmTypes", ("cond0", "cond1", "conds0", "conds1")):
        types |= _TYPES[t]
    return SymbolParams.iss(nq, kind, k, types, int(param.get("Nlbp", 3)), col_rows=col_rows)


def cabacEncode(G, Nq=2, param=None, engine=None, stats=False, serial=None):
    """-> (nbits, ctxInit0 uint8) like +coder/cabacEncode.m; with stats=True also (ctxHist, ctxCost, H) of :39-42,60-67.
    serial: "host" (the host twin codes the one stream; default) or "gpu" (default when `engine` is given)."""
    param = dict(param or {})
    on_host = (serial or ("gpu" if engine is not None else "host")) == "host"
    eng = None if (on_host and param.get("equalProb")) else _eng(engine)  # (no GPU touched at all for an equal-probability start)
    G = np.ascontiguousarray(G, np.uint8)
    rows, cols = G.shape
    cols_first = np.ascontiguousarray(G.T)  # one column per row: the engine's stream-major layout
    sp = _symbol_params(Nq, param)
    n = 7 * sp.nlbp + 2
    if param.get("equalProb"):
        ctx_init0 = np.full(n, 128, np.uint8)  # uint8(0.5 * 255) in MATLAB
    else:
        _, ctx_init0 = eng.iss_init_context_model(cols_first, sp)  # cabacEncode.m:23-31
    states = init_by_prob(ctx_init0.astype(np.float64) / 255.0)
    mp = _symbol_params(Nq, param, col_rows=rows)
    one = cols_first.reshape(1, -1)
    if stats:
        payload, _, _, hist, cost, hb = (cpu_encode_symbols(one, mp, states, stats=True) if on_host
                                         else eng.encode_symbols_stats(one, mp, states))
        extra = (hist[0, :n].copy(), cost[0, :n].copy(), hb[0, :rows * cols].reshape(cols, rows).T.copy())
    else:
        payload, _, _ = cpu_encode_symbols(one, mp, states) if on_host else eng.encode_symbols(one, mp, states)
        extra = ()
    if param.get("fn"):
        with open(param["fn"], "wb") as f:
            f.write(payload.tobytes())
    return (8 * payload.size, ctx_init0) + extra


def cabacDecode(Nq, param, ctxInit, siz, engine=None, payload=None, serial=None):
    """-> G (uint8 [rows, cols]) like +coder/cabacDecode.m; raises if the stream's terminate bin or stop bit is wrong
    (the reference asserts there).  serial: as in cabacEncode."""
    param = dict(param or {})
    on_host = (serial or ("gpu" if engine is not None else "host")) == "host"
    rows, cols = int(siz[0]), int(siz[1])
    if payload is None:
        with open(param["fn"], "rb") as f:
            payload = np.frombuffer(f.read(), np.uint8)
    payload = np.ascontiguousarray(payload, np.uint8)
    states = init_by_prob(np.asarray(ctxInit, np.float64) / 255.0)  # cabacDecode.m:13
    mp = _symbol_params(Nq, param, col_rows=rows)
    off = np.array([0, payload.size], np.uint64)
    sym, status = (cpu_decode_symbols(payload, off, rows * cols, mp, states) if on_host
                   else _eng(engine).decode_symbols(payload, off, rows * cols, mp, states))
    if status[0]:
        raise ValueError("CABAC stream did not terminate properly (status %d)" % int(status[0]))
    return sym[0, :rows * cols].reshape(cols, rows).T.copy()
