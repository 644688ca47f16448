"""ctypes binding of the C ABI in include/isscabac.h (lib/libisscabac_hip.so).

This is the Python form of the FFI stub INTEGRATION.md describes: plain pointers and sizes go straight
through.  Device buffers are passed as integer addresses (e.g. ``tensor.data_ptr()``); host numpy arrays
go through the ``encode`` / ``decode`` convenience methods, which call the host-buffer entry points.

There is no CPU coding path: if the library or a HIP device is missing, constructing an Engine raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

BYPASS = 0xFF
PROB_BASE = 155  # ctx byte ISSCABAC_PROB(p) = PROB_BASE + p: fixed-probability bin (batches of at most 128 contexts)
OP_BYPASS = 0xFFFF  # 16-bit ops of the single-stream handle (ISSCABAC_OP_*)


def op_prob(p):
    return 0x8000 | int(p)

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

_lib = None


class SymbolParams(C.Structure):
    """isscabac_symbol_params: binarisation (0 TU, 1 EGk), k, nq, selector (0 demo, 1 ISS), iss_types, nlbp."""
    _fields_ = [("binarization", C.c_uint32), ("k", C.c_uint32), ("nq", C.c_uint32), ("selector", C.c_uint32),
                ("iss_types", C.c_uint32), ("nlbp", C.c_uint32), ("col_rows", C.c_uint32)]

    @classmethod
    def demo(cls, nq):
        """cabacDemo.m: TU binarisation, 3 contexts from the previous symbol's first bin."""
        return cls(0, 0, nq, 0, 0, 0, 0)

    @classmethod
    def iss(cls, nq, method="EG", k=0, types=15, nlbp=3, col_rows=0):  # method: "TU", "EG", encode only: "TR", "FL32"
        """ISS defaults (ISS/ISS.m:49-53): EG0, {cond0, cond1, conds0, conds1}, Nlbp = 3 -> 23 contexts.
        col_rows > 0: whole-matrix stream (columns back to back), the reference's one-file-per-matrix mode."""
        return cls({"TU": 0, "EG": 1, "TR": 2, "FL32": 3}[method], k, nq, 1, types, nlbp, col_rows)


class SymbolStats(C.Structure):
    """isscabac_symbol_stats: the encoder-side statistics of +coder/cabacEncode.m (ctxHist, ctxCost, H)."""
    _fields_ = [("ctx_hist", C.POINTER(C.c_uint32)), ("ctx_cost", C.POINTER(C.c_uint32)), ("sym_bits", C.POINTER(C.c_uint16)),
                ("bin_trace", C.POINTER(C.c_uint32)), ("bin_trace_pitch", C.c_uint32)]


STAT_CTX = 64


class IssSideInfo(C.Structure):
    """isscabac_iss_side_info: what ISS.m transmits next to its two CABAC streams (ISS/ISS.m:197-201)."""
    _fields_ = [("cW", C.POINTER(C.c_double)), ("cH", C.POINTER(C.c_double)), ("Q", C.POINTER(C.c_double)),
                ("xw", C.POINTER(C.c_uint8)), ("xh", C.POINTER(C.c_uint8)), ("ncW", C.c_uint32), ("ncH", C.c_uint32),
                ("q_rows", C.c_uint32), ("q_cols", C.c_uint32), ("nxw", C.c_uint32), ("nxh", C.c_uint32)]


class IssContainerInfo(C.Structure):
    _fields_ = [("ncW", C.c_uint32), ("ncH", C.c_uint32), ("q_rows", C.c_uint32), ("q_cols", C.c_uint32),
                ("nxw", C.c_uint32), ("nxh", C.c_uint32), ("w_bytes", C.c_uint64), ("h_bytes", C.c_uint64)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("isscabac error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return _build.LIB


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path) and not os.environ.get("ISSCABAC_LIB"):
        try:  # a source checkout without the built library: compile it now (hipcc cross-compiles gfx950 anywhere)
            _build.build_lib()
        except Exception as exc:
            raise RuntimeError("%s is missing and could not be built (%s); there is no CPU fallback for the coding "
                               "path" % (path, exc))
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -m isscabac_amd.build` (needs hipcc); there is no CPU "
                           "fallback for the coding path" % path)
    try:  # when PyTorch-ROCm is around, load its bundled HIP runtime first: two copies of libamdhip64 in one process
        import torch  # noqa: F401   (ours via the rpath, torch's later) leave the later one without a device
    except Exception:
        pass
    L = C.CDLL(path)
    L.isscabac_abi_version.restype = C.c_int
    L.isscabac_create.argtypes = [C.c_int, C.POINTER(_vp)]
    L.isscabac_destroy.argtypes = [_vp]
    L.isscabac_destroy.restype = None
    L.isscabac_last_error.argtypes = [_vp]
    L.isscabac_last_error.restype = C.c_char_p
    L.isscabac_set_stream.argtypes = [_vp, _vp]
    L.isscabac_set_tuning.argtypes = [_vp, C.c_int]
    L.isscabac_last_instance.argtypes = [_vp, C.c_int]
    L.isscabac_last_instance.restype = C.c_char_p
    L.isscabac_init_by_prob.argtypes = [_f64p, C.c_int, _u8p]
    L.isscabac_init_by_state.argtypes = [_f64p, C.c_int, _u8p]
    L.isscabac_stream_bound.argtypes = [C.c_uint32]
    L.isscabac_stream_bound.restype = C.c_uint64
    L.isscabac_encode_batch_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                            _vp, C.c_uint64, _vp, _u64p]
    L.isscabac_decode_batch_dev.argtypes = [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                            _u8p, C.c_int, _vp, _vp]
    L.isscabac_encode_batch_dev_async.argtypes = [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                                  _vp, C.c_uint64, _vp]
    L.isscabac_decode_batch_dev_async.argtypes = L.isscabac_decode_batch_dev.argtypes
    L.isscabac_sync.argtypes = [_vp, _u64p]
    L.isscabac_kernel_ms_mean.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    if hasattr(L, "isscabac_kernel_ms_samples_dropped"):
        L.isscabac_kernel_ms_samples_dropped.argtypes = [_vp, C.c_int, _u64p]
    L.isscabac_kernel_ms_samples.argtypes = [_vp, C.c_int, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    L.isscabac_encode_batch.argtypes = [_vp, _u8p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                        _u8p, C.c_uint64, _u64p, _u64p]
    L.isscabac_encode_batch_packed.argtypes = [_vp, _u8p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                        _u8p, C.c_uint64, _u64p, _u64p]
    L.isscabac_decode_batch.argtypes = [_vp, _u8p, _u64p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p,
                                        C.c_int, _u8p, _u32p]
    L.isscabac_decode_batch_packed.argtypes = [_vp, _u8p, _u64p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p,
                                        C.c_int, _u8p, _u32p]
    _u16p = C.POINTER(C.c_uint16)
    L.isscabac_encode_batch16.argtypes = [_vp, _u8p, _u16p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int, _u8p, C.c_uint64,
                                          _u64p, _u64p]
    L.isscabac_decode_batch16.argtypes = [_vp, _u8p, _u64p, _u16p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int, _u8p, _u32p]
    L.isscabac_encode_batch16_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int, _vp, C.c_uint64,
                                              _vp, _u64p]
    L.isscabac_decode_batch16_dev.argtypes = [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                              _vp, _vp]
    L.isscabac_symbol_max_bins.argtypes = [C.POINTER(SymbolParams)]
    L.isscabac_symbol_max_bins.restype = C.c_uint32
    L.isscabac_encode_symbols_dev.argtypes = [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams),
                                              _u8p, C.c_int, _vp, C.c_uint64, _vp, _u64p, _vp]
    L.isscabac_encode_symbols_dev_async.argtypes = [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams),
                                                    _u8p, C.c_int, _vp, C.c_uint64, _vp, _vp]
    L.isscabac_decode_symbols_dev_async.argtypes = [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                                    C.POINTER(SymbolParams), _u8p, C.c_int, _vp, _vp]
    L.isscabac_decode_symbols_dev.argtypes = [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                              C.POINTER(SymbolParams), _u8p, C.c_int, _vp, _vp]
    L.isscabac_encode_symbols.argtypes = [_vp, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams),
                                          _u8p, C.c_int, _u8p, C.c_uint64, _u64p, _u64p, _u32p]
    L.isscabac_encode_symbols_stats.argtypes = [_vp, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                                C.POINTER(SymbolParams), _u8p, C.c_int, _u8p, C.c_uint64, _u64p, _u64p,
                                                _u32p, C.POINTER(SymbolStats)]
    L.isscabac_decode_symbols.argtypes = [_vp, _u8p, _u64p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.POINTER(SymbolParams), _u8p, C.c_int, _u8p, _u32p]
    L.isscabac_iss_init_context_model.argtypes = [_vp, _u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams),
                                                  _f64p, _u8p]
    L.isscabac_iss_init_context_model_dev.argtypes = [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                                      C.POINTER(SymbolParams), _f64p, _u8p]
    L.isscabac_stream_open.argtypes = [_vp, _u8p, C.c_int, C.POINTER(_vp)]
    L.isscabac_stream_close.argtypes = [_vp]
    L.isscabac_stream_close.restype = None
    L.isscabac_stream_encode_start.argtypes = [_vp]
    L.isscabac_stream_encode_bins.argtypes = [_vp, _u8p, _u8p, C.c_uint32]
    L.isscabac_stream_num_bits.argtypes = [_vp, _u64p]
    L.isscabac_stream_encode_finish.argtypes = [_vp, _u8p, C.c_uint64, _u64p]
    L.isscabac_stream_decode_start.argtypes = [_vp, _u8p, C.c_uint64]
    L.isscabac_stream_decode_bins.argtypes = [_vp, _u8p, C.c_uint32, _u8p]
    L.isscabac_stream_decode_finish.argtypes = [_vp, _u32p]
    _u16p = C.POINTER(C.c_uint16)
    L.isscabac_stream_encode_ops.argtypes = [_vp, _u8p, _u16p, C.c_uint32]
    L.isscabac_stream_decode_ops.argtypes = [_vp, _u16p, C.c_uint32, _u8p]
    # host twins of the serial entry points (no engine, no GPU)
    L.isscabac_cpu_stream_open.argtypes = [_u8p, C.c_int, C.POINTER(_vp)]
    L.isscabac_cpu_stream_close.argtypes = [_vp]
    L.isscabac_cpu_stream_close.restype = None
    L.isscabac_cpu_stream_last_error.argtypes = [_vp]
    L.isscabac_cpu_stream_last_error.restype = C.c_char_p
    L.isscabac_cpu_stream_encode_start.argtypes = [_vp]
    L.isscabac_cpu_stream_encode_bins.argtypes = [_vp, _u8p, _u8p, C.c_uint32]
    L.isscabac_cpu_stream_encode_ops.argtypes = [_vp, _u8p, _u16p, C.c_uint32]
    L.isscabac_cpu_stream_num_bits.argtypes = [_vp, _u64p]
    L.isscabac_cpu_stream_encode_finish.argtypes = [_vp, _u8p, C.c_uint64, _u64p]
    L.isscabac_cpu_stream_decode_start.argtypes = [_vp, _u8p, C.c_uint64]
    L.isscabac_cpu_stream_decode_bins.argtypes = [_vp, _u8p, C.c_uint32, _u8p]
    L.isscabac_cpu_stream_decode_ops.argtypes = [_vp, _u16p, C.c_uint32, _u8p]
    L.isscabac_cpu_stream_decode_finish.argtypes = [_vp, _u32p]
    L.isscabac_cpu_stream_set_trace.argtypes = [_vp, C.c_int]
    L.isscabac_cpu_stream_get_stats.argtypes = [_vp, C.c_int, C.c_int, _u8p, C.c_uint64, _u64p, _u32p]
    L.isscabac_cpu_encode_symbols.argtypes = [_u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams), _u8p, C.c_int,
                                              _u8p, C.c_uint64, _u64p, _u64p, _u32p, C.POINTER(SymbolStats)]
    L.isscabac_cpu_decode_symbols.argtypes = [_u8p, _u64p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams), _u8p,
                                              C.c_int, _u8p, _u32p]
    # several devices behind one handle
    L.isscabac_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]
    L.isscabac_destroy_multi.argtypes = [_vp]
    L.isscabac_destroy_multi.restype = None
    L.isscabac_multi_device_count.argtypes = [_vp]
    L.isscabac_multi_peer_access.argtypes = [_vp, C.c_int]
    L.isscabac_multi_engine.argtypes = [_vp, C.c_int]
    L.isscabac_multi_engine.restype = _vp
    L.isscabac_multi_last_error.argtypes = [_vp]
    L.isscabac_multi_last_error.restype = C.c_char_p
    L.isscabac_multi_shard.argtypes = [_vp, C.c_uint64, C.c_int, _u64p, _u64p]
    L.isscabac_multi_shard.restype = None
    L.isscabac_multi_encode_batch.argtypes = [_vp, _u8p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p, C.c_int,
                                              _u8p, C.c_uint64, _u64p, _u64p]
    L.isscabac_multi_decode_batch.argtypes = [_vp, _u8p, _u64p, _u8p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u8p,
                                              C.c_int, _u8p, _u32p]
    L.isscabac_multi_encode_batch_dev.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_uint64, C.c_uint32,
                                                  C.c_uint32, _u8p, C.c_int, _vp, C.c_uint64, _vp, _u64p]
    # (an older build of the ABI loaded for an A/B -- ISSCABAC_LIB, tools/ab_bench.py -- lacks the entry points of this round)
    if not os.environ.get("ISSCABAC_LIB") or hasattr(L, "isscabac_multi_decode_batch_dev"):
        L.isscabac_multi_decode_batch_dev.argtypes = [_vp, _vp, C.c_uint64, _vp, C.POINTER(_vp), C.POINTER(_vp), C.c_uint64,
                                                      C.c_uint32, C.c_uint32, _u8p, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]
        L.isscabac_multi_encode_batch_packed.argtypes = L.isscabac_encode_batch_packed.argtypes
        L.isscabac_multi_decode_batch_packed.argtypes = L.isscabac_decode_batch_packed.argtypes
        L.isscabac_multi_encode_symbols_stats.argtypes = L.isscabac_encode_symbols_stats.argtypes
        L.isscabac_multi_decode_symbols.argtypes = L.isscabac_decode_symbols.argtypes
    # on-disk container
    _strs = C.POINTER(C.c_char_p)
    L.isscabac_write_stream_files.argtypes = [_u8p, _u64p, C.c_uint64, _strs]
    L.isscabac_read_stream_files.argtypes = [_strs, C.c_uint64, _u8p, C.c_uint64, _u64p, _u64p]
    L.isscabac_stream_file_path.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    L.isscabac_iss_container_write.argtypes = [C.c_char_p, C.POINTER(IssSideInfo), _u8p, C.c_uint64, _u8p, C.c_uint64]
    L.isscabac_iss_container_sizes.argtypes = [C.c_char_p, C.POINTER(IssContainerInfo)]
    L.isscabac_iss_container_read.argtypes = [C.c_char_p, _f64p, _f64p, _f64p, _u8p, _u8p, _u8p, _u8p]
    L.isscabac_stream_set_trace.argtypes = [_vp, C.c_int]
    L.isscabac_stream_get_stats.argtypes = [_vp, C.c_int, C.c_int, _u8p, C.c_uint64, _u64p, _u32p]
    L.isscabac_trace_stats.argtypes = [_u32p, C.c_uint64, C.c_int, C.c_int, _u8p, C.c_uint64, _u64p, _u32p]
    L.isscabac_set_profiling.argtypes = [_vp, C.c_int]
    L.isscabac_last_kernel_ms.argtypes = [_vp, C.POINTER(C.c_float)]
    L.isscabac_synth_fill_dev.argtypes = [_vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_int, _vp, _vp]
    L.isscabac_dev_alloc.argtypes = [_vp, C.c_uint64, C.POINTER(_vp)]
    L.isscabac_dev_free.argtypes = [_vp, _vp]
    L.isscabac_copy_h2d.argtypes = [_vp, _vp, _vp, C.c_uint64]
    L.isscabac_copy_d2h.argtypes = [_vp, _vp, _vp, C.c_uint64]
    _lib = L
    return L


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def init_by_prob(p0):
    """ref: SimpleCABACMex('initByProb', ...) -> initContextModelsByP0Prob (CABAC_ContextModelsInit.cpp:82-148)."""
    p0 = _np(p0, np.float64).ravel()
    out = np.zeros(p0.size, np.uint8)
    rc = _load().isscabac_init_by_prob(_ptr(p0, _f64p), p0.size, _ptr(out, _u8p))
    if rc:
        raise EngineError(rc, "init_by_prob: p0 outside [0, 1]")
    return out


def init_by_state(triples):
    """ref: SimpleCABACMex('initByState', ...) -> initContextModelsByMpsState (:51-80); flat [ctxIdx,mps,state]*."""
    t = _np(triples, np.float64).ravel()
    n = t.size // 3
    out = np.zeros(n, np.uint8)
    rc = _load().isscabac_init_by_state(_ptr(t, _f64p), n, _ptr(out, _u8p))
    if rc:
        raise EngineError(rc, "init_by_state: mps must be 0/1 and state 0..62")
    return out


def trace_stats(records, ctx, decoder=False):
    """isscabac_trace_stats: trace records (uint32) of one stream -> (trace [N, 5] uint8, transitions [128, 128] uint32)
    of context `ctx`, the two outputs of the reference's getEncoderStats / getDecoderStats."""
    L = _load()
    rec = _np(records, np.uint32).ravel()
    n = C.c_uint64(0)
    trans = np.zeros((128, 128), np.uint32)
    rc = L.isscabac_trace_stats(_ptr(rec, _u32p), rec.size, int(ctx), int(bool(decoder)), None, 0, C.byref(n), _ptr(trans, _u32p))
    if rc:
        raise EngineError(rc, "isscabac_trace_stats")
    trace = np.zeros((max(1, int(n.value)), 5), np.uint8)
    L.isscabac_trace_stats(_ptr(rec, _u32p), rec.size, int(ctx), int(bool(decoder)), _ptr(trace, _u8p), trace.shape[0], C.byref(n), None)
    return trace[:int(n.value)].copy(), trans


def stream_bound(nbins):
    return int(_load().isscabac_stream_bound(int(nbins)))


class Engine:
    """One engine handle = one GPU + its scratch.  Not thread safe."""

    def __init__(self, device=0):
        self._L = _load()
        h = _vp()
        rc = self._L.isscabac_create(int(device), C.byref(h))
        if rc:
            raise EngineError(rc, self._L.isscabac_last_error(None).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.isscabac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise EngineError(rc, self._L.isscabac_last_error(self._h).decode())

    # ---- configuration -------------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0 = engine's own."""
        self._check(self._L.isscabac_set_stream(self._h, _vp(hip_stream or 0)))

    def set_tuning(self, mode):
        """mode: 'auto' | 'throughput' | 'latency' | 'latency_single' (or the ISSCABAC_TUNE_* integer): which kernel instance runs."""
        m = {"auto": 0, "throughput": 1, "latency": 2, "latency_single": 3}.get(mode, mode)
        self._check(self._L.isscabac_set_tuning(self._h, int(m)))

    def last_instance(self, which):
        """which: 'encode' | 'decode' -> name of the kernel instance the most recent 8-bit batch call launched."""
        return self._L.isscabac_last_instance(self._h, {"encode": 0, "decode": 1}[which]).decode()

    def set_profiling(self, on):
        self._check(self._L.isscabac_set_profiling(self._h, int(bool(on))))

    def last_kernel_ms(self):
        ms = (C.c_float * 4)()
        self._check(self._L.isscabac_last_kernel_ms(self._h, ms))
        return dict(encode=ms[0], scan=ms[1], gather=ms[2], decode=ms[3])

    def kernel_ms_mean(self):
        """Mean kernel times over the batch calls since the previous call of this method (waits for the stream first):
        ({'encode','scan','gather','decode'} in ms, the same keys -> number of launches averaged)."""
        ms, n = (C.c_float * 4)(), (C.c_uint32 * 4)()
        self._check(self._L.isscabac_kernel_ms_mean(self._h, ms, n))
        keys = ("encode", "scan", "gather", "decode")
        return {k: ms[i] for i, k in enumerate(keys)}, {k: int(n[i]) for i, k in enumerate(keys)}

    def kernel_ms_samples(self):
        """Every launch's kernel time since the previous call (waits for the stream first): {'encode': [...], ...} in ms,
        oldest first."""
        out = {}
        for i, k in enumerate(("encode", "scan", "gather", "decode")):
            n = C.c_uint32(0)
            self._check(self._L.isscabac_kernel_ms_samples(self._h, i, None, 0, C.byref(n)))
            buf = (C.c_float * max(1, n.value))()
            self._check(self._L.isscabac_kernel_ms_samples(self._h, i, buf, n.value, C.byref(n)))
            out[k] = [float(buf[j]) for j in range(n.value)]
        return out

    def kernel_ms_samples_dropped(self):
        """launches whose time was not kept since the last call (the series above is truncated if any): per kernel"""
        out = {}
        if not hasattr(self._L, "isscabac_kernel_ms_samples_dropped"):
            return {k: 0 for k in ("encode", "scan", "gather", "decode")}
        for i, k in enumerate(("encode", "scan", "gather", "decode")):
            n = C.c_uint64(0)
            self._check(self._L.isscabac_kernel_ms_samples_dropped(self._h, i, C.byref(n)))
            out[k] = int(n.value)
        return out

    # ---- device-buffer hot path ------------------------------------------------------------------------
    def encode_dev_async(self, d_bins, d_ctx, S, nbins, pitch, ctx_init, d_payload, payload_cap, d_offsets, d_nbins=0):
        """Enqueue only (no host wait, nothing read back): sync() reports the total and a capacity overflow."""
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_encode_batch_dev_async(self._h, _vp(d_bins), _vp(d_ctx), _vp(d_nbins or 0), int(S),
                                                            int(nbins), int(pitch), _ptr(ci, _u8p), ci.size,
                                                            _vp(d_payload), int(payload_cap), _vp(d_offsets)))

    def decode_dev_async(self, d_payload, payload_bytes, d_offsets, d_ctx, S, nbins, pitch, ctx_init, d_bins_out,
                         d_status=0, d_nbins=0):
        """Enqueue only.  payload_bytes: offsets[S], or any upper bound of it the buffer really has (its capacity)."""
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_decode_batch_dev_async(self._h, _vp(d_payload), int(payload_bytes), _vp(d_offsets),
                                                            _vp(d_ctx), _vp(d_nbins or 0), int(S), int(nbins),
                                                            int(pitch), _ptr(ci, _u8p), ci.size, _vp(d_bins_out),
                                                            _vp(d_status or 0)))

    def sync(self):
        """Wait for everything enqueued; returns the payload size of the most recent encode."""
        total = C.c_uint64(0)
        self._check(self._L.isscabac_sync(self._h, C.byref(total)))
        return int(total.value)

    def encode_dev(self, d_bins, d_ctx, S, nbins, pitch, ctx_init, d_payload, payload_cap, d_offsets, d_nbins=0):
        ci = _np(ctx_init, np.uint8)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_batch_dev(self._h, _vp(d_bins), _vp(d_ctx), _vp(d_nbins or 0), int(S),
                                                      int(nbins), int(pitch), _ptr(ci, _u8p), ci.size, _vp(d_payload),
                                                      int(payload_cap), _vp(d_offsets), C.byref(total)))
        return int(total.value)

    def decode_dev(self, d_payload, payload_bytes, d_offsets, d_ctx, S, nbins, pitch, ctx_init, d_bins_out,
                   d_status=0, d_nbins=0):
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_decode_batch_dev(self._h, _vp(d_payload), int(payload_bytes), _vp(d_offsets),
                                                      _vp(d_ctx), _vp(d_nbins or 0), int(S), int(nbins), int(pitch),
                                                      _ptr(ci, _u8p), ci.size, _vp(d_bins_out), _vp(d_status or 0)))

    def synth_dev(self, seed, s0, S, nbins, pitch, nctx, ep_half, d_bins, d_ctx):
        self._check(self._L.isscabac_synth_fill_dev(self._h, int(seed), int(s0), int(S), int(nbins), int(pitch),
                                                    int(nctx), int(bool(ep_half)), _vp(d_bins), _vp(d_ctx)))

    # ---- host-buffer entry points ---------------------------------------------------------------------
    @staticmethod
    def _pad_rows(a, pitch):
        a = _np(a, np.uint8)
        if a.shape[1] == pitch:
            return a
        out = np.zeros((a.shape[0], pitch), np.uint8)
        out[:, :a.shape[1]] = a
        return out

    def encode(self, bins, ctx, ctx_init, nbins=None, out=None):
        """bins, ctx: [S, B] uint8 (stream-major).  Returns (payload uint8[total], offsets uint64[S+1]).
        out: optional (payload buffer uint8[>= S * stream_bound(B)], offsets buffer uint64[S + 1]) to reuse between calls
        -- the results are then views into them (fresh arrays cost a page fault per 4 KiB they receive)."""
        bins = _np(bins, np.uint8)
        ctx = _np(ctx, np.uint8)
        assert bins.ndim == 2 and bins.shape == ctx.shape
        S, B = bins.shape
        pitch = max(16, (B + 15) // 16 * 16)
        bins, ctx = self._pad_rows(bins, pitch), self._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        if out is not None:
            payload, offsets = out
            assert payload.dtype == np.uint8 and offsets.dtype == np.uint64 and offsets.size >= S + 1
            cap = payload.size
        else:
            cap = S * stream_bound(B)
            payload = np.empty(cap, np.uint8)
            offsets = np.zeros(S + 1, np.uint64)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_batch(self._h, _ptr(bins, _u8p), _ptr(ctx, _u8p), _ptr(nb, _u32p), S, B,
                                                  pitch, _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap,
                                                  _ptr(offsets, _u64p), C.byref(total)))
        if out is not None:
            return payload[:int(total.value)], offsets[:S + 1]
        return payload[:int(total.value)].copy(), offsets

    def decode(self, payload, offsets, ctx, ctx_init, nbins=None, out=None):
        """Returns (bins uint8[S, B], status uint32[S]).  out: optional (bins buffer uint8[S, pitch], status buffer
        uint32[S]) to reuse between calls (pitch = B rounded up to 16)."""
        payload = _np(payload, np.uint8)
        offsets = _np(offsets, np.uint64)
        ctx = _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        ctxp = self._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        if out is not None:
            bins, status = out
            assert bins.shape == (S, pitch) and bins.dtype == np.uint8 and bins.flags["C_CONTIGUOUS"] and status.size >= S
        else:
            bins = np.empty((S, pitch), np.uint8)
            status = np.zeros(S, np.uint32)
        self._check(self._L.isscabac_decode_batch(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(ctxp, _u8p),
                                                  _ptr(nb, _u32p), S, B, pitch, _ptr(ci, _u8p), ci.size,
                                                  _ptr(bins, _u8p), _ptr(status, _u32p)))
        return (bins if pitch == B else bins[:, :B].copy()), status

    # ---- more than 255 contexts: 16-bit context indices, 0xFFFF = bypass (isscabac_*_batch16, ref: CommonDef.h:56) ----
    def encode16(self, bins, ctx16, ctx_init, nbins=None):
        """bins [S, B] uint8, ctx16 [S, B] uint16 (0xFFFF = bypass bin), up to 999 contexts.  Returns (payload, offsets)."""
        bins = _np(bins, np.uint8)
        ctx16 = _np(ctx16, np.uint16)
        assert bins.ndim == 2 and bins.shape == ctx16.shape
        S, B = bins.shape
        pitch = max(16, (B + 15) // 16 * 16)
        bins = self._pad_rows(bins, pitch)
        cp = np.zeros((S, pitch), np.uint16)
        cp[:, :B] = ctx16
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        cap = S * stream_bound(B)
        payload, offsets = np.empty(cap, np.uint8), np.zeros(S + 1, np.uint64)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_batch16(self._h, _ptr(bins, _u8p), _ptr(cp, C.POINTER(C.c_uint16)), _ptr(nb, _u32p), S, B,
                                                    pitch, _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap, _ptr(offsets, _u64p),
                                                    C.byref(total)))
        return payload[:int(total.value)].copy(), offsets

    def decode16(self, payload, offsets, ctx16, ctx_init, nbins=None):
        """Returns (bins uint8[S, B], status uint32[S])."""
        payload = _np(payload, np.uint8)
        offsets = _np(offsets, np.uint64)
        ctx16 = _np(ctx16, np.uint16)
        S, B = ctx16.shape
        pitch = max(16, (B + 15) // 16 * 16)
        cp = np.zeros((S, pitch), np.uint16)
        cp[:, :B] = ctx16
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        bins, status = np.empty((S, pitch), np.uint8), np.zeros(S, np.uint32)
        self._check(self._L.isscabac_decode_batch16(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(cp, C.POINTER(C.c_uint16)),
                                                    _ptr(nb, _u32p), S, B, pitch, _ptr(ci, _u8p), ci.size, _ptr(bins, _u8p),
                                                    _ptr(status, _u32p)))
        return bins[:, :B].copy(), status

    def encode16_dev(self, d_bins, d_ctx16, S, nbins, pitch, ctx_init, d_payload, payload_cap, d_offsets, d_nbins=0):
        ci = _np(ctx_init, np.uint8)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_batch16_dev(self._h, _vp(d_bins), _vp(d_ctx16), _vp(d_nbins or 0), int(S), int(nbins),
                                                        int(pitch), _ptr(ci, _u8p), ci.size, _vp(d_payload), int(payload_cap),
                                                        _vp(d_offsets), C.byref(total)))
        return int(total.value)

    def decode16_dev(self, d_payload, payload_bytes, d_offsets, d_ctx16, S, nbins, pitch, ctx_init, d_bins_out, d_status=0, d_nbins=0):
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_decode_batch16_dev(self._h, _vp(d_payload), int(payload_bytes), _vp(d_offsets), _vp(d_ctx16),
                                                        _vp(d_nbins or 0), int(S), int(nbins), int(pitch), _ptr(ci, _u8p), ci.size,
                                                        _vp(d_bins_out), _vp(d_status or 0)))

    # ---- the same with the bins as bits on the host side (an eighth of their bytes over the PCIe link) --------------
    def encode_packed(self, bins_bits, ctx, ctx_init, nbins=None, out=None):
        """bins_bits: [S, pitch / 8] uint8, bin i of a stream at bit i % 8 of byte i / 8 (numpy.packbits(bins, axis=1,
        bitorder="little") of rows padded to pitch); ctx: [S, B] uint8.  Otherwise like encode()."""
        ctx = _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        bits = _np(bins_bits, np.uint8)
        assert bits.shape == (S, pitch // 8), (bits.shape, (S, pitch // 8))
        ctxp = self._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        if out is not None:
            payload, offsets = out
            cap = payload.size
        else:
            cap = S * stream_bound(B)
            payload, offsets = np.empty(cap, np.uint8), np.zeros(S + 1, np.uint64)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_batch_packed(self._h, _ptr(bits, _u8p), _ptr(ctxp, _u8p), _ptr(nb, _u32p), S, B,
                                                         pitch, _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap,
                                                         _ptr(offsets, _u64p), C.byref(total)))
        if out is not None:
            return payload[:int(total.value)], offsets[:S + 1]
        return payload[:int(total.value)].copy(), offsets

    def decode_packed(self, payload, offsets, ctx, ctx_init, nbins=None, out=None):
        """Returns (bins as bits uint8[S, pitch / 8], status uint32[S]); numpy.unpackbits(bits, axis=1, bitorder="little")
        [:, :B] gives the bins.  out: optional (bits buffer, status buffer)."""
        payload, offsets, ctx = _np(payload, np.uint8), _np(offsets, np.uint64), _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        ctxp = self._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        if out is not None:
            bits, status = out
            assert bits.shape == (S, pitch // 8) and bits.dtype == np.uint8 and bits.flags["C_CONTIGUOUS"]
        else:
            bits, status = np.empty((S, pitch // 8), np.uint8), np.zeros(S, np.uint32)
        self._check(self._L.isscabac_decode_batch_packed(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(ctxp, _u8p),
                                                         _ptr(nb, _u32p), S, B, pitch, _ptr(ci, _u8p), ci.size,
                                                         _ptr(bits, _u8p), _ptr(status, _u32p)))
        return bits, status

    # ---- symbol-level batch calls (binariser + context selector fused on the device) --------------------
    def encode_symbols(self, sym, params, ctx_init, nsym=None):
        """sym: [S, R] uint8, one stream (column of symbols) per row.  Returns (payload, offsets, bins_per_stream)."""
        sym = _np(sym, np.uint8)
        S, R = sym.shape
        pitch = max(16, (R + 15) // 16 * 16)
        symp = self._pad_rows(sym, pitch)
        ci = _np(ctx_init, np.uint8)
        ns = None if nsym is None else _np(nsym, np.uint32)
        mb = int(self._L.isscabac_symbol_max_bins(C.byref(params)))
        if mb == 0:
            raise EngineError(1, "unsupported symbol parameters")
        cap = S * stream_bound(R * mb)
        payload = np.empty(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        nbins = np.zeros(S, np.uint32)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_symbols(self._h, _ptr(symp, _u8p), _ptr(ns, _u32p), S, R, pitch,
                                                    C.byref(params), _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap,
                                                    _ptr(offsets, _u64p), C.byref(total), _ptr(nbins, _u32p)))
        return payload[:int(total.value)].copy(), offsets, nbins

    def encode_symbols_stats(self, sym, params, ctx_init, nsym=None, trace=False):
        """encode_symbols plus cabacEncode.m's statistics: returns (payload, offsets, bins_per_stream,
        ctx_hist uint32[S, 64], ctx_cost uint32[S, 64], sym_bits uint16[S, R]) and, with trace=True, the per-bin state
        trace records uint32[S, R * max_bins] (trace_stats() reduces one stream's records to getEncoderStats' outputs)."""
        sym = _np(sym, np.uint8)
        S, R = sym.shape
        pitch = max(16, (R + 15) // 16 * 16)
        symp = self._pad_rows(sym, pitch)
        ci = _np(ctx_init, np.uint8)
        ns = None if nsym is None else _np(nsym, np.uint32)
        mb = int(self._L.isscabac_symbol_max_bins(C.byref(params)))
        if mb == 0:
            raise EngineError(1, "unsupported symbol parameters")
        cap = S * stream_bound(R * mb)
        payload = np.empty(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        nbins = np.zeros(S, np.uint32)
        hist = np.zeros((S, STAT_CTX), np.uint32)
        cost = np.zeros((S, STAT_CTX), np.uint32)
        hb = np.zeros((S, pitch), np.uint16)
        rec = np.zeros((S, R * mb), np.uint32) if trace else None
        st = SymbolStats(hist.ctypes.data_as(C.POINTER(C.c_uint32)), cost.ctypes.data_as(C.POINTER(C.c_uint32)),
                         hb.ctypes.data_as(C.POINTER(C.c_uint16)),
                         rec.ctypes.data_as(C.POINTER(C.c_uint32)) if trace else None, R * mb if trace else 0)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_encode_symbols_stats(self._h, _ptr(symp, _u8p), _ptr(ns, _u32p), S, R, pitch,
                                                          C.byref(params), _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p),
                                                          cap, _ptr(offsets, _u64p), C.byref(total), _ptr(nbins, _u32p),
                                                          C.byref(st)))
        if trace:
            return payload[:int(total.value)].copy(), offsets, nbins, hist, cost, hb[:, :R].copy(), rec
        return payload[:int(total.value)].copy(), offsets, nbins, hist, cost, hb[:, :R].copy()

    def decode_symbols(self, payload, offsets, R, params, ctx_init, nsym=None):
        """Returns (sym uint8[S, R], status uint32[S])."""
        payload = _np(payload, np.uint8)
        offsets = _np(offsets, np.uint64)
        S = offsets.size - 1
        pitch = max(16, (R + 15) // 16 * 16)
        ci = _np(ctx_init, np.uint8)
        ns = None if nsym is None else _np(nsym, np.uint32)
        out = np.zeros((S, pitch), np.uint8)
        status = np.zeros(S, np.uint32)
        self._check(self._L.isscabac_decode_symbols(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(ns, _u32p), S,
                                                    R, pitch, C.byref(params), _ptr(ci, _u8p), ci.size, _ptr(out, _u8p),
                                                    _ptr(status, _u32p)))
        return out[:, :R].copy(), status

    def encode_symbols_dev_async(self, d_sym, S, R, pitch, params, ctx_init, d_payload, payload_cap, d_offsets, d_nbins=0,
                                 d_nsym=0):
        """Enqueue only (device pointers as integers); sync() returns the payload size.  Steps chain with no host wait."""
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_encode_symbols_dev_async(self._h, _vp(d_sym), _vp(d_nsym or 0), S, R, pitch, C.byref(params),
                                                              _ptr(ci, _u8p), ci.size, _vp(d_payload), payload_cap,
                                                              _vp(d_offsets), _vp(d_nbins or 0)))

    def decode_symbols_dev_async(self, d_payload, payload_bytes, d_offsets, S, R, pitch, params, ctx_init, d_sym_out, d_status=0,
                                 d_nsym=0):
        """payload_bytes: the buffer's extent when the size is not on the host yet (streams are cut by the offsets)."""
        ci = _np(ctx_init, np.uint8)
        self._check(self._L.isscabac_decode_symbols_dev_async(self._h, _vp(d_payload), payload_bytes, _vp(d_offsets),
                                                              _vp(d_nsym or 0), S, R, pitch, C.byref(params), _ptr(ci, _u8p),
                                                              ci.size, _vp(d_sym_out), _vp(d_status or 0)))

    def iss_init_context_model(self, sym, params):
        """Empirical p(0) of the ISS contexts over all columns (ref: +coder/cabacInitContextModel.m).
        sym: [S, R] uint8, one column per row.  Returns (p0 float64[7*nlbp+2], uint8(p0*255))."""
        sym = _np(sym, np.uint8)
        S, R = sym.shape
        pitch = max(16, (R + 15) // 16 * 16)
        symp = self._pad_rows(sym, pitch)
        n = 7 * params.nlbp + 2
        p0 = np.zeros(n, np.float64)
        b = np.zeros(n, np.uint8)
        self._check(self._L.isscabac_iss_init_context_model(self._h, _ptr(symp, _u8p), S, R, pitch, C.byref(params),
                                                            _ptr(p0, _f64p), _ptr(b, _u8p)))
        return p0, b

    # ---- raw device memory (for callers without torch) ---------------------------------------------------
    def alloc(self, nbytes):
        p = _vp()
        self._check(self._L.isscabac_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return int(p.value)

    def free(self, d_ptr):
        self._check(self._L.isscabac_dev_free(self._h, _vp(d_ptr)))

    def h2d(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self._L.isscabac_copy_h2d(self._h, _vp(d_dst), _vp(arr.ctypes.data), arr.nbytes))

    def d2h(self, arr, d_src):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self._L.isscabac_copy_d2h(self._h, _vp(arr.ctypes.data), _vp(d_src), arr.nbytes))


def _host_err(rc, what):
    if rc:
        raise EngineError(rc, "%s: %s" % (what, _load().isscabac_last_error(None).decode()))


def write_stream_files(payload, offsets, paths):
    """isscabac_write_stream_files: stream s of the batch -> paths[s], the file the reference would have written."""
    p, o = _np(payload, np.uint8).ravel(), _np(offsets, np.uint64).ravel()
    arr = (C.c_char_p * len(paths))(*[str(x).encode() for x in paths])
    _host_err(_load().isscabac_write_stream_files(_ptr(p, _u8p) if p.size else None, _ptr(o, _u64p), o.size - 1, arr),
              "write_stream_files")


def read_stream_files(paths):
    """isscabac_read_stream_files: one file per stream -> (payload, offsets)."""
    L = _load()
    arr = (C.c_char_p * len(paths))(*[str(x).encode() for x in paths])
    offsets = np.zeros(len(paths) + 1, np.uint64)
    total = C.c_uint64(0)
    rc = L.isscabac_read_stream_files(arr, len(paths), None, 0, _ptr(offsets, _u64p), C.byref(total))
    if rc not in (0, 4):
        _host_err(rc, "read_stream_files")
    payload = np.zeros(max(1, int(total.value)), np.uint8)
    _host_err(L.isscabac_read_stream_files(arr, len(paths), _ptr(payload, _u8p), payload.size, _ptr(offsets, _u64p),
                                           C.byref(total)), "read_stream_files")
    return payload[:int(total.value)].copy(), offsets


def stream_file_path(prefix, s):
    buf = C.create_string_buffer(4096)
    _host_err(_load().isscabac_stream_file_path(str(prefix).encode(), int(s), buf, 4096), "stream_file_path")
    return buf.value.decode()


def iss_container_write(path, cW, cH, Q, xw, xh, w_stream, h_stream):
    """side information {cW, cH, Q, xw, xh} + the W and H bitstreams -> one file (isscabac_iss_container_write)."""
    cW, cH = _np(cW, np.float64).ravel(), _np(cH, np.float64).ravel()
    Q = np.asarray(Q, np.float64)
    Q = Q.reshape(Q.shape[0], -1) if Q.ndim else Q.reshape(1, 1)
    Qf = np.asfortranarray(Q)
    xw, xh = _np(xw, np.uint8).ravel(), _np(xh, np.uint8).ravel()
    w, h = _np(w_stream, np.uint8).ravel(), _np(h_stream, np.uint8).ravel()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a.size else None
    bp = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)) if a.size else None
    si = IssSideInfo(dp(cW), dp(cH), Qf.ctypes.data_as(C.POINTER(C.c_double)) if Qf.size else None, bp(xw), bp(xh), cW.size,
                     cH.size, Q.shape[0], Q.shape[1], xw.size, xh.size)
    _host_err(_load().isscabac_iss_container_write(str(path).encode(), C.byref(si), bp(w), w.size, bp(h), h.size),
              "iss_container_write")


def iss_container_read(path):
    """-> dict(cW, cH, Q, xw, xh, w_stream, h_stream)"""
    L = _load()
    n = IssContainerInfo()
    _host_err(L.isscabac_iss_container_sizes(str(path).encode(), C.byref(n)), "iss_container_sizes")
    cW, cH = np.zeros(n.ncW), np.zeros(n.ncH)
    Q = np.zeros((n.q_rows, n.q_cols), order="F")
    xw, xh = np.zeros(n.nxw, np.uint8), np.zeros(n.nxh, np.uint8)
    w, h = np.zeros(n.w_bytes, np.uint8), np.zeros(n.h_bytes, np.uint8)
    pp = lambda a, t: a.ctypes.data_as(t) if a.size else None
    _host_err(L.isscabac_iss_container_read(str(path).encode(), pp(cW, _f64p), pp(cH, _f64p), pp(Q, _f64p), pp(xw, _u8p),
                                            pp(xh, _u8p), pp(w, _u8p), pp(h, _u8p)), "iss_container_read")
    return dict(cW=cW, cH=cH, Q=np.ascontiguousarray(Q), xw=xw, xh=xh, w_stream=w, h_stream=h)


class MultiEngine:
    """isscabac_multi: several GPUs behind one handle in one process.  Streams are split into contiguous blocks, one per
    device; results equal a single engine's."""

    def __init__(self, devices):
        self._L = _load()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = _vp()
        rc = self._L.isscabac_create_multi(ids, len(devices), C.byref(h))
        if rc:
            raise EngineError(rc, self._L.isscabac_last_error(None).decode())
        self._h = h
        self.n = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._L.isscabac_destroy_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise EngineError(rc, self._L.isscabac_multi_last_error(self._h).decode())

    def shard(self, S, g):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        self._L.isscabac_multi_shard(self._h, int(S), int(g), C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def engine_handle(self, g):
        return self._L.isscabac_multi_engine(self._h, int(g))

    def peer_access(self, g):
        """how device g's shard reaches the root in the gather: 0 same device, 1 peer to peer, 2 staged through the host"""
        return int(self._L.isscabac_multi_peer_access(self._h, int(g)))

    def encode(self, bins, ctx, ctx_init, nbins=None):
        bins, ctx = _np(bins, np.uint8), _np(ctx, np.uint8)
        S, B = bins.shape
        pitch = max(16, (B + 15) // 16 * 16)
        bp, cp = Engine._pad_rows(bins, pitch), Engine._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        cap = S * stream_bound(B)
        payload = np.empty(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_multi_encode_batch(self._h, _ptr(bp, _u8p), _ptr(cp, _u8p), _ptr(nb, _u32p), S, B, pitch,
                                                        _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap, _ptr(offsets, _u64p),
                                                        C.byref(total)))
        return payload[:int(total.value)].copy(), offsets

    def decode(self, payload, offsets, ctx, ctx_init, nbins=None):
        payload, offsets, ctx = _np(payload, np.uint8), _np(offsets, np.uint64), _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        cp = Engine._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        out = np.zeros((S, pitch), np.uint8)
        status = np.zeros(S, np.uint32)
        self._check(self._L.isscabac_multi_decode_batch(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(cp, _u8p),
                                                        _ptr(nb, _u32p), S, B, pitch, _ptr(ci, _u8p), ci.size, _ptr(out, _u8p),
                                                        _ptr(status, _u32p)))
        return out[:, :B].copy(), status

    def encode_dev(self, d_bins, d_ctx, S, nbins, pitch, ctx_init, d_payload_root, payload_cap, d_offsets_root, d_nbins=None):
        """d_bins / d_ctx: one device address per device (that device's block of streams).  Payload + offsets are gathered
        on the first device.  Returns the payload size."""
        ci = _np(ctx_init, np.uint8)
        pb = (_vp * self.n)(*[_vp(int(x)) for x in d_bins])
        pc = (_vp * self.n)(*[_vp(int(x)) for x in d_ctx])
        pn = (_vp * self.n)(*[_vp(int(x) or None) for x in d_nbins]) if d_nbins is not None else None
        total = C.c_uint64(0)
        self._check(self._L.isscabac_multi_encode_batch_dev(self._h, pb, pc, pn, int(S), int(nbins), int(pitch), _ptr(ci, _u8p),
                                                            ci.size, _vp(d_payload_root), int(payload_cap), _vp(d_offsets_root),
                                                            C.byref(total)))
        return int(total.value)

    def decode_dev(self, d_payload_root, payload_bytes, d_offsets_root, d_ctx, S, nbins, pitch, ctx_init, d_bins_out,
                   d_status=None, d_nbins=None):
        """the way back: payload + offsets on the first device, every device decodes its block into its own d_bins_out[g]"""
        ci = _np(ctx_init, np.uint8)
        arr = lambda xs: (_vp * self.n)(*[_vp(int(x) or None) for x in xs]) if xs is not None else None
        self._check(self._L.isscabac_multi_decode_batch_dev(self._h, _vp(d_payload_root), int(payload_bytes), _vp(d_offsets_root),
                                                            arr(d_ctx), arr(d_nbins), int(S), int(nbins), int(pitch),
                                                            _ptr(ci, _u8p), ci.size, arr(d_bins_out), arr(d_status)))

    def encode_packed(self, bins_bits, ctx, ctx_init, nbins=None):
        """bins as bits (numpy.packbits(bins, axis=1, bitorder="little") of rows padded to the pitch): [S, pitch / 8]"""
        bits, ctx = _np(bins_bits, np.uint8), _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        assert bits.shape == (S, pitch // 8)
        cp = Engine._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        cap = S * stream_bound(B)
        payload = np.empty(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        total = C.c_uint64(0)
        self._check(self._L.isscabac_multi_encode_batch_packed(self._h, _ptr(bits, _u8p), _ptr(cp, _u8p), _ptr(nb, _u32p), S, B,
                                                               pitch, _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap,
                                                               _ptr(offsets, _u64p), C.byref(total)))
        return payload[:int(total.value)].copy(), offsets

    def decode_packed(self, payload, offsets, ctx, ctx_init, nbins=None):
        payload, offsets, ctx = _np(payload, np.uint8), _np(offsets, np.uint64), _np(ctx, np.uint8)
        S, B = ctx.shape
        pitch = max(16, (B + 15) // 16 * 16)
        cp = Engine._pad_rows(ctx, pitch)
        ci = _np(ctx_init, np.uint8)
        nb = None if nbins is None else _np(nbins, np.uint32)
        out = np.zeros((S, pitch // 8), np.uint8)
        status = np.zeros(S, np.uint32)
        self._check(self._L.isscabac_multi_decode_batch_packed(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(cp, _u8p),
                                                               _ptr(nb, _u32p), S, B, pitch, _ptr(ci, _u8p), ci.size,
                                                               _ptr(out, _u8p), _ptr(status, _u32p)))
        return out, status

    def encode_symbols(self, sym, params, ctx_init, nsym=None, stats=False):
        """as Engine.encode_symbols / encode_symbols_stats, the streams split over the devices"""
        sym = _np(sym, np.uint8)
        S, R = sym.shape
        pitch = max(16, (R + 15) // 16 * 16)
        symp = Engine._pad_rows(sym, pitch)
        ci = _np(ctx_init, np.uint8)
        ns = None if nsym is None else _np(nsym, np.uint32)
        mb = int(self._L.isscabac_symbol_max_bins(C.byref(params)))
        if mb == 0:
            raise EngineError(1, "unsupported symbol parameters")
        cap = S * stream_bound(R * mb)
        payload = np.empty(cap, np.uint8)
        offsets = np.zeros(S + 1, np.uint64)
        nbins = np.zeros(S, np.uint32)
        total = C.c_uint64(0)
        st, hist, cost, hb = None, None, None, None
        if stats:
            hist, cost = np.zeros((S, STAT_CTX), np.uint32), np.zeros((S, STAT_CTX), np.uint32)
            hb = np.zeros((S, pitch), np.uint16)
            st = SymbolStats(hist.ctypes.data_as(_u32p), cost.ctypes.data_as(_u32p), hb.ctypes.data_as(C.POINTER(C.c_uint16)), None, 0)
        self._check(self._L.isscabac_multi_encode_symbols_stats(self._h, _ptr(symp, _u8p), _ptr(ns, _u32p), S, R, pitch,
                                                                C.byref(params), _ptr(ci, _u8p), ci.size, _ptr(payload, _u8p), cap,
                                                                _ptr(offsets, _u64p), C.byref(total), _ptr(nbins, _u32p),
                                                                C.byref(st) if st is not None else None))
        out = (payload[:int(total.value)].copy(), offsets, nbins)
        return out + (hist, cost, hb[:, :R].copy()) if stats else out

    def decode_symbols(self, payload, offsets, R, params, ctx_init, nsym=None):
        payload, offsets = _np(payload, np.uint8), _np(offsets, np.uint64)
        S = offsets.size - 1
        pitch = max(16, (R + 15) // 16 * 16)
        ci = _np(ctx_init, np.uint8)
        ns = None if nsym is None else _np(nsym, np.uint32)
        out = np.zeros((S, pitch), np.uint8)
        status = np.zeros(S, np.uint32)
        self._check(self._L.isscabac_multi_decode_symbols(self._h, _ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(ns, _u32p), S,
                                                          R, pitch, C.byref(params), _ptr(ci, _u8p), ci.size, _ptr(out, _u8p),
                                                          _ptr(status, _u32p)))
        return out[:, :R].copy(), status


class Stream:
    """One reference handle (`class CABAC`): per-bin encode / decode on device-resident state.
    ref: CABAC/SimpleCABACMex.cpp:69-80, commands :186-354."""

    def __init__(self, engine, ctx_init):
        self._eng = engine
        self._L = engine._L
        ci = _np(ctx_init, np.uint8)
        h = _vp()
        engine._check(self._L.isscabac_stream_open(engine._h, _ptr(ci, _u8p), ci.size, C.byref(h)))
        self._s = h

    def close(self):
        if getattr(self, "_s", None):
            self._L.isscabac_stream_close(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_start(self):
        self._eng._check(self._L.isscabac_stream_encode_start(self._s))

    def encode_bins(self, bins, ctx):
        b, c = _np(bins, np.uint8).ravel(), _np(ctx, np.uint8).ravel()
        assert b.size == c.size
        self._eng._check(self._L.isscabac_stream_encode_bins(self._s, _ptr(b, _u8p), _ptr(c, _u8p), b.size))

    def encode_ops(self, bins, ops):
        """16-bit ops (ISSCABAC_OP_*): a context index up to 998, OP_BYPASS, or op_prob(p)."""
        b, o = _np(bins, np.uint8).ravel(), _np(ops, np.uint16).ravel()
        assert b.size == o.size
        self._eng._check(self._L.isscabac_stream_encode_ops(self._s, _ptr(b, _u8p), _ptr(o, C.POINTER(C.c_uint16)), b.size))

    def decode_ops(self, ops):
        o = _np(ops, np.uint16).ravel()
        out = np.zeros(o.size, np.uint8)
        self._eng._check(self._L.isscabac_stream_decode_ops(self._s, _ptr(o, C.POINTER(C.c_uint16)), o.size, _ptr(out, _u8p)))
        return out

    def set_trace(self, on=True):
        self._eng._check(self._L.isscabac_stream_set_trace(self._s, int(bool(on))))

    def get_stats(self, ctx, decoder=False):
        """(trace [N, 5] uint8, transitions [128, 128] uint32): getEncoderStats / getDecoderStats of context `ctx`."""
        n = C.c_uint64(0)
        trans = np.zeros((128, 128), np.uint32)
        self._eng._check(self._L.isscabac_stream_get_stats(self._s, int(bool(decoder)), int(ctx), None, 0, C.byref(n),
                                                           _ptr(trans, _u32p)))
        trace = np.zeros((max(1, int(n.value)), 5), np.uint8)
        self._eng._check(self._L.isscabac_stream_get_stats(self._s, int(bool(decoder)), int(ctx), _ptr(trace, _u8p),
                                                           trace.shape[0], C.byref(n), None))
        return trace[:int(n.value)].copy(), trans

    def num_bits(self):
        v = C.c_uint64(0)
        self._eng._check(self._L.isscabac_stream_num_bits(self._s, C.byref(v)))
        return int(v.value)

    def encode_finish(self):
        n = C.c_uint64(0)
        buf = np.empty(1 << 12, np.uint8)
        rc = self._L.isscabac_stream_encode_finish(self._s, _ptr(buf, _u8p), buf.size, C.byref(n))
        if rc == 4:  # capacity: ask again with the reported size
            buf = np.empty(int(n.value), np.uint8)
            rc = self._L.isscabac_stream_encode_finish(self._s, _ptr(buf, _u8p), buf.size, C.byref(n))
        self._eng._check(rc)
        return buf[:int(n.value)].copy()

    def decode_start(self, payload):
        p = _np(payload, np.uint8).ravel()
        self._eng._check(self._L.isscabac_stream_decode_start(self._s, _ptr(p, _u8p), p.size))

    def decode_bins(self, ctx):
        c = _np(ctx, np.uint8).ravel()
        out = np.zeros(c.size, np.uint8)
        self._eng._check(self._L.isscabac_stream_decode_bins(self._s, _ptr(c, _u8p), c.size, _ptr(out, _u8p)))
        return out

    def decode_finish(self):
        st = C.c_uint32(0)
        self._eng._check(self._L.isscabac_stream_decode_finish(self._s, C.byref(st)))
        return int(st.value)


class CpuStream:
    """Host twin of Stream (isscabac_cpu_stream_*): the reference's per-bin commands served by the engine's own per-lane
    code compiled for the host -- no engine, no GPU.  Same methods, same bytes, same getNumBits cadence, same status words."""

    def __init__(self, ctx_init):
        self._L = _load()
        ci = _np(ctx_init, np.uint8)
        h = _vp()
        rc = self._L.isscabac_cpu_stream_open(_ptr(ci, _u8p), ci.size, C.byref(h))
        if rc:
            raise EngineError(rc, "isscabac_cpu_stream_open: bad context initialisation")
        self._s = h

    def close(self):
        if getattr(self, "_s", None):
            self._L.isscabac_cpu_stream_close(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise EngineError(rc, self._L.isscabac_cpu_stream_last_error(self._s).decode())

    def encode_start(self):
        self._check(self._L.isscabac_cpu_stream_encode_start(self._s))

    def encode_bins(self, bins, ctx):
        b, c = _np(bins, np.uint8).ravel(), _np(ctx, np.uint8).ravel()
        assert b.size == c.size
        self._check(self._L.isscabac_cpu_stream_encode_bins(self._s, _ptr(b, _u8p), _ptr(c, _u8p), b.size))

    def encode_ops(self, bins, ops):
        b, o = _np(bins, np.uint8).ravel(), _np(ops, np.uint16).ravel()
        assert b.size == o.size
        self._check(self._L.isscabac_cpu_stream_encode_ops(self._s, _ptr(b, _u8p), _ptr(o, C.POINTER(C.c_uint16)), b.size))

    def num_bits(self):
        v = C.c_uint64(0)
        self._check(self._L.isscabac_cpu_stream_num_bits(self._s, C.byref(v)))
        return int(v.value)

    def encode_finish(self):
        n = C.c_uint64(0)
        buf = np.empty(1 << 12, np.uint8)
        rc = self._L.isscabac_cpu_stream_encode_finish(self._s, _ptr(buf, _u8p), buf.size, C.byref(n))
        if rc == 4:
            buf = np.empty(int(n.value), np.uint8)
            rc = self._L.isscabac_cpu_stream_encode_finish(self._s, _ptr(buf, _u8p), buf.size, C.byref(n))
        self._check(rc)
        return buf[:int(n.value)].copy()

    def decode_start(self, payload):
        p = _np(payload, np.uint8).ravel()
        self._check(self._L.isscabac_cpu_stream_decode_start(self._s, _ptr(p, _u8p), p.size))

    def decode_bins(self, ctx):
        c = _np(ctx, np.uint8).ravel()
        out = np.zeros(c.size, np.uint8)
        self._check(self._L.isscabac_cpu_stream_decode_bins(self._s, _ptr(c, _u8p), c.size, _ptr(out, _u8p)))
        return out

    def decode_ops(self, ops):
        o = _np(ops, np.uint16).ravel()
        out = np.zeros(o.size, np.uint8)
        self._check(self._L.isscabac_cpu_stream_decode_ops(self._s, _ptr(o, C.POINTER(C.c_uint16)), o.size, _ptr(out, _u8p)))
        return out

    def decode_finish(self):
        st = C.c_uint32(0)
        self._check(self._L.isscabac_cpu_stream_decode_finish(self._s, C.byref(st)))
        return int(st.value)

    def set_trace(self, on=True):
        self._check(self._L.isscabac_cpu_stream_set_trace(self._s, int(bool(on))))

    def get_stats(self, ctx, decoder=False):
        n = C.c_uint64(0)
        trans = np.zeros((128, 128), np.uint32)
        self._check(self._L.isscabac_cpu_stream_get_stats(self._s, int(bool(decoder)), int(ctx), None, 0, C.byref(n), _ptr(trans, _u32p)))
        trace = np.zeros((max(1, int(n.value)), 5), np.uint8)
        self._check(self._L.isscabac_cpu_stream_get_stats(self._s, int(bool(decoder)), int(ctx), _ptr(trace, _u8p), trace.shape[0],
                                                          C.byref(n), None))
        return trace[:int(n.value)].copy(), trans


def cpu_encode_symbols(sym, params, ctx_init, nsym=None, stats=False, trace=False):
    """Host twin of Engine.encode_symbols / encode_symbols_stats (isscabac_cpu_encode_symbols): a few streams of symbols, e.g.
    a whole matrix as ONE stream (params.col_rows), coded one after the other on the calling core.  Returns (payload, offsets,
    bins_per_stream) and, with stats=True, ctx_hist, ctx_cost, sym_bits (and the per-bin trace records with trace=True)."""
    L = _load()
    sym = _np(sym, np.uint8)
    S, R = sym.shape
    pitch = max(16, (R + 15) // 16 * 16)
    symp = Engine._pad_rows(sym, pitch)
    ci = _np(ctx_init, np.uint8)
    ns = None if nsym is None else _np(nsym, np.uint32)
    mb = int(L.isscabac_symbol_max_bins(C.byref(params)))
    if mb == 0:
        raise EngineError(1, "unsupported symbol parameters")
    cap = S * stream_bound(R * mb)
    payload = np.empty(cap, np.uint8)
    offsets = np.zeros(S + 1, np.uint64)
    nbins = np.zeros(S, np.uint32)
    total = C.c_uint64(0)
    st = None
    if stats or trace:
        hist = np.zeros((S, STAT_CTX), np.uint32)
        cost = np.zeros((S, STAT_CTX), np.uint32)
        hb = np.zeros((S, pitch), np.uint16)
        rec = np.zeros((S, R * mb), np.uint32) if trace else None
        st = SymbolStats(hist.ctypes.data_as(C.POINTER(C.c_uint32)), cost.ctypes.data_as(C.POINTER(C.c_uint32)),
                         hb.ctypes.data_as(C.POINTER(C.c_uint16)), rec.ctypes.data_as(C.POINTER(C.c_uint32)) if trace else None,
                         R * mb if trace else 0)
    rc = L.isscabac_cpu_encode_symbols(_ptr(symp, _u8p), _ptr(ns, _u32p), S, R, pitch, C.byref(params), _ptr(ci, _u8p), ci.size,
                                       _ptr(payload, _u8p), cap, _ptr(offsets, _u64p), C.byref(total), _ptr(nbins, _u32p),
                                       C.byref(st) if st is not None else None)
    if rc:
        raise EngineError(rc, "isscabac_cpu_encode_symbols")
    out = (payload[:int(total.value)].copy(), offsets, nbins)
    if st is not None:
        out += (hist, cost, hb[:, :R].copy())
        if trace:
            out += (rec,)
    return out


def cpu_decode_symbols(payload, offsets, R, params, ctx_init, nsym=None):
    """Host twin of Engine.decode_symbols.  Returns (sym uint8[S, R], status uint32[S])."""
    L = _load()
    payload = _np(payload, np.uint8)
    offsets = _np(offsets, np.uint64)
    S = offsets.size - 1
    pitch = max(16, (R + 15) // 16 * 16)
    ci = _np(ctx_init, np.uint8)
    ns = None if nsym is None else _np(nsym, np.uint32)
    out = np.zeros((S, pitch), np.uint8)
    status = np.zeros(S, np.uint32)
    rc = L.isscabac_cpu_decode_symbols(_ptr(payload, _u8p), _ptr(offsets, _u64p), _ptr(ns, _u32p), S, R, pitch, C.byref(params),
                                       _ptr(ci, _u8p), ci.size, _ptr(out, _u8p), _ptr(status, _u32p))
    if rc:
        raise EngineError(rc, "isscabac_cpu_decode_symbols")
    return out[:, :R].copy(), status
