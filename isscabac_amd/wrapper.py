"""Python mirror of the reference's MATLAB handle class (ref: CABAC/cabacWrapper.m:15-77): same constructor,
same method names and argument meaning, same error behaviour.  The per-bin commands are served by the engine's host twin
(isscabac_cpu_stream_*: the kernels' own per-lane code compiled for the host -- one bin per call has no parallelism to give a
GPU) or, with per_bin="gpu" / an engine handed in, by the device-resident handle; the batch methods run on the GPU.  It is what
matlab/cabacWrapper.m does through the MEX shim, written in Python so that the test-suite can drive the reference's
own usage patterns (cabacDemo.m, +coder/cabacEncode.m) on the GPU box, where no MATLAB exists.

Added on top of the reference surface: `encodeBatch` / `decodeBatch` (many independent streams per call).
"""
import numpy as np

from . import engine as _engine

_default_engine = None


def _get_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = _engine.Engine(0)
    return _default_engine


class cabacWrapper:
    def __init__(self, cm, fn, initByProb=1, engine=None, per_bin=None):
        # ref: cabacWrapper.m:15-30.  per_bin: "host" (default) or "gpu" (default when an engine is handed in)
        if not isinstance(fn, str):
            raise ValueError("bitStreamName should be string")
        self.bitStreamName = fn
        self.contextModelInitOptions = np.asarray(cm, np.float64)
        if self.contextModelInitOptions.ndim > 2:
            raise ValueError("context model initialization options should be like this: "
                             "[ctxID mps probIndex, ctxID mps probIndex, ...]")
        self._per_bin = per_bin or ("gpu" if engine is not None else "host")
        if self._per_bin not in ("host", "gpu"):
            raise ValueError("per_bin must be 'host' or 'gpu'")
        self._engine_arg = engine
        self.cabac_handle = None
        self.init(initByProb)

    def init(self, initByProb=1):
        # ref: cabacWrapper.m:31-39 -> SimpleCABACMex('initByProb' | 'initByState', fn, ctxInit), :120-185
        flat = self.contextModelInitOptions.ravel(order="F")  # the MEX reads the MATLAB array column-major
        self.ctx_init = _engine.init_by_prob(flat) if initByProb else _engine.init_by_state(flat)
        if self._per_bin == "gpu":
            self.cabac_handle = _engine.Stream(self._engine, self.ctx_init)
        else:
            self.cabac_handle = _engine.CpuStream(self.ctx_init)
        self.cabac_handle.set_trace(True)  # what the reference's Windows build always records (getEncoderStats)

    @property
    def _engine(self):  # the GPU engine, created when first needed (never for per-bin work on the host twin)
        return self._engine_arg or _get_engine()

    # ---- per-bin commands ----------------------------------------------------------------------------
    def encodeStart(self):
        # ref: SimpleCABACMex.cpp:186-209 opens fn for writing; the file is written at encodeFinish here
        try:
            open(self.bitStreamName, "wb").close()
        except OSError:
            raise RuntimeError("Error: bitstreamfile access error")
        self.cabac_handle.encode_start()

    def encodeBin(self, binValue, ctxID):
        # ref: cabacWrapper.m:45-52 (client-side check) and SimpleCABACMex.cpp:210-247
        if binValue not in (0, 1):
            raise ValueError("bin value to be encoded should either be 1 or 0")
        self.cabac_handle.encode_ops([int(binValue)], [self._ctx_op(ctxID)])

    def getNumBits(self):
        return self.cabac_handle.num_bits()  # ref: SimpleCABACMex.cpp:248-264

    def encodeFinish(self):
        payload = self.cabac_handle.encode_finish()  # ref: :265-279
        with open(self.bitStreamName, "wb") as f:
            f.write(payload.tobytes())

    def decodeStart(self):
        try:
            with open(self.bitStreamName, "rb") as f:  # ref: :280-300
                data = f.read()
        except OSError:
            raise RuntimeError("Error: bitstreamfile access error")
        self.cabac_handle.decode_start(np.frombuffer(data, np.uint8))

    def decodeBin(self, ctxID):
        return int(self.cabac_handle.decode_ops([self._ctx_op(ctxID)])[0])  # ref: :301-335

    def decodeFinish(self):
        # ref: :336-354; the reference asserts on a broken stream, here the flags come back (0 = intact)
        return self.cabac_handle.decode_finish()

    def _ctx_op(self, ctxID):
        # the reference does not range-check the context (ref: :223-224) and indexes a 1000-entry array with it; here an
        # index outside the initialised contexts is an error, never an alias of another context or a bypass bin
        c = float(ctxID)
        if c != int(c) or not 0 <= c < self.ctx_init.size:
            raise ValueError("Error: invalid context index")
        return int(c)

    def getEncoderStats(self, ctxID):
        """[trace, stats] = getEncoderStats(ctxID) of the reference's Windows build (ref: cabacWrapper.m:68-70,
        SimpleCABACMex.cpp:356-412): trace 5 x N uint8 (one column per bin coded with the context: bin, state before,
        mps before, state after, mps after; states numbered like ContextModel.cpp:128-129), stats 128 x 128 uint32."""
        trace, trans = self.cabac_handle.get_stats(self._ctx_op(ctxID), decoder=False)
        return trace.T.copy(), trans

    def getDecoderStats(self, ctxID):
        trace, trans = self.cabac_handle.get_stats(self._ctx_op(ctxID), decoder=True)  # ref: :413-466
        return trace.T.copy(), trans

    # ---- additive: the engine's other two bin kinds, per bin (the reference's MEX reaches neither) -----------------
    def encodeBinEP(self, binValue):
        self.cabac_handle.encode_ops([int(binValue)], [_engine.OP_BYPASS])  # ref: CABAC_ArithmeticEncoder.cpp:250-270

    def decodeBinEP(self):
        return int(self.cabac_handle.decode_ops([_engine.OP_BYPASS])[0])

    def encodeBinProb(self, binValue, probability):
        """probability of the bin being 1 in percent (ref: encodeBinProb, CABAC_ArithmeticEncoder.cpp:180-243)"""
        self.cabac_handle.encode_ops([int(binValue)], [_engine.op_prob(probability)])

    def decodeBinProb(self, probability):
        return int(self.cabac_handle.decode_ops([_engine.op_prob(probability)])[0])

    # ---- batch extension -------------------------------------------------------------------------------
    def encodeBatch(self, bins, ctx, nbins=None):
        """bins, ctx: [S, B] uint8, one independent stream per row (a MATLAB B x S matrix).  Every stream starts
        from this wrapper's context initialisation.  Returns (payload, offsets)."""
        return self._engine.encode(bins, ctx, self.ctx_init, nbins)

    def decodeBatch(self, payload, offsets, ctx, nbins=None):
        return self._engine.decode(payload, offsets, ctx, self.ctx_init, nbins)


# ---- on-disk container: the reference keeps ONE FILE PER STREAM (SimpleCABACMex.cpp:195,288) --------------------------
# isscabac_write_stream_files / isscabac_read_stream_files of the C ABI (csrc/cabac_container.cpp)
write_stream_files = _engine.write_stream_files
read_stream_files = _engine.read_stream_files
