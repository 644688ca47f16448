"""isscabac_amd -- MI355X-native batched CABAC engine (drop-in for the SimpleCABACMex / cabacWrapper path
of christianrohlfing/ISScabac).  The compute lives in lib/libisscabac_hip.so (HIP, gfx950) behind the C ABI of
include/isscabac.h; this package is the thin host-side binding."""
from .engine import (BYPASS, CpuStream, Engine, EngineError, MultiEngine, Stream, SymbolParams, init_by_prob, init_by_state, lib_path,  # noqa: F401
                     cpu_decode_symbols, cpu_encode_symbols, stream_bound)
from .wrapper import cabacWrapper  # noqa: F401
from . import coder  # noqa: F401

