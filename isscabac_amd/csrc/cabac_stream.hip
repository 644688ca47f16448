// cabac_stream.hip -- single-stream handle of include/isscabac.h: the reference's per-bin commands
// (encodeStart / encodeBin / getNumBits / encodeFinish / decodeStart / decodeBin / decodeFinish,
// ref: CABAC/SimpleCABACMex.cpp:186-354) served by one-lane HIP kernels on device-resident coder state.
//
// Same per-lane code as the batch kernels (cabac_lane.h); the differences are the cadence -- bytes leave `low`
// after EVERY bin, exactly like testAndWriteOut (ref: CABAC_ArithmeticEncoder.cpp:369-375), so the byte count that
// getNumBits reports between bins is the reference's -- and that the state survives between launches.
//
// One handle holds up to 999 context models per side like the reference's CABAC_ContextModels (ref: CommonDef.h:56,
// CABAC_ContextModelsInit.cpp:53,84), so an op names its context in 16 bits (ISSCABAC_OP_*).  What a call hands over
// travels in the kernel's arguments when it is short (the reference's MATLAB loops code ONE bin per call), and what a
// call observes (bytes written so far, decoded bins, the status word) comes back through a pinned host mailbox the
// kernel writes: an encodeBin + getNumBits pair costs one launch and one stream synchronisation, no copies.
//
// With tracing on, every op leaves a 4-byte record on the device: getEncoderStats / getDecoderStats of the reference's
// Windows build (ref: SimpleCABACMex.cpp:356-466, ContextModel.cpp:126-134) are computed from those records.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "cabac_internal.h"
#include "cabac_lane.h"

namespace isscabac {

__device__ const uint32_t s_lps_rows[64] = {ISSCABAC_LPS_ROWS};
__device__ const uint32_t s_prob_rows[49] = {ISSCABAC_PROB_ROWS};
struct StreamNextTab {
  uint32_t e[64];
  constexpr StreamNextTab() : e{} {
    constexpr uint8_t trans[64] = {ISSCABAC_TRANS_LPS};
    for (uint32_t s = 0; s < 64; s++) e[s] = next_state_entry(s, trans[s]);
  }
};
__device__ const StreamNextTab s_next{};

constexpr int kStreamCtx = 1000;  // RWTH_MAX_NUM_CONTEXTS, ref: CommonDef.h:56

struct StreamDev {
  EncCore enc;
  uint32_t enc_bytes;  // bytes written to the sink so far
  RefDec dec;  // the per-bin handle decodes in the reference's own formulation (cabac_lane.h: RefDec): one bin per call,
               // nothing to gain from look-ahead, and exact even on streams no encoder produces
  uint32_t dec_pos, dec_len, dec_last;  // bytes handed out so far, payload length, the last byte handed out
  uint8_t q_enc[kStreamCtx], q_dec[kStreamCtx];  // doubled context states, encoder and decoder sets (ref: :175-176)
};

// what the kernels report to the host without a copy: one pinned, device-visible block per handle
struct StreamMailbox {
  uint32_t enc_bytes;
  uint32_t status;
  uint8_t bins[64];  // decoded bins of a short call
};

// ops of a short call travel by value
constexpr uint32_t kInlineOps = 32;
struct InlineOps {
  uint16_t op[kInlineOps];
  uint8_t bin[kInlineOps];
};

struct GlobalStates {
  uint8_t *q;
  __device__ uint32_t load(uint32_t c) const { return q[c]; }
  __device__ void store(uint32_t c, uint32_t v) const { q[c] = (uint8_t)v; }
  __device__ uint32_t addr(uint32_t c) const { return c; }
  __device__ uint32_t load_at(uint32_t a) const { return q[a]; }
  __device__ void store_at(uint32_t a, uint32_t v) const { q[a] = (uint8_t)v; }
  __device__ uint32_t lps_row(uint32_t v) const { return s_lps_rows[v >> 2]; }
  __device__ uint32_t next_entry(uint32_t v) const { return s_next.e[v >> 2]; }
};
struct GlobalSink {
  uint8_t *buf;
  uint32_t pos;
  __device__ void put(uint32_t b) { buf[pos++] = (uint8_t)b; }
};

// trace record of one op: bits 0..9 context, 10 coded / decoded bin, 11 "a context bin" (bypass and fixed-probability
// ops touch no model), 16..23 packed state before, 24..31 packed state after
__device__ __forceinline__ uint32_t trace_record(uint32_t ctx, uint32_t bin, uint32_t q_before, uint32_t q_after) {
  return ctx | (bin << 10) | (1u << 11) | (packed_from_q(q_before) << 16) | (packed_from_q(q_after) << 24);
}

__global__ void k_stream_reset(StreamDev *st, const uint8_t *init /* kStreamCtx packed states */) {
  for (int t = threadIdx.x; t < kStreamCtx; t += blockDim.x) {
    st->q_enc[t] = (uint8_t)q_from_packed(init[t]);
    st->q_dec[t] = (uint8_t)q_from_packed(init[t]);
  }
}

__global__ void k_stream_enc_start(StreamDev *st, StreamMailbox *mb) {
  enc_start(st->enc);
  st->enc_bytes = 0;
  mb->enc_bytes = 0;
}

__device__ __forceinline__ void stream_enc_op(EncCore &e, GlobalStates &cs, GlobalSink &sink, uint32_t op, uint32_t bin,
                                              uint32_t *trace) {
  uint32_t rec = 0;
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    enc_bin_ep(e, bin);
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    enc_bin_prob(e, bin, p, s_prob_rows[prob_row_index(p)]);
  } else {
    const uint32_t q0 = cs.load(op);
    enc_bin(e, bin << 1, op, cs);
    rec = trace_record(op, bin, q0, cs.load(op));
  }
  enc_flush(e, sink);
  if (trace) *trace = rec;
}

__global__ void k_stream_enc_bins(StreamDev *st, StreamMailbox *mb, const uint8_t *bins, const uint16_t *ops, uint32_t n,
                                  uint8_t *out, uint32_t *trace) {
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalStates cs{st->q_enc};
  GlobalSink sink{out, st->enc_bytes};
  for (uint32_t i = 0; i < n; i++) stream_enc_op(e, cs, sink, ops[i], bins[i] & 1u, trace ? trace + i : nullptr);
  st->enc = e;
  st->enc_bytes = sink.pos;
  mb->enc_bytes = sink.pos;
}
__global__ void k_stream_enc_bins_inline(StreamDev *st, StreamMailbox *mb, InlineOps io, uint32_t n, uint8_t *out,
                                         uint32_t *trace) {
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalStates cs{st->q_enc};
  GlobalSink sink{out, st->enc_bytes};
  for (uint32_t i = 0; i < n; i++) stream_enc_op(e, cs, sink, io.op[i], io.bin[i] & 1u, trace ? trace + i : nullptr);
  st->enc = e;
  st->enc_bytes = sink.pos;
  mb->enc_bytes = sink.pos;
}

__global__ void k_stream_enc_finish(StreamDev *st, StreamMailbox *mb, uint8_t *out) {
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalSink sink{out, st->enc_bytes};
  enc_finish(e, sink);
  st->enc = e;
  st->enc_bytes = sink.pos;
  mb->enc_bytes = sink.pos;
}

__global__ void k_stream_dec_start(StreamDev *st, const uint8_t *payload, uint32_t len) {
  if (threadIdx.x != 0) return;
  RefReader rd{payload, len, 0, 0};
  RefDec d;
  ref_dec_start(d, rd);
  st->dec = d;
  st->dec_pos = rd.nread;
  st->dec_last = rd.last;
  st->dec_len = len;
}

__device__ __forceinline__ uint32_t stream_dec_op(RefDec &d, GlobalStates &cs, RefReader &rd, uint32_t op, uint32_t *trace) {
  uint32_t bin, rec = 0;
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    bin = ref_dec_bin_ep(d, rd);
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    bin = ref_dec_bin_prob(d, p, s_prob_rows[prob_row_index(p)], rd);
  } else {
    const uint32_t q0 = cs.load(op);
    bin = ref_dec_bin(d, op, cs, rd);
    rec = trace_record(op, bin, q0, cs.load(op));
  }
  if (trace) *trace = rec;
  return bin;
}

__global__ void k_stream_dec_bins(StreamDev *st, const uint8_t *payload, const uint16_t *ops, uint32_t n, uint8_t *bins_out,
                                  uint32_t *trace) {
  if (threadIdx.x != 0) return;
  RefDec d = st->dec;
  GlobalStates cs{st->q_dec};
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  for (uint32_t i = 0; i < n; i++) bins_out[i] = (uint8_t)stream_dec_op(d, cs, rd, ops[i], trace ? trace + i : nullptr);
  st->dec = d;
  st->dec_pos = rd.nread;
  st->dec_last = rd.last;
}
__global__ void k_stream_dec_bins_inline(StreamDev *st, StreamMailbox *mb, const uint8_t *payload, InlineOps io, uint32_t n,
                                         uint32_t *trace) {
  if (threadIdx.x != 0) return;
  RefDec d = st->dec;
  GlobalStates cs{st->q_dec};
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  for (uint32_t i = 0; i < n; i++) mb->bins[i] = (uint8_t)stream_dec_op(d, cs, rd, io.op[i], trace ? trace + i : nullptr);
  st->dec = d;
  st->dec_pos = rd.nread;
  st->dec_last = rd.last;
}

__global__ void k_stream_dec_finish(StreamDev *st, StreamMailbox *mb, const uint8_t *payload) {
  if (threadIdx.x != 0) return;
  RefDec d = st->dec;
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  mb->status = ref_dec_finish(d, rd, st->dec_len);
  st->dec = d;
}

}  // namespace isscabac

using namespace isscabac;

struct isscabac_stream {
  isscabac_engine *e = nullptr;
  int nctx = 0;
  StreamDev *st = nullptr;
  uint8_t *d_init = nullptr;
  StreamMailbox *mb = nullptr;    // pinned host memory, mapped
  StreamMailbox *d_mb = nullptr;  // the same block as the device sees it
  // encoder side
  bool encoding = false, enc_done = false;
  std::vector<uint8_t> pend_bins;
  std::vector<uint16_t> pend_ops;
  uint64_t bins_total = 0;  // ops already on the device
  uint8_t *d_out = nullptr;
  uint64_t out_cap = 0;
  uint8_t *d_io = nullptr;  // staging for bins / ops / decoded bins
  uint64_t io_cap = 0;
  // decoder side
  bool decoding = false;
  uint8_t *d_payload = nullptr;
  uint64_t payload_cap = 0, payload_len = 0;
  uint64_t dec_total = 0;  // ops decoded so far
  // state traces (off by default): one 4-byte record per op and side
  bool tracing = false;
  uint8_t *d_trace[2] = {nullptr, nullptr};
  uint64_t trace_cap[2] = {0, 0};
};

static int stream_grow(isscabac_engine *e, uint8_t **p, uint64_t *cap, uint64_t need, uint64_t keep) {
  if (*cap >= need) return ISSCABAC_OK;
  uint64_t ncap = *cap ? *cap : 4096;
  while (ncap < need) ncap *= 2;
  uint8_t *np = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&np), ncap));
  if (*p) {
    if (keep) HIP_TRY(e, hipMemcpyAsync(np, *p, keep, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipFree(*p));
  }
  *p = np;
  *cap = ncap;
  return ISSCABAC_OK;
}

// room for the records of ops [have, have + n) of one side; nullptr when tracing is off
static int trace_room(isscabac_stream *s, int side, uint64_t have, uint64_t n, uint32_t **at) {
  *at = nullptr;
  if (!s->tracing) return ISSCABAC_OK;
  const int rc = stream_grow(s->e, &s->d_trace[side], &s->trace_cap[side], 4 * (have + n), 4 * have);
  if (rc) return rc;
  *at = reinterpret_cast<uint32_t *>(s->d_trace[side]) + have;
  return ISSCABAC_OK;
}

// code the queued ops on the device; leaves the stream synchronised and the mailbox current
static int stream_flush_pending(isscabac_stream *s) {
  isscabac_engine *e = s->e;
  const uint64_t n = s->pend_bins.size();
  if (n == 0) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return ISSCABAC_OK;
  }
  int rc;
  // the mailbox holds the byte count of everything coded so far (the stream is synchronised between calls)
  if ((rc = stream_grow(e, &s->d_out, &s->out_cap, isscabac_stream_bound((uint32_t)std::min<uint64_t>(s->bins_total + n, 0xFFFFFFF0ull)) + 16,
                        s->mb->enc_bytes)))
    return rc;
  uint32_t *tr;
  if ((rc = trace_room(s, 0, s->bins_total, n, &tr))) return rc;
  if (n <= kInlineOps) {
    InlineOps io;
    memcpy(io.op, s->pend_ops.data(), n * sizeof(uint16_t));
    memcpy(io.bin, s->pend_bins.data(), n);
    hipLaunchKernelGGL(k_stream_enc_bins_inline, dim3(1), dim3(64), 0, e->stream, s->st, s->d_mb, io, (uint32_t)n, s->d_out, tr);
  } else {
    if ((rc = stream_grow(e, &s->d_io, &s->io_cap, 3 * n + 2, 0))) return rc;
    uint8_t *d_ops = s->d_io + ((n + 1) & ~1ull);
    HIP_TRY(e, hipMemcpyAsync(s->d_io, s->pend_bins.data(), n, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(d_ops, s->pend_ops.data(), 2 * n, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_stream_enc_bins, dim3(1), dim3(64), 0, e->stream, s->st, s->d_mb, s->d_io,
                       reinterpret_cast<const uint16_t *>(d_ops), (uint32_t)n, s->d_out, tr);
  }
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  s->bins_total += n;
  s->pend_bins.clear();
  s->pend_ops.clear();
  return ISSCABAC_OK;
}

// one op of the u8 API as a 16-bit op; false = out of range
static bool op_from_u8(const isscabac_stream *s, uint8_t c, uint16_t *op) {
  if (c == ISSCABAC_BYPASS) {
    *op = ISSCABAC_OP_BYPASS;
  } else if (s->nctx <= ISSCABAC_PROB_MAX_CTX && c > ISSCABAC_PROB_BASE) {  // a fixed-probability bin, as in the batch calls
    *op = (uint16_t)ISSCABAC_OP_PROB(c - ISSCABAC_PROB_BASE);
  } else {
    *op = c;
  }
  return true;
}
static const char *op_problem(const isscabac_stream *s, uint16_t op) {
  if (op == ISSCABAC_OP_BYPASS) return nullptr;
  if (op & 0x8000u) {
    const unsigned p = op & 0x7FFFu;
    if (p < 1 || p > 99) return "Error: fixed probability must be 1..99 percent";
    if (p >= 46 && p <= 54 && p != 50)
      return "Error: fixed probabilities 46..49 and 51..54 are not supported (the reference's range register leaves 256..510 there)";
    return nullptr;
  }
  return op < (unsigned)s->nctx ? nullptr : "Error: invalid context index";
}

extern "C" {

int isscabac_stream_open(isscabac_engine *e, const uint8_t *ctx_init, int nctx, isscabac_stream **out) {
  if (!e || !out) return ISSCABAC_ERR_ARG;
  *out = nullptr;
  if (!ctx_init || nctx < 1 || nctx > ISSCABAC_STREAM_MAX_CTX)
    return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_stream_open: nctx must be 1..%d", ISSCABAC_STREAM_MAX_CTX);
  uint8_t init[kStreamCtx];
  for (int c = 0; c < kStreamCtx; c++) init[c] = 1;  // ContextModel() default (mps 1, state 0), ref: ContextModel.cpp:54
  for (int c = 0; c < nctx; c++) {
    if ((ctx_init[c] >> 1) > 62) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_stream_open: ctx_init[%d] has state 63", c);
    init[c] = ctx_init[c];
  }
  HIP_TRY(e, hipSetDevice(e->device));
  isscabac_stream *s = new isscabac_stream;
  s->e = e;
  s->nctx = nctx;
  hipError_t err;
  if ((err = hipMalloc(reinterpret_cast<void **>(&s->st), sizeof(StreamDev))) != hipSuccess ||
      (err = hipMalloc(reinterpret_cast<void **>(&s->d_init), kStreamCtx)) != hipSuccess ||
      (err = hipHostMalloc(reinterpret_cast<void **>(&s->mb), sizeof(StreamMailbox), hipHostMallocMapped)) != hipSuccess ||
      (err = hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_mb), s->mb, 0)) != hipSuccess ||
      (err = hipMemcpyAsync(s->d_init, init, kStreamCtx, hipMemcpyHostToDevice, e->stream)) != hipSuccess) {
    isscabac_stream_close(s);
    return engine_fail(e, ISSCABAC_ERR_HIP, "isscabac_stream_open: %s", hipGetErrorString(err));
  }
  memset(s->mb, 0, sizeof(StreamMailbox));
  hipLaunchKernelGGL(k_stream_reset, dim3(1), dim3(256), 0, e->stream, s->st, s->d_init);
  if ((err = hipStreamSynchronize(e->stream)) != hipSuccess) {
    isscabac_stream_close(s);
    return engine_fail(e, ISSCABAC_ERR_HIP, "isscabac_stream_open: %s", hipGetErrorString(err));
  }
  *out = s;
  return ISSCABAC_OK;
}

void isscabac_stream_close(isscabac_stream *s) {
  if (!s) return;
  (void)hipSetDevice(s->e->device);
  (void)hipStreamSynchronize(s->e->stream);
  for (void *p : {(void *)s->st, (void *)s->d_init, (void *)s->d_out, (void *)s->d_io, (void *)s->d_payload,
                  (void *)s->d_trace[0], (void *)s->d_trace[1]})
    if (p) (void)hipFree(p);
  if (s->mb) (void)hipHostFree(s->mb);
  delete s;
}

int isscabac_stream_set_trace(isscabac_stream *s, int enabled) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (s->encoding || s->decoding)
    return engine_fail(s->e, ISSCABAC_ERR_STATE, "isscabac_stream_set_trace: switch tracing before encodeStart / decodeStart");
  s->tracing = enabled != 0;
  return ISSCABAC_OK;
}

int isscabac_stream_encode_start(isscabac_stream *s) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (s->encoding) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeStart: the bitstream is already open for writing");  // ref asserts, CABAC_BitstreamFile.cpp:52
  HIP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_stream_enc_start, dim3(1), dim3(1), 0, e->stream, s->st, s->d_mb);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  s->encoding = true;
  s->enc_done = false;
  s->bins_total = 0;
  s->pend_bins.clear();
  s->pend_ops.clear();
  return ISSCABAC_OK;
}

int isscabac_stream_encode_ops(isscabac_stream *s, const uint8_t *bins, const uint16_t *ops, uint32_t n) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->encoding) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeBin before encodeStart");
  if (n && (!bins || !ops)) return engine_fail(e, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++) {
    if (bins[i] > 1) return engine_fail(e, ISSCABAC_ERR_ARG, "Error: invalid input 3, bin to be encoded should either be 1 or 0");  // ref: :227
    if (const char *why = op_problem(s, ops[i])) return engine_fail(e, ISSCABAC_ERR_ARG, "%s (op 0x%04X, %d contexts)", why, ops[i], s->nctx);
  }
  s->pend_bins.insert(s->pend_bins.end(), bins, bins + n);
  s->pend_ops.insert(s->pend_ops.end(), ops, ops + n);
  if (s->pend_bins.size() >= (1u << 16)) {
    HIP_TRY(e, hipSetDevice(e->device));
    return stream_flush_pending(s);
  }
  return ISSCABAC_OK;
}

int isscabac_stream_encode_bins(isscabac_stream *s, const uint8_t *bins, const uint8_t *ctx, uint32_t n) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (n && !ctx) return engine_fail(s->e, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  std::vector<uint16_t> ops(n);
  for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &ops[i]);
  return isscabac_stream_encode_ops(s, bins, ops.data(), n);
}

int isscabac_stream_num_bits(isscabac_stream *s, uint64_t *bits) {
  if (!s || !bits) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  HIP_TRY(e, hipSetDevice(e->device));
  const int rc = stream_flush_pending(s);
  if (rc) return rc;
  *bits = 8ull * s->mb->enc_bytes;  // CABAC_BitstreamFile::getNumberOfWrittenBits: no held bits before finish()
  return ISSCABAC_OK;
}

int isscabac_stream_encode_finish(isscabac_stream *s, uint8_t *out, uint64_t cap, uint64_t *nbytes) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->encoding && !s->enc_done) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeFinish before encodeStart");
  HIP_TRY(e, hipSetDevice(e->device));
  if (!s->enc_done) {
    int rc = stream_flush_pending(s);
    if (rc) return rc;
    if ((rc = stream_grow(e, &s->d_out, &s->out_cap, isscabac_stream_bound((uint32_t)std::min<uint64_t>(s->bins_total, 0xFFFFFFF0ull)) + 16,
                          s->mb->enc_bytes)))
      return rc;
    hipLaunchKernelGGL(k_stream_enc_finish, dim3(1), dim3(64), 0, e->stream, s->st, s->d_mb, s->d_out);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    s->enc_done = true;
    s->encoding = false;
  }
  const uint32_t total = s->mb->enc_bytes;
  if (nbytes) *nbytes = total;
  if (total > cap || !out) return engine_fail(e, ISSCABAC_ERR_CAPACITY, "encodeFinish: bitstream has %u bytes, buffer holds %llu", total, (unsigned long long)cap);
  HIP_TRY(e, hipMemcpyAsync(out, s->d_out, total, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_stream_decode_start(isscabac_stream *s, const uint8_t *payload, uint64_t nbytes) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!payload && nbytes) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeStart: null payload");
  if (nbytes > 0xFFFFFFF0ull) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeStart: payload too large");
  HIP_TRY(e, hipSetDevice(e->device));
  int rc = stream_grow(e, &s->d_payload, &s->payload_cap, nbytes + 16, 0);
  if (rc) return rc;
  if (nbytes) HIP_TRY(e, hipMemcpyAsync(s->d_payload, payload, nbytes, hipMemcpyHostToDevice, e->stream));
  s->payload_len = nbytes;
  hipLaunchKernelGGL(k_stream_dec_start, dim3(1), dim3(64), 0, e->stream, s->st, s->d_payload, (uint32_t)nbytes);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  s->decoding = true;
  s->dec_total = 0;
  return ISSCABAC_OK;
}

int isscabac_stream_decode_ops(isscabac_stream *s, const uint16_t *ops, uint32_t n, uint8_t *bins_out) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeBin before decodeStart");
  if (n == 0) return ISSCABAC_OK;
  if (!ops || !bins_out) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++)
    if (const char *why = op_problem(s, ops[i])) return engine_fail(e, ISSCABAC_ERR_ARG, "%s (op 0x%04X, %d contexts)", why, ops[i], s->nctx);
  HIP_TRY(e, hipSetDevice(e->device));
  uint32_t *tr;
  int rc = trace_room(s, 1, s->dec_total, n, &tr);
  if (rc) return rc;
  if (n <= kInlineOps) {
    InlineOps io;
    memcpy(io.op, ops, n * sizeof(uint16_t));
    hipLaunchKernelGGL(k_stream_dec_bins_inline, dim3(1), dim3(64), 0, e->stream, s->st, s->d_mb, s->d_payload, io, n, tr);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    memcpy(bins_out, s->mb->bins, n);
  } else {
    if ((rc = stream_grow(e, &s->d_io, &s->io_cap, 3ull * n, 0))) return rc;
    HIP_TRY(e, hipMemcpyAsync(s->d_io, ops, 2ull * n, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_stream_dec_bins, dim3(1), dim3(64), 0, e->stream, s->st, s->d_payload,
                       reinterpret_cast<const uint16_t *>(s->d_io), n, s->d_io + 2ull * n, tr);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(bins_out, s->d_io + 2ull * n, n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  s->dec_total += n;
  return ISSCABAC_OK;
}

int isscabac_stream_decode_bins(isscabac_stream *s, const uint8_t *ctx, uint32_t n, uint8_t *bins_out) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (n && !ctx) return engine_fail(s->e, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  std::vector<uint16_t> ops(n);
  for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &ops[i]);
  return isscabac_stream_decode_ops(s, ops.data(), n, bins_out);
}

int isscabac_stream_decode_finish(isscabac_stream *s, uint32_t *status) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeFinish before decodeStart");
  HIP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_stream_dec_finish, dim3(1), dim3(64), 0, e->stream, s->st, s->d_mb, s->d_payload);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (status) *status = s->mb->status;
  s->decoding = false;
  return ISSCABAC_OK;
}

// ---- getEncoderStats / getDecoderStats ------------------------------------------------------------------------------------
// ContextModel::addCabacStep's state numbering, ref: ContextModel.cpp:128-129
static inline uint8_t trace_state(uint32_t packed) { return (packed & 1u) == 0 ? (uint8_t)(63u - (packed >> 1)) : (uint8_t)((packed >> 1) + 64u); }

int isscabac_trace_stats(const uint32_t *records, uint64_t nrecords, int ctx, int decoder, uint8_t *trace5, uint64_t cap_steps,
                         uint64_t *nsteps, uint32_t *transitions) {
  if ((!records && nrecords) || ctx < 0 || ctx >= kStreamCtx) return ISSCABAC_ERR_ARG;
  if (transitions) memset(transitions, 0, 128 * 128 * sizeof(uint32_t));
  uint64_t steps = 0;
  for (uint64_t i = 0; i < nrecords; i++) {
    const uint32_t r = records[i];
    if (!(r & (1u << 11)) || (int)(r & 0x3FFu) != ctx) continue;
    const uint32_t pb = (r >> 16) & 0xFFu, pa = r >> 24;
    const uint8_t tp = trace_state(pb), ta = trace_state(pa);
    if (trace5 && steps < cap_steps) {
      uint8_t *t = trace5 + 5 * steps;
      // the reference's gateway stores the bin variable BEFORE decodeBin has written it: always 0 on the decoder side
      // (ref: SimpleCABACMex.cpp:316-327); kept, it is what a caller of getDecoderStats sees
      t[0] = decoder ? 0 : (uint8_t)((r >> 10) & 1u);
      t[1] = tp;
      t[2] = (uint8_t)(pb & 1u);
      t[3] = ta;
      t[4] = (uint8_t)(pa & 1u);
    }
    if (transitions) transitions[(size_t)tp * 128 + ta]++;
    steps++;
  }
  if (nsteps) *nsteps = steps;
  return ISSCABAC_OK;
}

int isscabac_stream_get_stats(isscabac_stream *s, int decoder, int ctx, uint8_t *trace5, uint64_t cap_steps, uint64_t *nsteps,
                              uint32_t *transitions) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->tracing) return engine_fail(e, ISSCABAC_ERR_STATE, "getEncoderStats: tracing is off (isscabac_stream_set_trace)");
  if (ctx < 0 || ctx >= s->nctx) return engine_fail(e, ISSCABAC_ERR_ARG, "Error: invalid context index");
  HIP_TRY(e, hipSetDevice(e->device));
  const int side = decoder ? 1 : 0;
  if (!decoder && s->encoding) {
    const int rc = stream_flush_pending(s);
    if (rc) return rc;
  }
  const uint64_t n = decoder ? s->dec_total : s->bins_total;
  std::vector<uint32_t> rec(n);
  if (n) {
    HIP_TRY(e, hipMemcpyAsync(rec.data(), s->d_trace[side], 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  return isscabac_trace_stats(rec.data(), n, ctx, decoder, trace5, cap_steps, nsteps, transitions);
}

}  // extern "C"
