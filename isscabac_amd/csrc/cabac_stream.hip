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
// Calls that need an answer at once -- decodeBin, and getNumBits behind a few queued encodeBin -- are served by a LIVE
// kernel: one lane that holds the coder in registers and polls a request word in the pinned mailbox, so that such a call is
// one write over the link and one word back instead of a kernel launch and a stream synchronisation (10 -> 3-4 us).  It leaves
// by itself after 200 us without requests and is started again by the next call; everything else (start, finish, long
// calls, traces) stops it first and runs as ordinary launches on the engine's stream.
//
// With tracing on, every op leaves a 4-byte record on the device: getEncoderStats / getDecoderStats of the reference's
// Windows build (ref: SimpleCABACMex.cpp:356-466, ContextModel.cpp:126-134) are computed from those records.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
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
  // the live kernel's words (coherent host memory: every access crosses the link)
  uint64_t req;      // host -> device: seq (bits 0..31) | command << 32 | n << 36 | bin[0] << 44 | op[0] << 48
  uint64_t done;     // device -> host: seq | answer << 32 (decode, n = 1: the bin; encode: bytes written so far)
  uint32_t exited;   // device -> host: the kernel's last word
  uint32_t pad;
  uint16_t op[32];   // request payload of a call with n > 1
  uint8_t bin[32];
};
constexpr uint32_t kLiveDec = 1, kLiveEnc = 2, kLiveQuit = 3, kLiveMaxOps = 32;

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
// The same with the handle's 1 000 states and the tables staged in the LDS for the length of a kernel (ops in bulk and the live
// kernel): a bin is LDS round trips instead of three dependent trips to device memory (0.46 -> 0.1-0.2 us per queued bin)
struct StagedStates {
  uint8_t *q;
  const uint32_t *rows, *next, *prob;
  __device__ uint32_t load(uint32_t c) const { return q[c]; }
  __device__ void store(uint32_t c, uint32_t v) const { q[c] = (uint8_t)v; }
  __device__ uint32_t addr(uint32_t c) const { return c; }
  __device__ uint32_t load_at(uint32_t a) const { return q[a]; }
  __device__ void store_at(uint32_t a, uint32_t v) const { q[a] = (uint8_t)v; }
  __device__ uint32_t lps_row(uint32_t v) const { return rows[v >> 2]; }
  __device__ uint32_t next_entry(uint32_t v) const { return next[v >> 2]; }
  __device__ uint32_t prob_row(uint32_t p) const { return prob[prob_row_index(p)]; }
};
struct StageLds {
  uint32_t rows[64], next[64], prob[52];
  uint8_t q[1024];
};
// all threads of the workgroup: stage one side's states + the tables / write the states back
__device__ __forceinline__ StagedStates stage_in(StageLds &l, const uint8_t *q_dev) {
  for (int t = threadIdx.x; t < 64; t += blockDim.x) {
    l.rows[t] = s_lps_rows[t];
    l.next[t] = s_next.e[t];
    if (t < 49) l.prob[t] = s_prob_rows[t];
  }
  for (int t = threadIdx.x; t < kStreamCtx; t += blockDim.x) l.q[t] = q_dev[t];
  __syncthreads();
  return StagedStates{l.q, l.rows, l.next, l.prob};
}
__device__ __forceinline__ void stage_out(const StageLds &l, uint8_t *q_dev) {
  __syncthreads();
  for (int t = threadIdx.x; t < kStreamCtx; t += blockDim.x) q_dev[t] = l.q[t];
}

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

__device__ __forceinline__ uint32_t prob_row_of(const GlobalStates &, uint32_t p) { return s_prob_rows[prob_row_index(p)]; }
__device__ __forceinline__ uint32_t prob_row_of(const StagedStates &cs, uint32_t p) { return cs.prob_row(p); }

template <class States>
__device__ __forceinline__ void stream_enc_op(EncCore &e, States &cs, GlobalSink &sink, uint32_t op, uint32_t bin,
                                              uint32_t *trace) {
  uint32_t rec = 0;
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    enc_bin_ep(e, bin);
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    enc_bin_prob(e, bin, p, prob_row_of(cs, p));
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
  __shared__ StageLds lds;
  StagedStates cs = stage_in(lds, st->q_enc);
  if (threadIdx.x == 0) {
    EncCore e = st->enc;
    GlobalSink sink{out, st->enc_bytes};
    // eight ops at a time, the next eight requested before these are coded (a lone lane waits out every load otherwise)
    uint32_t o[8], b[8], no[8], nb[8];
    for (int j = 0; j < 8; j++) {
      const uint32_t idx = min((uint32_t)j, n - 1);
      o[j] = ops[idx];
      b[j] = bins[idx];
    }
    for (uint32_t i = 0; i < n; i += 8) {
      for (int j = 0; j < 8; j++) {
        const uint32_t idx = min(i + 8 + j, n - 1);
        no[j] = ops[idx];
        nb[j] = bins[idx];
      }
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (i + j < n) stream_enc_op(e, cs, sink, o[j], b[j] & 1u, trace ? trace + i + j : nullptr);
      for (int j = 0; j < 8; j++) {
        o[j] = no[j];
        b[j] = nb[j];
      }
    }
    st->enc = e;
    st->enc_bytes = sink.pos;
    mb->enc_bytes = sink.pos;
  }
  stage_out(lds, st->q_enc);
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

template <class States>
__device__ __forceinline__ uint32_t stream_dec_op(RefDec &d, States &cs, RefReader &rd, uint32_t op, uint32_t *trace) {
  uint32_t bin, rec = 0;
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    bin = ref_dec_bin_ep(d, rd);
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    bin = ref_dec_bin_prob(d, p, prob_row_of(cs, p), rd);
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
  __shared__ StageLds lds;
  StagedStates cs = stage_in(lds, st->q_dec);
  if (threadIdx.x == 0) {
    RefDec d = st->dec;
    RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
    uint32_t o[8], no[8];
    for (int j = 0; j < 8; j++) o[j] = ops[min((uint32_t)j, n - 1)];
    for (uint32_t i = 0; i < n; i += 8) {
      for (int j = 0; j < 8; j++) no[j] = ops[min(i + 8 + j, n - 1)];
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (i + j < n) bins_out[i + j] = (uint8_t)stream_dec_op(d, cs, rd, o[j], trace ? trace + i + j : nullptr);
      for (int j = 0; j < 8; j++) o[j] = no[j];
    }
    st->dec = d;
    st->dec_pos = rd.nread;
    st->dec_last = rd.last;
  }
  stage_out(lds, st->q_dec);
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

// The live kernel.  `last` = the sequence number of the last request served before it started.  Leaving: it writes the coder
// back, then `exited`; a request that arrives while it leaves is either served on its last look at the request word or left
// for the next kernel -- the host tells the two apart by `done` (cabac_stream.hip: live_call).
__global__ void k_stream_live(StreamDev *st, StreamMailbox *mb, const uint8_t *payload, uint8_t *out, uint32_t *tr_enc,
                              uint32_t *tr_dec /* the next free trace record of either side, or null */, uint32_t last,
                              uint64_t idle_ticks, uint64_t max_ticks) {
  __shared__ StageLds lds_e, lds_d;  // (one wave: the other lanes only help to stage the states in)
  StagedStates ce = stage_in(lds_e, st->q_enc), cd = stage_in(lds_d, st->q_dec);
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalSink sink{out, st->enc_bytes};
  RefDec d = st->dec;
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  // Two clocks: it leaves after idle_ticks without a request, and after max_ticks of residency whatever the traffic -- a caller
  // that keeps it busy must not keep it in front of a foreign kernel (another engine's, another HIP user's: the engine's own
  // work stops it first) that the runtime mapped onto the same hardware queue.  The next call starts the next kernel.
  const uint64_t t_in = wall_clock64();
  uint64_t t0 = t_in;
  bool leaving = false;
  for (uint32_t it = 0; it < (1u << 24); it++) {  // (the count is a backstop for the clocks)
    const uint64_t r = __hip_atomic_load(&mb->req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t seq = (uint32_t)r;
    if (seq != last) {
      const uint32_t cmd = (uint32_t)(r >> 32) & 0xFu, n = (uint32_t)(r >> 36) & 0xFFu;
      uint64_t reply = seq;
      if (cmd == kLiveDec) {
        if (n == 1) {
          reply |= (uint64_t)stream_dec_op(d, cd, rd, (uint32_t)(r >> 48), tr_dec ? tr_dec++ : nullptr) << 32;
        } else {
          for (uint32_t i = 0; i < n; i++) {
            const uint32_t op = *reinterpret_cast<volatile uint16_t *>(&mb->op[i]);
            *reinterpret_cast<volatile uint8_t *>(&mb->bins[i]) = (uint8_t)stream_dec_op(d, cd, rd, op, tr_dec ? tr_dec++ : nullptr);
          }
        }
      } else if (cmd == kLiveEnc) {
        if (n == 1) {
          stream_enc_op(e, ce, sink, (uint32_t)(r >> 48), (uint32_t)(r >> 44) & 1u, tr_enc ? tr_enc++ : nullptr);
        } else {
          for (uint32_t i = 0; i < n; i++) {
            const uint32_t op = *reinterpret_cast<volatile uint16_t *>(&mb->op[i]);
            const uint32_t bin = *reinterpret_cast<volatile uint8_t *>(&mb->bin[i]);
            stream_enc_op(e, ce, sink, op, bin & 1u, tr_enc ? tr_enc++ : nullptr);
          }
        }
        reply |= (uint64_t)sink.pos << 32;
      } else {
        leaving = true;
      }
      __threadfence_system();
      __hip_atomic_store(&mb->done, reply, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      last = seq;
      if (leaving) break;
      t0 = wall_clock64();
      if (t0 - t_in > max_ticks) break;  // (the answer is out: the host finds `exited` and starts the next kernel with its next call)
      continue;
    }
    if (leaving) break;  // looked once more after deciding to leave: nothing came
    if (wall_clock64() - t0 > idle_ticks) leaving = true;
  }
  for (int t = 0; t < kStreamCtx / 4; t++) {
    reinterpret_cast<uint32_t *>(st->q_enc)[t] = reinterpret_cast<const uint32_t *>(lds_e.q)[t];
    reinterpret_cast<uint32_t *>(st->q_dec)[t] = reinterpret_cast<const uint32_t *>(lds_d.q)[t];
  }
  st->enc = e;
  st->enc_bytes = sink.pos;
  st->dec = d;
  st->dec_pos = rd.nread;
  st->dec_last = rd.last;
  __threadfence_system();
  __hip_atomic_store(&mb->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace isscabac

using namespace isscabac;

struct isscabac_stream {
  isscabac_engine *e = nullptr;  // null once its engine has been destroyed: every call but close then returns ISSCABAC_ERR_STATE
  int device = 0;
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
  // the live kernel (see the top of this file)
  bool live_on = true;        // ISSCABAC_STREAM_LIVE=0 switches it off
  bool live = false;          // a live kernel has been started and not yet been seen to leave
  hipStream_t live_stream = nullptr;
  uint32_t live_seq = 0;
  uint64_t idle_ticks = 20000;  // 200 us of wall_clock64() at 100 MHz (set from the device's clock rate at open)
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

// ---- the live kernel's host side ---------------------------------------------------------------------------------------------
static inline uint64_t mb_load64(const uint64_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline uint32_t mb_load32(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

constexpr uint64_t kLiveMaxIdles = 10;  // a live kernel's residency bound in idle times: 2 ms
static int quiesce_others(isscabac_stream *s);
static int live_launch(isscabac_stream *s) {
  isscabac_engine *e = s->e;
  __atomic_store_n(&s->mb->exited, 0u, __ATOMIC_RELEASE);
  uint32_t *tr_enc = s->tracing && s->d_trace[0] ? reinterpret_cast<uint32_t *>(s->d_trace[0]) + s->bins_total : nullptr;
  uint32_t *tr_dec = s->tracing && s->d_trace[1] ? reinterpret_cast<uint32_t *>(s->d_trace[1]) + s->dec_total : nullptr;
  hipLaunchKernelGGL(k_stream_live, dim3(1), dim3(64), 0, s->live_stream, s->st, s->d_mb, s->d_payload, s->d_out, tr_enc, tr_dec,
                     (uint32_t)mb_load64(&s->mb->done), s->idle_ticks, kLiveMaxIdles * s->idle_ticks);
  HIP_TRY(e, hipGetLastError());
  s->live = true;
  return ISSCABAC_OK;
}

// post one request and wait for its answer (the whole `done` word); starts the kernel when there is none
static int live_call(isscabac_stream *s, uint32_t cmd, uint32_t n, const uint16_t *ops, const uint8_t *bins, uint64_t *answer) {
  isscabac_engine *e = s->e;
  StreamMailbox *mb = s->mb;
  if (n > 1) {
    memcpy(mb->op, ops, n * sizeof(uint16_t));
    if (bins) memcpy(mb->bin, bins, n);
  }
  const uint32_t seq = ++s->live_seq;
  const uint64_t word = (uint64_t)seq | (uint64_t)cmd << 32 | (uint64_t)n << 36 | (uint64_t)(bins && n ? bins[0] & 1u : 0u) << 44 |
                        (uint64_t)(ops && n ? ops[0] : 0u) << 48;
  __atomic_store_n(&mb->req, word, __ATOMIC_RELEASE);
  if (s->live && mb_load32(&mb->exited)) s->live = false;  // it left since the last call (its write-back is behind that word)
  int rc;
  if (!s->live && ((rc = quiesce_others(s)) || (rc = live_launch(s)))) return rc;
  std::chrono::steady_clock::time_point t_start;
  for (uint64_t spins = 0;; spins++) {
    uint64_t d = mb_load64(&mb->done);
    if ((uint32_t)d == seq) {
      *answer = d;
      return ISSCABAC_OK;
    }
    if (mb_load32(&mb->exited)) {
      // the kernel left; did it serve this request on its way out?
      d = mb_load64(&mb->done);
      if ((uint32_t)d == seq) {
        *answer = d;
        s->live = false;
        return ISSCABAC_OK;
      }
      if ((rc = quiesce_others(s)) || (rc = live_launch(s))) return rc;  // no: the next kernel finds the request word waiting
    }
    if ((spins & 0xFFFFu) == 0xFFFFu) {
      // (a kernel that has not answered for seconds: report it instead of spinning for ever)
      const auto t_now = std::chrono::steady_clock::now();
      if (spins == 0xFFFFu) t_start = t_now;
      else if (std::chrono::duration<double>(t_now - t_start).count() > 5.0)
        return engine_fail(e, ISSCABAC_ERR_HIP, "per-bin handle: the live kernel does not answer");
    }
  }
}

// make the live kernel leave and wait until it has: afterwards the coder state is in device memory again
static int live_stop_self(isscabac_stream *s);
// ... and, since every caller goes on to launch on the engine's stream, the other handles' live kernels too
static int live_stop(isscabac_stream *s) {
  const int rc = live_stop_self(s);
  if (rc) return rc;
  if (s->e->handles.size() > 1)
    for (isscabac_stream *h : s->e->handles)
      if (h != s && h->live) {
        const int rc2 = live_stop_self(h);
        if (rc2) return rc2;
      }
  return ISSCABAC_OK;
}
static int live_stop_self(isscabac_stream *s) {
  if (!s->live) return ISSCABAC_OK;
  isscabac_engine *e = s->e;
  if (!mb_load32(&s->mb->exited)) {
    uint64_t a;
    const int rc = live_call(s, kLiveQuit, 0, nullptr, nullptr, &a);
    if (rc) return rc;
  }
  HIP_TRY(e, hipStreamSynchronize(s->live_stream));
  s->live = false;
  return ISSCABAC_OK;
}

// A polling kernel must never stand in front of another kernel: the runtime maps streams onto a few hardware queues, and a
// kernel queued behind a live one waits until that one leaves by itself (measured: four handles decoding in turn 500 us per
// call, a batch call right behind a decodeBin 1 ms; tools/live_interleave.py).  So whatever is enqueued for an engine -- a
// batch call, another handle's kernels -- first makes the live kernels of the engine's handles leave.
static int quiesce_others(isscabac_stream *s) {
  return s->e->handles.size() > 1 ? engine_quiesce_live(s->e, s) : ISSCABAC_OK;
}

// code the queued ops on the device; leaves the stream synchronised and the mailbox current
static int stream_flush_pending(isscabac_stream *s) {
  isscabac_engine *e = s->e;
  const uint64_t n = s->pend_bins.size();
  if (n && n <= kLiveMaxOps && s->live_on) {
    // room for the bytes and the trace records first: the live kernel holds the buffers' addresses, so it leaves before they move
    const uint64_t need_out = isscabac_stream_bound((uint32_t)std::min<uint64_t>(s->bins_total + n, 0xFFFFFFF0ull)) + 16;
    int rc;
    if (s->out_cap < need_out || (s->tracing && s->trace_cap[0] < 4 * (s->bins_total + n))) {
      uint32_t *tr;
      if ((rc = live_stop(s)) || (rc = stream_grow(e, &s->d_out, &s->out_cap, need_out, s->mb->enc_bytes)) ||
          (rc = trace_room(s, 0, s->bins_total, n, &tr)))
        return rc;
    }
    uint64_t a;
    if ((rc = live_call(s, kLiveEnc, (uint32_t)n, s->pend_ops.data(), s->pend_bins.data(), &a))) return rc;
    s->mb->enc_bytes = (uint32_t)(a >> 32);
    s->bins_total += n;
    s->pend_bins.clear();
    s->pend_ops.clear();
    return ISSCABAC_OK;
  }
  if (n == 0) {  // (the mailbox's byte count is current either way)
    if (!s->live) HIP_TRY(e, hipStreamSynchronize(e->stream));
    return ISSCABAC_OK;
  }
  int rc;
  if ((rc = live_stop(s))) return rc;
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

namespace isscabac {
// isscabac_destroy: stop the live kernels and cut the handles loose -- they stay valid to close, nothing else
void engine_orphan_handles(isscabac_engine *e) {
  (void)engine_quiesce_live(e, nullptr);
  for (isscabac_stream *h : e->handles) {
    if (h->live_stream) (void)hipStreamSynchronize(h->live_stream);
    h->e = nullptr;
  }
  e->handles.clear();
}
int engine_quiesce_live(isscabac_engine *e, isscabac_stream *except) {
  for (isscabac_stream *h : e->handles)
    if (h != except && h->live) {
      const int rc = live_stop_self(h);
      if (rc) return rc;
    }
  return ISSCABAC_OK;
}
}  // namespace isscabac

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
  ENGINE_ENTER(e);  // (the reset kernel below goes into the engine's stream: no live kernel of an older handle may stand in front of it)
  isscabac_stream *s = new isscabac_stream;
  s->e = e;
  s->device = e->device;
  s->nctx = nctx;
  hipError_t err;
  if ((err = hipMalloc(reinterpret_cast<void **>(&s->st), sizeof(StreamDev))) != hipSuccess ||
      (err = hipMalloc(reinterpret_cast<void **>(&s->d_init), kStreamCtx)) != hipSuccess ||
      (err = hipHostMalloc(reinterpret_cast<void **>(&s->mb), sizeof(StreamMailbox), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
      (err = hipStreamCreateWithFlags(&s->live_stream, hipStreamNonBlocking)) != hipSuccess ||
      (err = hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_mb), s->mb, 0)) != hipSuccess ||
      (err = hipMemcpyAsync(s->d_init, init, kStreamCtx, hipMemcpyHostToDevice, e->stream)) != hipSuccess) {
    isscabac_stream_close(s);
    return engine_fail(e, ISSCABAC_ERR_HIP, "isscabac_stream_open: %s", hipGetErrorString(err));
  }
  memset(s->mb, 0, sizeof(StreamMailbox));
  {
    const char *env = getenv("ISSCABAC_STREAM_LIVE");
    s->live_on = !(env && env[0] == '0');
    int khz = 0;  // wall_clock64() ticks per millisecond
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device) == hipSuccess && khz > 0) s->idle_ticks = (uint64_t)khz / 5;  // 200 us
    else (void)hipGetLastError();
  }
  hipLaunchKernelGGL(k_stream_reset, dim3(1), dim3(256), 0, e->stream, s->st, s->d_init);
  if ((err = hipStreamSynchronize(e->stream)) != hipSuccess) {
    isscabac_stream_close(s);
    return engine_fail(e, ISSCABAC_ERR_HIP, "isscabac_stream_open: %s", hipGetErrorString(err));
  }
  e->handles.push_back(s);
  *out = s;
  return ISSCABAC_OK;
}

void isscabac_stream_close(isscabac_stream *s) {
  if (!s) return;
  if (s->e) {  // (null: the engine was destroyed first, isscabac_destroy stopped the live kernel and left the handle orphaned)
    (void)hipSetDevice(s->e->device);
    (void)live_stop(s);
    auto &hs = s->e->handles;
    hs.erase(std::remove(hs.begin(), hs.end(), s), hs.end());
    (void)hipStreamSynchronize(s->e->stream);
  } else {
    (void)hipSetDevice(s->device);
  }
  if (s->live_stream) (void)hipStreamDestroy(s->live_stream);
  for (void *p : {(void *)s->st, (void *)s->d_init, (void *)s->d_out, (void *)s->d_io, (void *)s->d_payload,
                  (void *)s->d_trace[0], (void *)s->d_trace[1]})
    if (p) (void)hipFree(p);
  if (s->mb) (void)hipHostFree(s->mb);
  delete s;
}

int isscabac_stream_set_trace(isscabac_stream *s, int enabled) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  if (s->encoding || s->decoding)
    return engine_fail(s->e, ISSCABAC_ERR_STATE, "isscabac_stream_set_trace: switch tracing before encodeStart / decodeStart");
  if (const int rc = live_stop(s)) return rc;
  s->tracing = enabled != 0;
  return ISSCABAC_OK;
}

int isscabac_stream_encode_start(isscabac_stream *s) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (s->encoding) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeStart: the bitstream is already open for writing");  // ref asserts, CABAC_BitstreamFile.cpp:52
  HIP_TRY(e, hipSetDevice(e->device));
  if (const int rc = live_stop(s)) return rc;
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  if (n && !ctx) return engine_fail(s->e, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  if (n <= kInlineOps) {  // (the per-bin callers: no heap traffic)
    uint16_t small[kInlineOps];
    for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &small[i]);
    return isscabac_stream_encode_ops(s, bins, small, n);
  }
  std::vector<uint16_t> ops(n);
  for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &ops[i]);
  return isscabac_stream_encode_ops(s, bins, ops.data(), n);
}

int isscabac_stream_num_bits(isscabac_stream *s, uint64_t *bits) {
  if (!s || !bits) return ISSCABAC_ERR_ARG;
  if (!s->e) return ISSCABAC_ERR_STATE;
  isscabac_engine *e = s->e;
  HIP_TRY(e, hipSetDevice(e->device));
  const int rc = stream_flush_pending(s);
  if (rc) return rc;
  *bits = 8ull * s->mb->enc_bytes;  // CABAC_BitstreamFile::getNumberOfWrittenBits: no held bits before finish()
  return ISSCABAC_OK;
}

int isscabac_stream_encode_finish(isscabac_stream *s, uint8_t *out, uint64_t cap, uint64_t *nbytes) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (!s->encoding && !s->enc_done) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeFinish before encodeStart");
  HIP_TRY(e, hipSetDevice(e->device));
  if (!s->enc_done) {
    int rc = stream_flush_pending(s);
    if (rc) return rc;
    if ((rc = live_stop(s))) return rc;
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (!payload && nbytes) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeStart: null payload");
  if (nbytes > 0xFFFFFFF0ull) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeStart: payload too large");
  HIP_TRY(e, hipSetDevice(e->device));
  int rc = live_stop(s);
  if (rc) return rc;
  rc = stream_grow(e, &s->d_payload, &s->payload_cap, nbytes + 16, 0);
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeBin before decodeStart");
  if (n == 0) return ISSCABAC_OK;
  if (!ops || !bins_out) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++)
    if (const char *why = op_problem(s, ops[i])) return engine_fail(e, ISSCABAC_ERR_ARG, "%s (op 0x%04X, %d contexts)", why, ops[i], s->nctx);
  HIP_TRY(e, hipSetDevice(e->device));
  if (n <= kLiveMaxOps && s->live_on) {
    int rc;
    if (s->tracing && s->trace_cap[1] < 4 * (s->dec_total + n)) {
      uint32_t *tr;
      if ((rc = live_stop(s)) || (rc = trace_room(s, 1, s->dec_total, n, &tr))) return rc;
    }
    uint64_t a;
    if ((rc = live_call(s, kLiveDec, n, ops, nullptr, &a))) return rc;
    if (n == 1) bins_out[0] = (uint8_t)(a >> 32);
    else memcpy(bins_out, s->mb->bins, n);
    s->dec_total += n;
    return ISSCABAC_OK;
  }
  uint32_t *tr;
  int rc = live_stop(s);
  if (rc) return rc;
  if ((rc = trace_room(s, 1, s->dec_total, n, &tr))) return rc;
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  if (n && !ctx) return engine_fail(s->e, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  if (n <= kInlineOps) {
    uint16_t small[kInlineOps];
    for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &small[i]);
    return isscabac_stream_decode_ops(s, small, n, bins_out);
  }
  std::vector<uint16_t> ops(n);
  for (uint32_t i = 0; i < n; i++) op_from_u8(s, ctx[i], &ops[i]);
  return isscabac_stream_decode_ops(s, ops.data(), n, bins_out);
}

int isscabac_stream_decode_finish(isscabac_stream *s, uint32_t *status) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeFinish before decodeStart");
  HIP_TRY(e, hipSetDevice(e->device));
  if (const int rc = live_stop(s)) return rc;
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
  if (!s->e) return ISSCABAC_ERR_STATE;  // its engine is gone (isscabac_destroy): only isscabac_stream_close is left
  isscabac_engine *e = s->e;
  if (!s->tracing) return engine_fail(e, ISSCABAC_ERR_STATE, "getEncoderStats: tracing is off (isscabac_stream_set_trace)");
  if (ctx < 0 || ctx >= s->nctx) return engine_fail(e, ISSCABAC_ERR_ARG, "Error: invalid context index");
  HIP_TRY(e, hipSetDevice(e->device));
  const int side = decoder ? 1 : 0;
  if (!decoder && s->encoding) {
    const int rc = stream_flush_pending(s);
    if (rc) return rc;
  }
  if (const int rc = live_stop(s)) return rc;  // (its records reach device memory for a copy engine when it has left)
  const uint64_t n = decoder ? s->dec_total : s->bins_total;
  std::vector<uint32_t> rec(n);
  if (n) {
    HIP_TRY(e, hipMemcpyAsync(rec.data(), s->d_trace[side], 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  return isscabac_trace_stats(rec.data(), n, ctx, decoder, trace5, cap_steps, nsteps, transitions);
}

}  // extern "C"
