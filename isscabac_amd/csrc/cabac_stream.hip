// cabac_stream.hip -- single-stream handle of include/isscabac.h: the reference's per-bin commands
// (encodeStart / encodeBin / getNumBits / encodeFinish / decodeStart / decodeBin / decodeFinish,
// ref: CABAC/SimpleCABACMex.cpp:186-354) served by one-lane HIP kernels on device-resident coder state.
//
// Same per-lane code as the batch kernels (cabac_lane.h); the differences are the cadence -- bytes leave `low`
// after EVERY bin, exactly like testAndWriteOut (ref: CABAC_ArithmeticEncoder.cpp:369-375), so the byte count that
// getNumBits reports between bins is the reference's -- and that the state survives between launches.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "cabac_internal.h"
#include "cabac_lane.h"

namespace isscabac {

__device__ const uint32_t s_lps_rows[64] = {ISSCABAC_LPS_ROWS};
struct StreamNextTab {
  uint32_t e[64];
  constexpr StreamNextTab() : e{} {
    constexpr uint8_t trans[64] = {ISSCABAC_TRANS_LPS};
    for (uint32_t s = 0; s < 64; s++) e[s] = next_state_entry(s, trans[s]);
  }
};
__device__ const StreamNextTab s_next{};

struct StreamDev {
  EncCore enc;
  uint32_t enc_bytes;  // bytes written to the sink so far
  RefDec dec;  // the per-bin handle decodes in the reference's own formulation (cabac_lane.h: RefDec): one bin per call,
               // nothing to gain from look-ahead, and exact even on streams no encoder produces
  uint32_t dec_pos, dec_len, dec_last;  // bytes handed out so far, payload length, the last byte handed out
  uint8_t q_enc[256], q_dec[256];  // doubled context states, encoder and decoder sets (ref: :175-176)
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
__global__ void k_stream_reset(StreamDev *st, const uint8_t *init /* 256 packed states */) {
  const int t = threadIdx.x;
  st->q_enc[t] = (uint8_t)q_from_packed(init[t]);
  st->q_dec[t] = (uint8_t)q_from_packed(init[t]);
}

__global__ void k_stream_enc_start(StreamDev *st) {
  enc_start(st->enc);
  st->enc_bytes = 0;
}

__global__ void k_stream_enc_bins(StreamDev *st, const uint8_t *bins, const uint8_t *ctx, uint32_t n, uint8_t *out) {
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalStates cs{st->q_enc};
  GlobalSink sink{out, st->enc_bytes};
  for (uint32_t i = 0; i < n; i++) {
    if (ctx[i] == kBypass) enc_bin_ep(e, bins[i] & 1u);
    else enc_bin(e, (uint32_t)bins[i] << 1, ctx[i], cs);
    enc_flush(e, sink);
  }
  st->enc = e;
  st->enc_bytes = sink.pos;
}

__global__ void k_stream_enc_finish(StreamDev *st, uint8_t *out) {
  if (threadIdx.x != 0) return;
  EncCore e = st->enc;
  GlobalSink sink{out, st->enc_bytes};
  enc_finish(e, sink);
  st->enc = e;
  st->enc_bytes = sink.pos;
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

__global__ void k_stream_dec_bins(StreamDev *st, const uint8_t *payload, const uint8_t *ctx, uint32_t n, uint8_t *bins_out) {
  if (threadIdx.x != 0) return;
  RefDec d = st->dec;
  GlobalStates cs{st->q_dec};
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  for (uint32_t i = 0; i < n; i++)
    bins_out[i] = (uint8_t)(ctx[i] == kBypass ? ref_dec_bin_ep(d, rd) : ref_dec_bin(d, ctx[i], cs, rd));
  st->dec = d;
  st->dec_pos = rd.nread;
  st->dec_last = rd.last;
}

__global__ void k_stream_dec_finish(StreamDev *st, const uint8_t *payload, uint32_t *status) {
  if (threadIdx.x != 0) return;
  RefDec d = st->dec;
  RefReader rd{payload, st->dec_len, st->dec_pos, st->dec_last};
  *status = ref_dec_finish(d, rd, st->dec_len);
  st->dec = d;
}

}  // namespace isscabac

using namespace isscabac;

struct isscabac_stream {
  isscabac_engine *e = nullptr;
  int nctx = 0;
  StreamDev *st = nullptr;
  uint8_t *d_init = nullptr;
  // encoder side
  bool encoding = false, enc_done = false;
  std::vector<uint8_t> pend_bins, pend_ctx;
  uint64_t bins_total = 0;  // bins already on the device
  uint8_t *d_out = nullptr;
  uint64_t out_cap = 0;
  uint8_t *d_io = nullptr;  // staging for bins/ctx/decoded bins
  uint64_t io_cap = 0;
  // decoder side
  bool decoding = false;
  uint8_t *d_payload = nullptr;
  uint64_t payload_cap = 0, payload_len = 0;
  uint32_t *d_status = nullptr;
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

// code the queued bins on the device
static int stream_flush_pending(isscabac_stream *s) {
  isscabac_engine *e = s->e;
  const uint64_t n = s->pend_bins.size();
  if (n == 0) return ISSCABAC_OK;
  int rc;
  if ((rc = stream_grow(e, &s->d_io, &s->io_cap, 2 * n, 0))) return rc;
  uint32_t enc_bytes = 0;
  HIP_TRY(e, hipMemcpyAsync(&enc_bytes, &s->st->enc_bytes, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if ((rc = stream_grow(e, &s->d_out, &s->out_cap, isscabac_stream_bound((uint32_t)(s->bins_total + n)) + 16, enc_bytes))) return rc;
  HIP_TRY(e, hipMemcpyAsync(s->d_io, s->pend_bins.data(), n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(s->d_io + n, s->pend_ctx.data(), n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_stream_enc_bins, dim3(1), dim3(64), 0, e->stream, s->st, s->d_io, s->d_io + n, (uint32_t)n, s->d_out);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  s->bins_total += n;
  s->pend_bins.clear();
  s->pend_ctx.clear();
  return ISSCABAC_OK;
}

extern "C" {

int isscabac_stream_open(isscabac_engine *e, const uint8_t *ctx_init, int nctx, isscabac_stream **out) {
  if (!e || !out) return ISSCABAC_ERR_ARG;
  *out = nullptr;
  if (!ctx_init || nctx < 1 || nctx > ISSCABAC_MAX_CTX) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_stream_open: nctx must be 1..%d", ISSCABAC_MAX_CTX);
  uint8_t init[256];
  for (int c = 0; c < 256; c++) init[c] = 1;  // ContextModel() default (mps 1, state 0), ref: ContextModel.cpp:54
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
      (err = hipMalloc(reinterpret_cast<void **>(&s->d_init), 256)) != hipSuccess ||
      (err = hipMalloc(reinterpret_cast<void **>(&s->d_status), 4)) != hipSuccess ||
      (err = hipMemcpyAsync(s->d_init, init, 256, hipMemcpyHostToDevice, e->stream)) != hipSuccess) {
    isscabac_stream_close(s);
    return engine_fail(e, ISSCABAC_ERR_HIP, "isscabac_stream_open: %s", hipGetErrorString(err));
  }
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
  for (void *p : {(void *)s->st, (void *)s->d_init, (void *)s->d_out, (void *)s->d_io, (void *)s->d_payload, (void *)s->d_status})
    if (p) (void)hipFree(p);
  delete s;
}

int isscabac_stream_encode_start(isscabac_stream *s) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (s->encoding) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeStart: the bitstream is already open for writing");  // ref asserts, CABAC_BitstreamFile.cpp:52
  HIP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_stream_enc_start, dim3(1), dim3(1), 0, e->stream, s->st);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  s->encoding = true;
  s->enc_done = false;
  s->bins_total = 0;
  s->pend_bins.clear();
  s->pend_ctx.clear();
  return ISSCABAC_OK;
}

int isscabac_stream_encode_bins(isscabac_stream *s, const uint8_t *bins, const uint8_t *ctx, uint32_t n) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->encoding) return engine_fail(e, ISSCABAC_ERR_STATE, "encodeBin before encodeStart");
  if (n && (!bins || !ctx)) return engine_fail(e, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++) {
    if (bins[i] > 1) return engine_fail(e, ISSCABAC_ERR_ARG, "Error: invalid input 3, bin to be encoded should either be 1 or 0");  // ref: :227
    if (ctx[i] != ISSCABAC_BYPASS && ctx[i] >= s->nctx)
      return engine_fail(e, ISSCABAC_ERR_ARG, "encodeBin: context %u out of range (%d contexts)", ctx[i], s->nctx);
  }
  s->pend_bins.insert(s->pend_bins.end(), bins, bins + n);
  s->pend_ctx.insert(s->pend_ctx.end(), ctx, ctx + n);
  if (s->pend_bins.size() >= (1u << 16)) {
    HIP_TRY(e, hipSetDevice(e->device));
    return stream_flush_pending(s);
  }
  return ISSCABAC_OK;
}

int isscabac_stream_num_bits(isscabac_stream *s, uint64_t *bits) {
  if (!s || !bits) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  HIP_TRY(e, hipSetDevice(e->device));
  int rc = stream_flush_pending(s);
  if (rc) return rc;
  uint32_t enc_bytes = 0;
  HIP_TRY(e, hipMemcpyAsync(&enc_bytes, &s->st->enc_bytes, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  *bits = 8ull * enc_bytes;  // CABAC_BitstreamFile::getNumberOfWrittenBits: no held bits before finish()
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
    uint32_t before = 0;
    HIP_TRY(e, hipMemcpyAsync(&before, &s->st->enc_bytes, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if ((rc = stream_grow(e, &s->d_out, &s->out_cap, isscabac_stream_bound((uint32_t)s->bins_total) + 16, before))) return rc;
    hipLaunchKernelGGL(k_stream_enc_finish, dim3(1), dim3(64), 0, e->stream, s->st, s->d_out);
    HIP_TRY(e, hipGetLastError());
    s->enc_done = true;
    s->encoding = false;
  }
  uint32_t total = 0;
  HIP_TRY(e, hipMemcpyAsync(&total, &s->st->enc_bytes, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
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
  return ISSCABAC_OK;
}

int isscabac_stream_decode_bins(isscabac_stream *s, const uint8_t *ctx, uint32_t n, uint8_t *bins_out) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeBin before decodeStart");
  if (n == 0) return ISSCABAC_OK;
  if (!ctx || !bins_out) return engine_fail(e, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++)
    if (ctx[i] != ISSCABAC_BYPASS && ctx[i] >= s->nctx)
      return engine_fail(e, ISSCABAC_ERR_ARG, "decodeBin: context %u out of range (%d contexts)", ctx[i], s->nctx);
  HIP_TRY(e, hipSetDevice(e->device));
  int rc = stream_grow(e, &s->d_io, &s->io_cap, 2ull * n, 0);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(s->d_io, ctx, n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_stream_dec_bins, dim3(1), dim3(64), 0, e->stream, s->st, s->d_payload, s->d_io, n, s->d_io + n);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(bins_out, s->d_io + n, n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_stream_decode_finish(isscabac_stream *s, uint32_t *status) {
  if (!s) return ISSCABAC_ERR_ARG;
  isscabac_engine *e = s->e;
  if (!s->decoding) return engine_fail(e, ISSCABAC_ERR_STATE, "decodeFinish before decodeStart");
  HIP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_stream_dec_finish, dim3(1), dim3(64), 0, e->stream, s->st, s->d_payload, s->d_status);
  HIP_TRY(e, hipGetLastError());
  uint32_t st = 0;
  HIP_TRY(e, hipMemcpyAsync(&st, s->d_status, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (status) *status = st;
  s->decoding = false;
  return ISSCABAC_OK;
}

}  // extern "C"
