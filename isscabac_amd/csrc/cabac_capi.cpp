// cabac_capi.cpp -- the C ABI of include/isscabac.h on top of the HIP launchers.
//
// Host-side only: argument checking, workspace ownership, staging for host-buffer calls, error text.
// There is deliberately NO CPU coding path in this library: without a working HIP device every coding
// call fails with ISSCABAC_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/isscabac.h"
#include "cabac_internal.h"

#ifndef ISSCABAC_DEC_PAIR_WAVES_PER_CU
#define ISSCABAC_DEC_PAIR_WAVES_PER_CU 8  // launches of at most this many waves of streams per CU decode on the wave pair
#endif

using namespace isscabac;

static thread_local std::string g_create_err;

int isscabac::engine_fail(isscabac_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  else g_create_err = buf;
  return code;
}
#define fail isscabac::engine_fail

static int grow(isscabac_engine *e, void **p, uint64_t *have, uint64_t need) {
  if (*have >= need) return ISSCABAC_OK;
  if (*p) {
    HIP_TRY(e, hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  HIP_TRY(e, hipMalloc(p, need));
  *have = need;
  return ISSCABAC_OK;
}

// the engine's four device words (cabac_internal.h: flag) and its pinned, device-mapped size words, once per engine
static int ensure_words(isscabac_engine *e) {
  if (!e->flag) {
    HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&e->flag), 16));
    HIP_TRY(e, hipMemset(e->flag, 0, 16));
  }
  if (!e->h_total) HIP_TRY(e, hipHostMalloc(reinterpret_cast<void **>(&e->h_total), 64, hipHostMallocMapped));
  if (!e->h_total_dev) {  // (the symbol calls may have allocated the block first: cabac_symbols.hip)
    void *dv = nullptr;
    HIP_TRY(e, hipHostGetDevicePointer(&dv, e->h_total, 0));
    e->h_total_dev = static_cast<uint64_t *>(dv);
  }
  return ISSCABAC_OK;
}

extern "C" {

int isscabac_abi_version(void) { return 2; }

int isscabac_create(int device, isscabac_engine **out) {
  if (!out) return fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_create: out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev <= 0)
    return fail(nullptr, ISSCABAC_ERR_HIP, "isscabac_create: no HIP device (%s)", hipGetErrorString(err));
  if (device < 0 || device >= ndev)
    return fail(nullptr, ISSCABAC_ERR_ARG, "isscabac_create: device %d out of range (%d devices)", device, ndev);
  isscabac_engine *e = new isscabac_engine;
  e->device = device;
  if ((err = hipSetDevice(device)) != hipSuccess || (err = hipStreamCreate(&e->own_stream)) != hipSuccess) {
    const int rc = fail(nullptr, ISSCABAC_ERR_HIP, "isscabac_create: %s", hipGetErrorString(err));
    delete e;
    return rc;
  }
  e->stream = e->own_stream;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->num_cus = cus;
  *out = e;
  return ISSCABAC_OK;
}

void isscabac_destroy(isscabac_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (!e->handles.empty()) engine_orphan_handles(e);  // per-bin handles still open: their live kernels leave, they can only be closed from now on
  (void)hipStreamSynchronize(e->stream);
  if (e->slots) (void)hipFree(e->slots);
  if (e->lens) (void)hipFree(e->lens);
  if (e->scan_tmp) (void)hipFree(e->scan_tmp);
  if (e->flag) (void)hipFree(e->flag);
  if (e->order) (void)hipFree(e->order);
  if (e->order_tmp) (void)hipFree(e->order_tmp);
  for (void *p : e->stage)
    if (p) (void)hipFree(p);
  if (e->prof) {
    for (int i = 0; i < isscabac_engine::kProfRing; i++)
      for (hipEvent_t ev : e->prof[i].ev) (void)hipEventDestroy(ev);
    delete[] e->prof;
  }
  for (int i = 0; i < isscabac_engine::kPinSlots; i++) {
    if (e->pin[i]) (void)hipHostFree(e->pin[i]);
    if (e->pin_ev[i]) (void)hipEventDestroy(e->pin_ev[i]);
  }
  for (hipEvent_t ev : e->pipe_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (e->h2d_stream) (void)hipStreamDestroy(e->h2d_stream);
  if (e->d2h_stream) (void)hipStreamDestroy(e->d2h_stream);
  if (e->h_total) (void)hipHostFree(e->h_total);
  if (e->sym_pin) (void)hipHostFree(e->sym_pin);
  if (e->wide_init) (void)hipFree(e->wide_init);
  if (e->mat_bins) (void)hipFree(e->mat_bins);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

const char *isscabac_last_error(const isscabac_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int isscabac_set_tuning(isscabac_engine *e, int mode) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (mode != ISSCABAC_TUNE_AUTO && mode != ISSCABAC_TUNE_THROUGHPUT && mode != ISSCABAC_TUNE_LATENCY && mode != ISSCABAC_TUNE_LATENCY_SINGLE)
    return fail(e, ISSCABAC_ERR_ARG, "isscabac_set_tuning: unknown mode %d", mode);
  e->tuning = mode;
  return ISSCABAC_OK;
}

const char *isscabac_last_instance(const isscabac_engine *e, int which) {
  return e && (which == 0 || which == 1) && e->last_instance[which] ? e->last_instance[which] : "";
}

int isscabac_set_stream(isscabac_engine *e, void *hip_stream) {
  if (!e) return ISSCABAC_ERR_ARG;
  e->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
  return ISSCABAC_OK;
}

// ---- context initialisation ------------------------------------------------------------------------
// xMapProbabilityToState, ref: CABAC/CABAC_ContextModelsInit.cpp:124-148 (double arithmetic, host)
static uint8_t p0_to_packed(double p0) {
  const bool zero_is_mps = p0 >= 0.5;
  double p_lps = zero_is_mps ? 1.0 - p0 : p0;
  if (p_lps < 0.01875) p_lps = 0.01875;
  int state = (int)std::round(62.0 * std::log10(2.0 * p_lps) / std::log10(2.0 * 0.01875));
  state = state < 0 ? 0 : (state > 62 ? 62 : state);
  return (uint8_t)((state << 1) | (zero_is_mps ? 0 : 1));
}

int isscabac_init_by_prob(const double *p0, int nctx, uint8_t *packed_out) {
  if (!p0 || !packed_out || nctx < 0) return ISSCABAC_ERR_ARG;
  for (int c = 0; c < nctx; c++) {
    if (!(p0[c] >= 0.0 && p0[c] <= 1.0)) return ISSCABAC_ERR_ARG;  // the reference asserts, :127
    packed_out[c] = p0_to_packed(p0[c]);
  }
  return ISSCABAC_OK;
}

int isscabac_init_by_state(const double *triples, int nctx, uint8_t *packed_out) {
  if (!triples || !packed_out || nctx < 0) return ISSCABAC_ERR_ARG;
  for (int c = 0; c < nctx; c++) {
    const unsigned mps = (unsigned)triples[3 * c + 1], state = (unsigned)triples[3 * c + 2];
    // the reference stores (state << 1) + mps unchecked; state 63 makes its own decoder index out of
    // bounds (CABAC_ArithmeticDecoder.cpp:94), so anything outside 0..62 / 0..1 is refused here
    if (mps > 1 || state > 62) return ISSCABAC_ERR_ARG;
    packed_out[c] = (uint8_t)((state << 1) | mps);
  }
  return ISSCABAC_OK;
}

uint64_t isscabac_stream_bound(uint32_t nbins) { return (6ull * nbins + 9 + 7) / 8; }

// ---- helpers ---------------------------------------------------------------------------------------
static int check_shape(isscabac_engine *e, const char *who, uint64_t S, uint32_t nbins, uint32_t pitch,
                       const uint8_t *ctx_init, int nctx, BatchShape *sh) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (S == 0) return fail(e, ISSCABAC_ERR_ARG, "%s: S must be > 0", who);
  if (pitch == 0 || (pitch & 15u)) return fail(e, ISSCABAC_ERR_ARG, "%s: pitch (%u) must be a positive multiple of 16", who, pitch);
  if (nbins > pitch) return fail(e, ISSCABAC_ERR_ARG, "%s: nbins (%u) exceeds pitch (%u)", who, nbins, pitch);
  if (pitch > ISSCABAC_MAX_PITCH)
    return fail(e, ISSCABAC_ERR_ARG, "%s: pitch (%u) exceeds %u bins per stream (use a stream handle for longer streams)", who,
                pitch, (unsigned)ISSCABAC_MAX_PITCH);
  if (nctx < 1 || nctx > ISSCABAC_MAX_CTX) return fail(e, ISSCABAC_ERR_ARG, "%s: nctx (%d) must be 1..%d", who, nctx, ISSCABAC_MAX_CTX);
  if (!ctx_init) return fail(e, ISSCABAC_ERR_ARG, "%s: ctx_init is null", who);
  if (S > (1ull << 38)) return fail(e, ISSCABAC_ERR_ARG, "%s: S too large", who);
  memset(sh, 0, sizeof *sh);
  sh->S = S;
  sh->nbins = nbins;
  sh->pitch = pitch;
  sh->nctx = (uint32_t)nctx;
  sh->d_order = nullptr;
  // Kernel instance.  The throughput instance hides its three dependent LDS round trips per bin behind the other waves
  // of the SIMD and needs about three of them for that.  Measured on MI355X (tools/s_sweep.py, profiles/r02): with two
  // waves per SIMD (128 Ki streams) the latency-oriented instance is 5 % faster, with one 17 %; from 2.5 waves per SIMD
  // on the throughput instance wins by 3-5 %.  The switch sits at two waves per SIMD.
  // More than 128 contexts: the state image (256 bytes per four contexts and wave) leaves at most 2.5 waves per SIMD
  // whatever the batch's size, the latency instance's regime: 190 contexts 514 -> 542 Gbin/s, 253 contexts 434 -> 483.
  sh->latency = e->tuning == ISSCABAC_TUNE_LATENCY || e->tuning == ISSCABAC_TUNE_LATENCY_SINGLE ||
                (e->tuning == ISSCABAC_TUNE_AUTO && (nctx > 128 || S <= 2ull * (uint64_t)e->num_cus * 4 * 64));
  // 65 .. 128 contexts leave three waves on a SIMD: there the encoder's latency instance still wins (127 contexts, 1 Mi streams:
  // 3.34 against 3.43 .. 3.48 ms), the decoder's does not (3.64 against 3.19) -- profiles/r06/ab_ctx_rates_by_tuning.txt
  sh->latency_enc = sh->latency || (e->tuning == ISSCABAC_TUNE_AUTO && nctx > 64);
  // The decoder's helper wave (cabac_pair.hip: k_decode_pair) takes a fifth of the instructions off the decoding wave but not
  // its chain: same-box sweep 16 Ki streams 0.358 -> 0.306 ms, 64 Ki 0.364 -> 0.363, 128 Ki 0.460 -> 0.486 (profiles/r05) -- with the
  // hardware's own placement of its two-wave workgroups.  With the pair's waves on the SIMDs a per-CU ticket names (round 6:
  // pair_pin_role) 64 Ki 0.369 -> 0.321, 128 Ki 0.463 -> 0.448 (profiles/r06/y2_dpair_pin_sweep.txt): every launch of at most
  // eight waves of streams per CU, which is all that its 18.5 KiB of LDS per pair lets a CU hold at once
  sh->tiny = S <= (uint64_t)e->num_cus * ISSCABAC_DEC_PAIR_WAVES_PER_CU * 64;
  sh->pair = e->tuning != ISSCABAC_TUNE_LATENCY_SINGLE;
  for (int c = 0; c < nctx; c++) {
    if ((ctx_init[c] >> 1) > 62) return fail(e, ISSCABAC_ERR_ARG, "%s: ctx_init[%d] has state 63 (allowed: 0..62)", who, c);
    sh->ctx_init[c] = ctx_init[c];
  }
  // contexts the caller did not initialise behave like the reference's default ContextModel(): (mps 1, state 0)
  for (int c = nctx; c < 256; c++) sh->ctx_init[c] = 1;
  return ISSCABAC_OK;
}

// Ragged batch (per-stream bin counts on the device): sort the stream ids by length so that the 64 streams of a wave
// are alike (cabac_order.hip).  Enqueued on the engine's stream in front of the coder kernel; sh->d_order stays null for
// small or uniform batches.
static int bucket_by_length(isscabac_engine *e, BatchShape *sh, const uint32_t *d_nbins) {
  sh->d_order = nullptr;
  if (!d_nbins || sh->S < 4096 || sh->S >= (1ull << 31) || e->tuning == ISSCABAC_TUNE_LATENCY ||
      e->tuning == ISSCABAC_TUNE_LATENCY_SINGLE)
    return ISSCABAC_OK;
  int rc;
  if ((rc = grow(e, reinterpret_cast<void **>(&e->order), &e->order_bytes, sh->S * sizeof(uint32_t)))) return rc;
  const uint64_t need = order_tmp_bytes(sh->S);
  if ((rc = grow(e, &e->order_tmp, &e->order_tmp_cap, need))) return rc;
  HIP_TRY(e, launch_order(d_nbins, sh->S, sh->nbins, e->order, e->order_tmp, e->order_tmp_cap, e->stream));
  sh->d_order = e->order;
  return ISSCABAC_OK;
}

static bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// ---- profiling: one event set per batch call, read back only at synchronisation points ---------------------------
// read the recorded-but-unread slots (everything enqueued has completed: the caller synchronised the stream)
static int prof_drain(isscabac_engine *e) {
  while (e->prof_count > 0) {
    const int i = (e->prof_head - e->prof_count + isscabac_engine::kProfRing) % isscabac_engine::kProfRing;
    isscabac_engine::ProfSlot &p = e->prof[i];
    float ms = 0;
    if (p.kind == 0) {
      for (int k = 0; k < 3; k++) {
        HIP_TRY(e, hipEventElapsedTime(&ms, p.ev[k], p.ev[k + 1]));
        e->last_ms[k] = ms;
        e->sum_ms[k] += ms;
        e->sum_n[k]++;
        if (e->samples[k].size() < isscabac_engine::kMaxSamples) e->samples[k].push_back(ms);
        else e->samples_dropped[k]++;
      }
    } else {
      HIP_TRY(e, hipEventElapsedTime(&ms, p.ev[0], p.ev[1]));
      e->last_ms[3] = ms;
      e->sum_ms[3] += ms;
      e->sum_n[3]++;
      if (e->samples[3].size() < isscabac_engine::kMaxSamples) e->samples[3].push_back(ms);
      else e->samples_dropped[3]++;
    }
    e->prof_count--;
  }
  return ISSCABAC_OK;
}
// next slot of the ring (nullptr when profiling is off); a full ring is read out first, which waits for the stream
static int prof_slot(isscabac_engine *e, int kind, isscabac_engine::ProfSlot **out) {
  *out = nullptr;
  if (!e->profiling) return ISSCABAC_OK;
  if (!e->prof) {
    e->prof = new isscabac_engine::ProfSlot[isscabac_engine::kProfRing]();
    for (int i = 0; i < isscabac_engine::kProfRing; i++)
      for (hipEvent_t &ev : e->prof[i].ev) HIP_TRY(e, hipEventCreate(&ev));
  }
  if (e->prof_count == isscabac_engine::kProfRing) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const int rc = prof_drain(e);
    if (rc) return rc;
  }
  *out = &e->prof[e->prof_head];
  (*out)->kind = kind;
  e->prof_head = (e->prof_head + 1) % isscabac_engine::kProfRing;
  e->prof_count++;
  return ISSCABAC_OK;
}
#define PROF_MARK(e, slot, i)                                                   \
  do {                                                                          \
    if (slot) HIP_TRY(e, hipEventRecord((slot)->ev[i], (e)->stream));           \
  } while (0)
// wait for everything enqueued on the engine's stream, then collect what became readable
static int engine_sync(isscabac_engine *e) {
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return e->prof_count ? prof_drain(e) : ISSCABAC_OK;
}

// ---- batch encode, device buffers ---------------------------------------------------------------------
int isscabac_encode_batch_dev_async(isscabac_engine *e, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                                    uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                    uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets) {
  BatchShape sh;
  int rc = check_shape(e, "isscabac_encode_batch_dev", S, nbins, pitch, ctx_init, nctx, &sh);
  if (rc) return rc;
  if (!d_bins || !d_ctx || !d_payload || !d_offsets) return fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_batch_dev: null buffer");
  if (misaligned(d_bins) || misaligned(d_ctx)) return fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_batch_dev: bins/ctx must be 16-byte aligned");
  ENGINE_ENTER(e);
  const uint64_t stride = slot_stride_for(nbins);
  // (growing the scratch frees the old one, which waits for the device: only the first call of a size pays that)
  if ((rc = grow(e, reinterpret_cast<void **>(&e->slots), &e->slots_bytes, S * stride))) return rc;
  if ((rc = grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->scan_tmp, &e->scan_tmp_bytes, scan_tmp_bytes(S)))) return rc;
  if ((rc = ensure_words(e))) return rc;
  isscabac_engine::ProfSlot *ps;
  if ((rc = prof_slot(e, 0, &ps))) return rc;

  if ((rc = bucket_by_length(e, &sh, d_nbins))) return rc;
  PROF_MARK(e, ps, 0);
  HIP_TRY(e, launch_encode(sh, d_bins, d_ctx, d_nbins, e->slots, stride, e->lens, e->stream));
  e->last_instance[0] = sh.instance;
  PROF_MARK(e, ps, 1);
  // the payload size (offsets[S]) also lands in the engine's pinned word 0, written by the scan kernel itself: the host reads it
  // after its synchronisation, no copy command in the stream
  if (scan_gather_is_fused(S)) {  // a handful of streams: offsets and payload in one launch (timed as the scan; the gather slot reads ~0)
    HIP_TRY(e, launch_scan_gather(e->slots, stride, e->lens, S, d_offsets, e->scan_tmp, d_payload, payload_cap, e->stream, nullptr,
                                  e->h_total_dev));
    PROF_MARK(e, ps, 2);
  } else if (scan_gather_is_paired(S, e->flag + 2)) {  // up to 256 Ki streams: two launches instead of four (launch gaps are 5 % of a 64 Ki step)
    HIP_TRY(e, launch_scan_partial_top(e->lens, S, d_offsets, e->scan_tmp, e->flag + 2, e->h_total_dev, e->stream));
    PROF_MARK(e, ps, 2);
    HIP_TRY(e, launch_gather_scan(e->slots, stride, e->lens, S, d_offsets, e->scan_tmp, d_payload, payload_cap, e->stream));
  } else {
    HIP_TRY(e, launch_scan(e->lens, S, d_offsets, e->scan_tmp, e->stream, e->h_total_dev));
    PROF_MARK(e, ps, 2);
    HIP_TRY(e, launch_gather(e->slots, stride, e->lens, d_offsets, S, d_payload, payload_cap, e->stream));
  }
  PROF_MARK(e, ps, 3);
  e->pending_total = true;
  e->pending_cap = payload_cap;
  return ISSCABAC_OK;
}

int isscabac_sync(isscabac_engine *e, uint64_t *total_bytes) {
  if (!e) return ISSCABAC_ERR_ARG;
  ENGINE_ENTER(e);
  const int rc = engine_sync(e);
  if (rc) return rc;
  if (e->pending_total) {
    e->pending_total = false;
    const uint64_t total = *e->h_total;
    if (total_bytes) *total_bytes = total;
    if (total > e->pending_cap)
      return fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_batch_dev: payload needs %llu bytes, capacity is %llu",
                  (unsigned long long)total, (unsigned long long)e->pending_cap);
  }
  return ISSCABAC_OK;
}

int isscabac_encode_batch_dev(isscabac_engine *e, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                              uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                              uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint64_t *total_bytes) {
  const int rc = isscabac_encode_batch_dev_async(e, d_bins, d_ctx, d_nbins, S, nbins, pitch, ctx_init, nctx, d_payload,
                                                 payload_cap, d_offsets);
  return rc ? rc : isscabac_sync(e, total_bytes);
}

int isscabac_decode_batch_dev_async(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes,
                                    const uint64_t *d_offsets, const uint8_t *d_ctx, const uint32_t *d_nbins, uint64_t S,
                                    uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *d_bins_out,
                                    uint32_t *d_status) {
  BatchShape sh;
  int rc = check_shape(e, "isscabac_decode_batch_dev", S, nbins, pitch, ctx_init, nctx, &sh);
  if (rc) return rc;
  if (!d_payload || !d_offsets || !d_ctx || !d_bins_out) return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch_dev: null buffer");
  if (misaligned(d_ctx) || misaligned(d_bins_out) || (reinterpret_cast<uintptr_t>(d_payload) & 3u))
    return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch_dev: ctx/bins_out must be 16-byte aligned, payload 4-byte aligned");
  ENGINE_ENTER(e);
  if ((rc = ensure_words(e))) return rc;
  isscabac_engine::ProfSlot *ps;
  if ((rc = prof_slot(e, 1, &ps))) return rc;
  // this call's flag word is zero: the previous call's second kernel left it so (the two words take turns).  The turn moves on
  // only when both kernels are in the stream; a failure in between restores "both words zero" with a memset
  uint32_t *flag = e->flag + e->flag_turn, *next_flag = e->flag + (e->flag_turn ^ 1);
  if ((rc = bucket_by_length(e, &sh, d_nbins))) return rc;
  PROF_MARK(e, ps, 0);
  hipError_t herr = launch_decode(sh, d_payload, payload_bytes, d_offsets, d_ctx, d_nbins, d_bins_out, d_status, flag, e->stream);
  if (herr == hipSuccess) {
    e->last_instance[1] = sh.instance;
    if (ps) herr = hipEventRecord(ps->ev[1], e->stream);
  }
  // streams that start with 0xFF (corrupt input): the reference's value register overflows on those, and only its own
  // formulation reproduces what it then decodes (cabac_lane.h: RefDec).  The kernel looks at the flag the decode kernel
  // left on the device and returns at once when it is clear, so the host never has to read it.
  if (herr == hipSuccess)
    herr = launch_decode_exact(sh, d_payload, payload_bytes, d_offsets, d_ctx, d_nbins, d_bins_out, d_status, flag, e->stream, next_flag);
  if (herr != hipSuccess) {
    (void)hipMemsetAsync(e->flag, 0, 2 * sizeof(uint32_t), e->stream);
    return fail(e, herr == hipErrorOutOfMemory ? ISSCABAC_ERR_NOMEM : ISSCABAC_ERR_HIP, "isscabac_decode_batch_dev: %s", hipGetErrorString(herr));
  }
  e->flag_turn ^= 1;
  return ISSCABAC_OK;
}

int isscabac_decode_batch_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                              const uint8_t *d_ctx, const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                              const uint8_t *ctx_init, int nctx, uint8_t *d_bins_out, uint32_t *d_status) {
  const int rc = isscabac_decode_batch_dev_async(e, d_payload, payload_bytes, d_offsets, d_ctx, d_nbins, S, nbins, pitch,
                                                 ctx_init, nctx, d_bins_out, d_status);
  if (rc) return rc;
  ENGINE_ENTER(e);
  return engine_sync(e);
}

// ---- batches of more than 255 contexts: 16-bit context indices (cabac_wide.hip) -------------------------------------
// pure argument validation (no device work): first thing in the host forms as well as in the device forms
static int wide_check(isscabac_engine *e, const char *who, uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (S == 0) return fail(e, ISSCABAC_ERR_ARG, "%s: S must be > 0", who);
  if (pitch == 0 || (pitch & 15u)) return fail(e, ISSCABAC_ERR_ARG, "%s: pitch (%u) must be a positive multiple of 16", who, pitch);
  if (nbins > pitch) return fail(e, ISSCABAC_ERR_ARG, "%s: nbins (%u) exceeds pitch (%u)", who, nbins, pitch);
  if (pitch > ISSCABAC_MAX_PITCH) return fail(e, ISSCABAC_ERR_ARG, "%s: pitch (%u) exceeds %u bins per stream", who, pitch, (unsigned)ISSCABAC_MAX_PITCH);
  if (nctx < 1 || nctx > ISSCABAC_MAX_CTX16) return fail(e, ISSCABAC_ERR_ARG, "%s: nctx (%d) must be 1..%d", who, nctx, ISSCABAC_MAX_CTX16);
  if (!ctx_init) return fail(e, ISSCABAC_ERR_ARG, "%s: ctx_init is null", who);
  if (S > (1ull << 31)) return fail(e, ISSCABAC_ERR_ARG, "%s: S too large", who);  // (S * pitch <= 2^56: no overflow in the callers' sizes)
  for (int c = 0; c < nctx; c++)
    if ((ctx_init[c] >> 1) > 62) return fail(e, ISSCABAC_ERR_ARG, "%s: ctx_init[%d] has state 63 (allowed: 0..62)", who, c);
  return ISSCABAC_OK;
}
static int wide_prepare(isscabac_engine *e, const char *who, uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init,
                        int nctx) {
  const int rc = wide_check(e, who, S, nbins, pitch, ctx_init, nctx);
  if (rc) return rc;
  uint8_t init[1024];
  memset(init, 1, sizeof init);  // contexts the caller did not initialise: the reference's default ContextModel(), (mps 1, state 0)
  memcpy(init, ctx_init, (size_t)nctx);
  ENGINE_ENTER(e);
  if (!e->wide_init) HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&e->wide_init), sizeof init));
  // (a blocking copy of 1 KiB out of this stack frame: the previous call's kernels have long read theirs -- every wide call synchronises)
  HIP_TRY(e, hipMemcpyAsync(e->wide_init, init, sizeof init, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_encode_batch16_dev(isscabac_engine *e, const uint8_t *d_bins, const uint16_t *d_ctx16, const uint32_t *d_nbins,
                                uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint64_t *total_bytes) {
  int rc = wide_prepare(e, "isscabac_encode_batch16_dev", S, nbins, pitch, ctx_init, nctx);
  if (rc) return rc;
  if (!d_bins || !d_ctx16 || !d_payload || !d_offsets) return fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_batch16_dev: null buffer");
  if (misaligned(d_bins) || misaligned(d_ctx16)) return fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_batch16_dev: bins/ctx must be 16-byte aligned");
  const uint64_t stride = slot_stride_for(nbins);
  if ((rc = grow(e, reinterpret_cast<void **>(&e->slots), &e->slots_bytes, S * stride))) return rc;
  if ((rc = grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->scan_tmp, &e->scan_tmp_bytes, scan_tmp_bytes(S)))) return rc;
  if ((rc = ensure_words(e))) return rc;
  HIP_TRY(e, launch_encode_wide(d_bins, d_ctx16, d_nbins, S, nbins, pitch, (uint32_t)nctx, e->wide_init, e->slots, stride,
                                e->lens, e->stream));
  HIP_TRY(e, launch_scan_gather(e->slots, stride, e->lens, S, d_offsets, e->scan_tmp, d_payload, payload_cap, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->h_total + 2, d_offsets + S, sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const uint64_t total = e->h_total[2];
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap)
    return fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_batch16_dev: payload needs %llu bytes, capacity is %llu",
                (unsigned long long)total, (unsigned long long)payload_cap);
  return ISSCABAC_OK;
}

int isscabac_decode_batch16_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                                const uint16_t *d_ctx16, const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                const uint8_t *ctx_init, int nctx, uint8_t *d_bins_out, uint32_t *d_status) {
  int rc = wide_prepare(e, "isscabac_decode_batch16_dev", S, nbins, pitch, ctx_init, nctx);
  if (rc) return rc;
  if (!d_payload || !d_offsets || !d_ctx16 || !d_bins_out) return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch16_dev: null buffer");
  if (misaligned(d_ctx16) || misaligned(d_bins_out) || (reinterpret_cast<uintptr_t>(d_payload) & 3u))
    return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch16_dev: ctx/bins_out must be 16-byte aligned, payload 4-byte aligned");
  HIP_TRY(e, launch_decode_wide(d_payload, payload_bytes, d_offsets, d_ctx16, d_nbins, S, nbins, pitch, (uint32_t)nctx, e->wide_init,
                                d_bins_out, d_status, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

// host-buffer forms: plain copies around the device forms (this path is about the API surface, not about the link)
int isscabac_encode_batch16(isscabac_engine *e, const uint8_t *bins, const uint16_t *ctx16, const uint32_t *nbins_per_stream, uint64_t S,
                            uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                            uint64_t *offsets, uint64_t *total_bytes) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!bins || !ctx16 || !payload || !offsets) return fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_batch16: null buffer");
  int rc;
  if ((rc = wide_check(e, "isscabac_encode_batch16", S, nbins, pitch, ctx_init, nctx))) return rc;  // before any allocation or copy
  ENGINE_ENTER(e);
  const uint64_t nb = S * pitch, dev_cap = S * isscabac_stream_bound(nbins);
  if ((rc = grow(e, &e->stage[0], &e->stage_bytes[0], nb))) return rc;
  if ((rc = grow(e, &e->stage[1], &e->stage_bytes[1], 2 * nb))) return rc;
  if ((rc = grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->stage[3], &e->stage_bytes[3], dev_cap))) return rc;
  if ((rc = grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->stage[0], bins, nb, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->stage[1], ctx16, 2 * nb, hipMemcpyHostToDevice, e->stream));
  if (nbins_per_stream) HIP_TRY(e, hipMemcpyAsync(e->stage[2], nbins_per_stream, S * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  uint64_t total = 0;
  rc = isscabac_encode_batch16_dev(e, static_cast<uint8_t *>(e->stage[0]), static_cast<uint16_t *>(e->stage[1]),
                                   nbins_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr, S, nbins, pitch, ctx_init, nctx,
                                   static_cast<uint8_t *>(e->stage[3]), dev_cap, static_cast<uint64_t *>(e->stage[4]), &total);
  if (rc) return rc;
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap)
    return fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_batch16: payload needs %llu bytes, capacity is %llu", (unsigned long long)total,
                (unsigned long long)payload_cap);
  HIP_TRY(e, hipMemcpyAsync(payload, e->stage[3], total, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipMemcpyAsync(offsets, e->stage[4], (S + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_decode_batch16(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint16_t *ctx16,
                            const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                            uint8_t *bins_out, uint32_t *status) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!payload || !offsets || !ctx16 || !bins_out) return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch16: null buffer");
  int rc;
  if ((rc = wide_check(e, "isscabac_decode_batch16", S, nbins, pitch, ctx_init, nctx))) return rc;  // before any allocation or copy
  for (uint64_t s = 0; s < S; s++)
    if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] > 0xFFFFFFFFull)
      return fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_batch16: offsets not monotone at stream %llu", (unsigned long long)s);
  ENGINE_ENTER(e);
  const uint64_t nb = S * pitch, total = offsets[S];
  if ((rc = grow(e, &e->stage[0], &e->stage_bytes[0], nb))) return rc;
  if ((rc = grow(e, &e->stage[1], &e->stage_bytes[1], 2 * nb))) return rc;
  if ((rc = grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->stage[3], &e->stage_bytes[3], total + 16))) return rc;
  if ((rc = grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = grow(e, &e->stage[5], &e->stage_bytes[5], S * sizeof(uint32_t)))) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->stage[1], ctx16, 2 * nb, hipMemcpyHostToDevice, e->stream));
  if (nbins_per_stream) HIP_TRY(e, hipMemcpyAsync(e->stage[2], nbins_per_stream, S * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->stage[3], payload, total, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->stage[4], offsets, (S + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->stage[0], 0, nb, e->stream));  // (columns behind nbins rounded up to 16 are not written by the kernel)
  rc = isscabac_decode_batch16_dev(e, static_cast<uint8_t *>(e->stage[3]), total, static_cast<uint64_t *>(e->stage[4]),
                                   static_cast<uint16_t *>(e->stage[1]), nbins_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr,
                                   S, nbins, pitch, ctx_init, nctx, static_cast<uint8_t *>(e->stage[0]), static_cast<uint32_t *>(e->stage[5]));
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(bins_out, e->stage[0], nb, hipMemcpyDeviceToHost, e->stream));
  if (status) HIP_TRY(e, hipMemcpyAsync(status, e->stage[5], S * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

// ---- batch coding, host buffers -----------------------------------------------------------------------
// What a MEX / FFI caller holding host arrays pays for is PCIe, not coding (4 B/bin cross the link against ~4 ms of
// kernels per GiB), so these entry points are built around the copies:
//   * caller memory is pageable: a hipMemcpy from it is staged by the runtime through one thread.  Here the bytes go through
//     the engine's own pinned ring (kPinSlots x 16 MiB): a few host threads copy a block into a pinned slot while the DMA
//     engine drains the previous slots, H2D and D2H on streams of their own, so that both directions and the kernels overlap;
//   * the batch is cut into groups of streams: group g is decoded while group g + 1's contexts are still arriving and
//     group g - 1's bins are already leaving;
//   * every device buffer is grow-only scratch of the engine: no allocation after the first call of a size.
}  // extern "C"

namespace isscabac {
// copy n bytes with a few threads (one memcpy stream tops out near 10 GB/s; the PCIe link takes 50+)
void parallel_copy(void *dst, const void *src, size_t n) {
  const size_t kMin = 1u << 20;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, hw), n / kMin + 1);
  if (nt <= 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  // (the share rounded UP before it is aligned: n / nt rounded down left the last n % nt bytes to nobody whenever it was
  // a multiple of 64 already -- found by the soak as a payload whose last byte stayed behind)
  const size_t per = ((n + nt - 1) / nt + 63) & ~(size_t)63;
  for (unsigned t = 1; t < nt; t++) {
    const size_t off = (size_t)t * per;
    if (off >= n) break;
    th.emplace_back([=] { memcpy((char *)dst + off, (const char *)src + off, std::min(per, n - off)); });
  }
  memcpy(dst, src, std::min(per, n));
  for (auto &t : th) t.join();
}

int pipe_init(isscabac_engine *e) {
  if (e->h2d_stream) return ISSCABAC_OK;
  HIP_TRY(e, hipStreamCreateWithFlags(&e->h2d_stream, hipStreamNonBlocking));
  HIP_TRY(e, hipStreamCreateWithFlags(&e->d2h_stream, hipStreamNonBlocking));
  for (hipEvent_t &ev : e->pipe_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int i = 0; i < isscabac_engine::kPinSlots; i++) {
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void **>(&e->pin[i]), isscabac_engine::kPinBytes, hipHostMallocDefault));
    HIP_TRY(e, hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming));
  }
  return ISSCABAC_OK;
}
}  // namespace isscabac

extern "C" {
namespace {

// streams per group: about 64 MiB of bins (and as much ctx) per group, at least 4 groups for a big batch
uint64_t group_streams(uint64_t S, uint32_t pitch) {
  const uint64_t by_bytes = std::max<uint64_t>(1, (64ull << 20) / pitch);
  return std::max<uint64_t>(64, std::min(by_bytes, (S + 3) / 4)) / 64 * 64 + (S < 64 ? S : 0);
}

}  // namespace

// packed: `bins` holds one BIT per bin (bin i of a stream at bit i % 8 of byte i / 8 of its row of pitch / 8 bytes): an
// eighth of the bins' bytes crosses the link and is spread to bytes on the device (k_unpack_bits) in front of the coder
static int encode_host(isscabac_engine *e, const uint8_t *bins, const uint8_t *ctx, const uint32_t *nbins_per_stream,
                       uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                       uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes, bool packed) {
  const char *who = packed ? "isscabac_encode_batch_packed" : "isscabac_encode_batch";  // the entry point error texts name
  BatchShape sh;
  int rc = check_shape(e, who, S, nbins, pitch, ctx_init, nctx, &sh);
  if (rc) return rc;
  if (!bins || !ctx || !payload || !offsets) return fail(e, ISSCABAC_ERR_ARG, "%s: null buffer", who);
  ENGINE_ENTER(e);
  if ((rc = pipe_init(e))) return rc;
  const uint64_t in_bytes = S * pitch;
  // worst-case payload on device so that the gather can never overflow; copy back only what is used
  const uint64_t dev_cap = S * isscabac_stream_bound(nbins), stride = slot_stride_for(nbins);
  if ((rc = grow(e, &e->stage[0], &e->stage_bytes[0], in_bytes))) return rc;
  if ((rc = grow(e, &e->stage[1], &e->stage_bytes[1], in_bytes))) return rc;
  if ((rc = grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->stage[3], &e->stage_bytes[3], dev_cap))) return rc;
  if ((rc = grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  if (packed && (rc = grow(e, &e->stage[6], &e->stage_bytes[6], in_bytes / 8))) return rc;
  if ((rc = grow(e, reinterpret_cast<void **>(&e->slots), &e->slots_bytes, S * stride))) return rc;
  if ((rc = grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->scan_tmp, &e->scan_tmp_bytes, scan_tmp_bytes(S)))) return rc;
  uint8_t *d_bins = static_cast<uint8_t *>(e->stage[0]), *d_ctx = static_cast<uint8_t *>(e->stage[1]);
  uint32_t *d_nb = nbins_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr;
  uint8_t *d_payload = static_cast<uint8_t *>(e->stage[3]);
  uint64_t *d_offsets = static_cast<uint64_t *>(e->stage[4]);
  // everything enqueued earlier on the engine's stream may still read the staging buffers -- and an asynchronous device
  // encode still outstanding has its size word in e->h_total, which this call is about to reuse: settle it first (its
  // capacity error, if any, is this call's result: the caller never synchronised it)
  if ((rc = isscabac_sync(e, nullptr))) return rc;
  HostPipe up{e, 0, isscabac_engine::kPinSlots};
  if (d_nb && (rc = up.upload(d_nb, nbins_per_stream, S * sizeof(uint32_t)))) return rc;
  // groups of streams: group g is coded while group g + 1 is still crossing the link
  const uint64_t G = group_streams(S, pitch);
  for (uint64_t s0 = 0; s0 < S; s0 += G) {
    const uint64_t n = std::min(G, S - s0);
    uint8_t *d_bits = static_cast<uint8_t *>(e->stage[6]);
    if (packed) rc = up.upload(d_bits + s0 * (pitch / 8), bins + s0 * (pitch / 8), n * (pitch / 8));
    else rc = up.upload(d_bins + s0 * pitch, bins + s0 * pitch, n * pitch);
    if (rc) return rc;
    if ((rc = up.upload(d_ctx + s0 * pitch, ctx + s0 * pitch, n * pitch))) return rc;
    HIP_TRY(e, hipEventRecord(e->pipe_ev[0], e->h2d_stream));
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->pipe_ev[0], 0));
    BatchShape g = sh;
    g.S = n;
    if (packed) HIP_TRY(e, launch_unpack_bits(d_bits + s0 * (pitch / 8), d_bins + s0 * pitch, n, pitch, e->stream));
    HIP_TRY(e, launch_encode(g, d_bins + s0 * pitch, d_ctx + s0 * pitch, d_nb ? d_nb + s0 : nullptr, e->slots + s0 * stride, stride,
                             e->lens + s0, e->stream));
    e->last_instance[0] = g.instance;
  }
  HIP_TRY(e, launch_scan_gather(e->slots, stride, e->lens, S, d_offsets, e->scan_tmp, d_payload, dev_cap, e->stream));
  if ((rc = ensure_words(e))) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->h_total, d_offsets + S, sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const uint64_t total = *e->h_total;
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap)
    return fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_batch: payload needs %llu bytes, capacity is %llu",
                (unsigned long long)total, (unsigned long long)payload_cap);
  HostPipe down{e, 0, isscabac_engine::kPinSlots};  // (the uploads are done: the whole ring is free)
  if ((rc = down.download(payload, d_payload, total))) return rc;
  if ((rc = down.download(offsets, d_offsets, (S + 1) * sizeof(uint64_t)))) return rc;
  return down.drain();
}

int isscabac_encode_batch(isscabac_engine *e, const uint8_t *bins, const uint8_t *ctx, const uint32_t *nbins_per_stream,
                          uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                          uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes) {
  return encode_host(e, bins, ctx, nbins_per_stream, S, nbins, pitch, ctx_init, nctx, payload, payload_cap, offsets, total_bytes, false);
}
int isscabac_encode_batch_packed(isscabac_engine *e, const uint8_t *bins_bits, const uint8_t *ctx, const uint32_t *nbins_per_stream,
                                 uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                 uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes) {
  return encode_host(e, bins_bits, ctx, nbins_per_stream, S, nbins, pitch, ctx_init, nctx, payload, payload_cap, offsets, total_bytes, true);
}

// packed: the decoded bins come back one BIT per bin (rows of pitch / 8 bytes, see encode_host)
static int decode_host(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                       const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                       const uint8_t *ctx_init, int nctx, uint8_t *bins_out, uint32_t *status, bool packed) {
  const char *who = packed ? "isscabac_decode_batch_packed" : "isscabac_decode_batch";
  BatchShape sh;
  int rc = check_shape(e, who, S, nbins, pitch, ctx_init, nctx, &sh);
  if (rc) return rc;
  if (!payload || !offsets || !ctx || !bins_out) return fail(e, ISSCABAC_ERR_ARG, "%s: null buffer", who);
  for (uint64_t s = 0; s < S; s++)
    if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] > 0xFFFFFFFFull)
      return fail(e, ISSCABAC_ERR_ARG, "%s: offsets not monotone at stream %llu", who, (unsigned long long)s);
  ENGINE_ENTER(e);
  if ((rc = pipe_init(e))) return rc;
  const uint64_t in_bytes = S * pitch, total = offsets[S];
  if ((rc = grow(e, &e->stage[0], &e->stage_bytes[0], in_bytes))) return rc;  // decoded bins
  if ((rc = grow(e, &e->stage[1], &e->stage_bytes[1], in_bytes))) return rc;  // ctx
  if ((rc = grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = grow(e, &e->stage[3], &e->stage_bytes[3], total + 16))) return rc;
  if ((rc = grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = grow(e, &e->stage[5], &e->stage_bytes[5], S * sizeof(uint32_t)))) return rc;
  if (packed && (rc = grow(e, &e->stage[6], &e->stage_bytes[6], in_bytes / 8))) return rc;
  if ((rc = ensure_words(e))) return rc;
  uint8_t *d_out = static_cast<uint8_t *>(e->stage[0]), *d_ctx = static_cast<uint8_t *>(e->stage[1]);
  uint32_t *d_nb = nbins_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr;
  uint8_t *d_payload = static_cast<uint8_t *>(e->stage[3]);
  uint64_t *d_offsets = static_cast<uint64_t *>(e->stage[4]);
  uint32_t *d_status = static_cast<uint32_t *>(e->stage[5]);
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  // half of the ring feeds the device, the other half brings the bins back: both directions run at once
  const int half = isscabac_engine::kPinSlots / 2;
  HostPipe up{e, 0, half}, down{e, half, isscabac_engine::kPinSlots - half};
  if ((rc = up.upload(d_payload, payload, total))) return rc;
  if ((rc = up.upload(d_offsets, offsets, (S + 1) * sizeof(uint64_t)))) return rc;
  if (d_nb && (rc = up.upload(d_nb, nbins_per_stream, S * sizeof(uint32_t)))) return rc;
  HIP_TRY(e, hipMemsetAsync(e->flag + 3, 0, sizeof(uint32_t), e->stream));
  const uint64_t G = group_streams(S, pitch);
  for (uint64_t s0 = 0; s0 < S; s0 += G) {
    const uint64_t n = std::min(G, S - s0);
    if ((rc = up.upload(d_ctx + s0 * pitch, ctx + s0 * pitch, n * pitch))) return rc;
    HIP_TRY(e, hipEventRecord(e->pipe_ev[0], e->h2d_stream));
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->pipe_ev[0], 0));
    BatchShape g = sh;
    g.S = n;
    // the kernel writes a row up to nbins rounded up to its 128-bin block (clipped to the pitch); with a pitch beyond that,
    // the rest of the staging rows -- scratch that earlier calls used for their own bins -- would travel back as it is
    if ((uint64_t)(nbins + 127u) / 128u * 128u < pitch) HIP_TRY(e, hipMemsetAsync(d_out + s0 * pitch, 0, n * pitch, e->stream));
    // (offsets are positions in the whole payload: the group's kernels get the array from its first stream on)
    HIP_TRY(e, launch_decode(g, d_payload, total, d_offsets + s0, d_ctx + s0 * pitch, d_nb ? d_nb + s0 : nullptr, d_out + s0 * pitch,
                             d_status + s0, e->flag + 3, e->stream));
    e->last_instance[1] = g.instance;
    // streams that start with 0xFF (corrupt input): redone in the reference's own formulation, see isscabac_decode_batch_dev
    HIP_TRY(e, launch_decode_exact(g, d_payload, total, d_offsets + s0, d_ctx + s0 * pitch, d_nb ? d_nb + s0 : nullptr,
                                   d_out + s0 * pitch, d_status + s0, e->flag + 3, e->stream));
    uint8_t *d_bits = static_cast<uint8_t *>(e->stage[6]);
    if (packed) HIP_TRY(e, launch_pack_bits(d_out + s0 * pitch, d_bits + s0 * (pitch / 8), n, pitch, e->stream));
    HIP_TRY(e, hipEventRecord(e->pipe_ev[1], e->stream));
    HIP_TRY(e, hipStreamWaitEvent(e->d2h_stream, e->pipe_ev[1], 0));
    if (packed) rc = down.download(bins_out + s0 * (pitch / 8), d_bits + s0 * (pitch / 8), n * (pitch / 8));
    else rc = down.download(bins_out + s0 * pitch, d_out + s0 * pitch, n * pitch);
    if (rc) return rc;
  }
  if (status && (rc = down.download(status, d_status, S * sizeof(uint32_t)))) return rc;
  if ((rc = down.drain())) return rc;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_decode_batch(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                          const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                          const uint8_t *ctx_init, int nctx, uint8_t *bins_out, uint32_t *status) {
  return decode_host(e, payload, offsets, ctx, nbins_per_stream, S, nbins, pitch, ctx_init, nctx, bins_out, status, false);
}
int isscabac_decode_batch_packed(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                                 const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                 const uint8_t *ctx_init, int nctx, uint8_t *bins_bits_out, uint32_t *status) {
  return decode_host(e, payload, offsets, ctx, nbins_per_stream, S, nbins, pitch, ctx_init, nctx, bins_bits_out, status, true);
}

// ---- measurement hooks, utilities -----------------------------------------------------------------------
int isscabac_set_profiling(isscabac_engine *e, int enabled) {
  if (!e) return ISSCABAC_ERR_ARG;
  e->profiling = enabled != 0;
  return ISSCABAC_OK;
}

int isscabac_last_kernel_ms(isscabac_engine *e, float ms[4]) {
  if (!e || !ms) return ISSCABAC_ERR_ARG;
  if (e->prof_count) {
    ENGINE_ENTER(e);
    const int rc = engine_sync(e);
    if (rc) return rc;
  }
  for (int i = 0; i < 4; i++) ms[i] = e->last_ms[i];
  return ISSCABAC_OK;
}

int isscabac_kernel_ms_mean(isscabac_engine *e, float ms[4], uint32_t launches[4]) {
  if (!e || !ms) return ISSCABAC_ERR_ARG;
  if (e->prof_count) {
    ENGINE_ENTER(e);
    const int rc = engine_sync(e);
    if (rc) return rc;
  }
  for (int i = 0; i < 4; i++) {
    ms[i] = e->sum_n[i] ? (float)(e->sum_ms[i] / e->sum_n[i]) : 0.f;
    if (launches) launches[i] = e->sum_n[i];
    e->sum_ms[i] = 0;
    e->sum_n[i] = 0;
  }
  return ISSCABAC_OK;
}

int isscabac_kernel_ms_samples(isscabac_engine *e, int slot, float *ms, uint32_t cap, uint32_t *count) {
  if (!e || slot < 0 || slot > 3 || !count || (cap && !ms)) return ISSCABAC_ERR_ARG;
  if (e->prof_count) {
    ENGINE_ENTER(e);
    const int rc = engine_sync(e);
    if (rc) return rc;
  }
  std::vector<float> &v = e->samples[slot];
  *count = (uint32_t)v.size();
  for (uint32_t i = 0; i < cap && i < v.size(); i++) ms[i] = v[i];
  if (cap >= v.size()) v.clear();  // a sizing call (cap too small) leaves them in place
  return ISSCABAC_OK;
}

int isscabac_kernel_ms_samples_dropped(isscabac_engine *e, int slot, uint64_t *dropped) {
  if (!e || slot < 0 || slot > 3 || !dropped) return ISSCABAC_ERR_ARG;
  *dropped = e->samples_dropped[slot];
  e->samples_dropped[slot] = 0;
  return ISSCABAC_OK;
}

int isscabac_synth_fill_dev(isscabac_engine *e, uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins, uint32_t pitch,
                            uint32_t nctx, int ep_half, uint8_t *d_bins, uint8_t *d_ctx) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!d_bins || !d_ctx || S == 0 || pitch == 0 || (pitch & 15u) || nbins > pitch || nctx < 1 || nctx > ISSCABAC_MAX_CTX ||
      misaligned(d_bins) || misaligned(d_ctx))
    return fail(e, ISSCABAC_ERR_ARG, "isscabac_synth_fill_dev: bad argument");
  ENGINE_ENTER(e);
  HIP_TRY(e, launch_synth(seed, s0, S, nbins, pitch, nctx, ep_half, d_bins, d_ctx, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_dev_alloc(isscabac_engine *e, uint64_t bytes, void **d_ptr) {
  if (!e || !d_ptr) return ISSCABAC_ERR_ARG;
  ENGINE_ENTER(e);
  HIP_TRY(e, hipMalloc(d_ptr, bytes ? bytes : 16));
  return ISSCABAC_OK;
}

int isscabac_dev_free(isscabac_engine *e, void *d_ptr) {
  if (!e) return ISSCABAC_ERR_ARG;
  ENGINE_ENTER(e);
  HIP_TRY(e, hipFree(d_ptr));
  return ISSCABAC_OK;
}

int isscabac_copy_h2d(isscabac_engine *e, void *d_dst, const void *h_src, uint64_t bytes) {
  if (!e) return ISSCABAC_ERR_ARG;
  ENGINE_ENTER(e);
  HIP_TRY(e, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_copy_d2h(isscabac_engine *e, void *h_dst, const void *d_src, uint64_t bytes) {
  if (!e) return ISSCABAC_ERR_ARG;
  ENGINE_ENTER(e);
  HIP_TRY(e, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

}  // extern "C"
