/*
 * isscabac.h -- C ABI of the MI355X-native batched CABAC engine (libisscabac_hip.so).
 *
 * This is the drop-in boundary for the hot path of christianrohlfing/ISScabac: everything below
 * the reference's MEX gateway (CABAC/SimpleCABACMex.cpp) that touches bins -- the HM-derived
 * CABAC_ArithmeticEncoder / CABAC_ArithmeticDecoder, ContextModel, CABAC_BitstreamFile and the
 * context-init half of CABAC_ContextModels -- re-built as HIP kernels for gfx950.  Plain pointers and
 * sizes only; no torch / MATLAB types.  A thin MEX shim (matlab/SimpleCABACMex.cpp) or any FFI
 * (ctypes: isscabac_amd/engine.py) binds exactly these symbols; see INTEGRATION.md.
 *
 * Citations "ref:" are relative to the reference checkout (/root/reference).
 *
 * Conventions
 *   - A *stream* is one independent CABAC bitstream (the reference: one `class CABAC` handle + one
 *     file, ref: CABAC/SimpleCABACMex.cpp:69-80).  A *batch* is S streams coded in one call.
 *   - bins / ctx arrays are stream-major: element i of stream s lives at [s * pitch + i]
 *     (a MATLAB B x S uint8 matrix, column-major, is exactly this with pitch = B).
 *     pitch must be a multiple of 16, at most ISSCABAC_MAX_PITCH, and the base pointers 16-byte aligned
 *     (the kernels address a wave's 64 rows with 32-bit offsets; one longer stream goes through a stream handle).
 *   - bins[i] is 0 or 1 (the reference's MEX refuses anything else, ref: CABAC/SimpleCABACMex.cpp:221-228); other
 *     values are not checked in the batch calls and give an unspecified (but memory-safe) result for that stream.
 *   - ctx[i] = context index 0..nctx-1, or ISSCABAC_BYPASS (0xFF) for an equiprobable bin; any other value is
 *     invalid: the result for that stream is unspecified, the other streams of the batch are not affected
 *     (ref: encodeBinEP, CABAC/CABAC_ArithmeticEncoder.cpp:250-270 -- not reachable through the
 *     reference MEX; additive).
 *   - ctx_init: nctx packed context states, byte = (state << 1) | mps, i.e. ContextModel::m_ucState
 *     (ref: CABAC/ContextModel.h:78-80,101); the same initial states are used for every stream of the
 *     batch, as one reference handle would.  state must be 0..62.
 *   - payload: tight concatenation of the S bitstreams; stream s occupies
 *     [offsets[s], offsets[s+1]) -- byte-identical to the file the reference writes for that stream.
 *   - all calls are synchronous on the given HIP stream: they return after the work has completed
 *     unless stated otherwise.  A handle is not thread safe; distinct handles are independent.
 *   - return value: 0 = ok, otherwise an ISSCABAC_ERR_* code; isscabac_last_error() has the text.
 */
#ifndef ISSCABAC_H
#define ISSCABAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISSCABAC_BYPASS 0xFFu
#define ISSCABAC_MAX_CTX 255 /* contexts of a batch call (one byte per bin names the context) */
/* The reference holds up to 999 context models per handle (ref: CABAC/CommonDef.h:56, CABAC_ContextModelsInit.cpp:53,84).
 * The single-stream handle (isscabac_stream_*, what the reference's per-bin commands map to) takes as many; an 8-bit batch
 * call with more than ISSCABAC_MAX_CTX contexts is refused with ISSCABAC_ERR_ARG (its context indices are bytes) ... */
#define ISSCABAC_STREAM_MAX_CTX 999
/* ... and so do the batch calls with 16-bit context indices (isscabac_*_batch16*): 0xFFFF = bypass bin */
#define ISSCABAC_MAX_CTX16 999
#define ISSCABAC_BYPASS16 0xFFFFu
/* Fixed-probability bins (ref: encodeBinProb / decodeBinProb, CABAC_ArithmeticEncoder.cpp:180-243,
 * CABAC_ArithmeticDecoder.cpp:192-286; compiled into the reference only with RWTH_CABAC_FIXED_PROBABILITY, CommonDef.h:50).
 * In a batch call of at most ISSCABAC_PROB_MAX_CTX contexts a ctx byte of ISSCABAC_PROB(p) codes the bin with a fixed
 * probability of p percent of being 1, no context model involved; p = 50 is a bypass bin, as in the reference.
 * p = 46..49 and 51..54 are not supported (see cabac_lane.h: prob_valid): like any invalid ctx byte they leave that
 * stream's result unspecified.  With more contexts, bytes 129..254 are context indices. */
#define ISSCABAC_PROB_BASE 155u
#define ISSCABAC_PROB(p) (ISSCABAC_PROB_BASE + (p)) /* p = 1..99 -> 156..254 */
#define ISSCABAC_PROB_MAX_CTX 128
#define ISSCABAC_MAX_PITCH (1u << 25) /* bins per stream in one batch call */

enum {
  ISSCABAC_OK = 0,
  ISSCABAC_ERR_ARG = 1,      /* bad argument (null, alignment, range) */
  ISSCABAC_ERR_HIP = 2,      /* HIP runtime error (no GPU, launch failure, ...) */
  ISSCABAC_ERR_NOMEM = 3,    /* device allocation failed */
  ISSCABAC_ERR_CAPACITY = 4, /* caller's payload buffer too small; *total_bytes holds the need */
  ISSCABAC_ERR_STATE = 5,    /* call sequence error on a single-stream handle */
  ISSCABAC_ERR_IO = 6        /* a file could not be opened, read or written (container calls) */
};

/* decoder integrity flags (per stream), the two asserts of ref: CABAC_ArithmeticDecoder.cpp:78-81
 * turned into data, plus an end-of-payload over-read flag */
#define ISSCABAC_ST_NO_TERMINATE 1u
#define ISSCABAC_ST_BAD_STOP_BIT 2u
#define ISSCABAC_ST_OVERREAD 4u

typedef struct isscabac_engine isscabac_engine;

/* ---- lifetime -------------------------------------------------------------------------------
 * replaces `new CABAC` / never-freed handle of ref: SimpleCABACMex.cpp:149-156 */
int isscabac_create(int device, isscabac_engine **out);
void isscabac_destroy(isscabac_engine *e);
/* e may be NULL: returns the text of the last error of a failed isscabac_create on this thread */
const char *isscabac_last_error(const isscabac_engine *e);
int isscabac_abi_version(void);
/* Which instance of the batch coder kernels runs (results are bit-identical, only the schedule differs):
 *   AUTO        by launch size: the latency-oriented instance when the launch leaves at most two waves on a SIMD
 *               (up to 512 streams per compute unit: 128 Ki streams on an MI355X), else the throughput one
 *   THROUGHPUT  fewest instructions per bin; relies on four resident waves per SIMD to hide LDS latency
 *   LATENCY     state prefetch + register forwarding, no LDS round trip on the range recurrence; for small batches.
 *               With <= 64 contexts the encoder runs as a producer / consumer wave pair and the decoder with a helper
 *               wave, for launches of at most eight waves of streams per compute unit (128 Ki streams on an MI355X: a
 *               strong-scaled job's per-GPU share)
 *   LATENCY_SINGLE  the latency-oriented instances on ONE wave per 64 streams whatever the launch size (what LATENCY
 *               runs with more than 64 contexts); lets tests and A/B runs reach those instances with small batches
 * (a strong-scaled job leaves each GPU 1/N of the streams: SURVEY.md 8(e)) */
#define ISSCABAC_TUNE_AUTO 0
#define ISSCABAC_TUNE_THROUGHPUT 1
#define ISSCABAC_TUNE_LATENCY 2
#define ISSCABAC_TUNE_LATENCY_SINGLE 3
int isscabac_set_tuning(isscabac_engine *e, int mode);
/* the kernel instance the engine's most recent 8-bit batch call launched: which = 0 encode ("k_encode_pair",
 * "k_encode<1,false,false>" = <context-image size class, latency-oriented, length-bucketed>, ...), 1 decode.  A string
 * owned by the library, "" before the first call.  For tests, A/B runs and the bench line; results never depend on it */
const char *isscabac_last_instance(const isscabac_engine *e, int which);
/* use this HIP stream (hipStream_t) for all work of the engine; NULL = the engine's own stream */
int isscabac_set_stream(isscabac_engine *e, void *hip_stream);

/* ---- context initialisation (host side, pure functions) --------------------------------------
 * ref: CABAC_ContextModels::initContextModelsByP0Prob / xMapProbabilityToState,
 *      CABAC/CABAC_ContextModelsInit.cpp:82-148 */
int isscabac_init_by_prob(const double *p0, int nctx, uint8_t *packed_out);
/* ref: initContextModelsByMpsState, CABAC_ContextModelsInit.cpp:51-80: flat triples
 * [ctxIdx, mps, state]; ctxIdx is ignored exactly as the reference ignores it */
int isscabac_init_by_state(const double *triples, int nctx, uint8_t *packed_out);

/* ---- sizing ------------------------------------------------------------------------------------ */
/* provable upper bound of one stream's bitstream length: ceil((6*nbins + 9) / 8) */
uint64_t isscabac_stream_bound(uint32_t nbins);

/* ---- batch coding on device-resident buffers (the hot path) ---------------------------------------
 * replaces the per-bin loop  encodeBin -> CABAC_ArithmeticEncoder::encodeBin -> writeOut ->
 * CABAC_BitstreamFile::write  (ref: SimpleCABACMex.cpp:210-247, CABAC_ArithmeticEncoder.cpp:113-178,
 * 369-412, CABAC_BitstreamFile.cpp:95-138) and encodeFinish (ref: :70-105) for S streams at once.
 *
 *   d_bins, d_ctx : device, [S * pitch] bytes each
 *   d_nbins       : device, S x u32 bins actually coded per stream (<= pitch), or NULL => nbins
 *   d_payload     : device, payload_cap bytes
 *   d_offsets     : device, (S + 1) x u64
 *   total_bytes   : host, receives offsets[S] (may be NULL)
 */
int isscabac_encode_batch_dev(isscabac_engine *e, const uint8_t *d_bins, const uint8_t *d_ctx,
                              const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                              const uint8_t *ctx_init, int nctx, uint8_t *d_payload,
                              uint64_t payload_cap, uint64_t *d_offsets, uint64_t *total_bytes);

/* replaces decodeStart / decodeBin / decodeFinish (ref: SimpleCABACMex.cpp:280-354,
 * CABAC_ArithmeticDecoder.cpp:54-190,288-331,423-472) for S streams at once.
 *   d_bins_out : device, [S * pitch] bytes; bins beyond a stream's count are written as 0 -- up to nbins rounded up to a
 *                multiple of 128 (or the pitch, if smaller): row bytes behind that are not written by the device calls
 *                (the caller's buffer keeps what it held) and are zero in the host-buffer calls (bytes or bits)
 *   d_status   : device, S x u32 ISSCABAC_ST_* flags, or NULL
 *   payload_bytes : offsets[S] (size of the valid part of d_payload) */
int isscabac_decode_batch_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes,
                              const uint64_t *d_offsets, const uint8_t *d_ctx,
                              const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                              const uint8_t *ctx_init, int nctx, uint8_t *d_bins_out,
                              uint32_t *d_status);

/* ---- the same two calls without the host in the loop ---------------------------------------------------
 * The *_async forms only enqueue work on the engine's stream and return: nothing is read back, the host does not wait.
 * A caller that chains encode -> decode (or many batches) keeps the GPU busy back to back, independent of host wake-up
 * latency.  isscabac_sync waits for everything enqueued so far and reports what the synchronous forms report at once:
 *   - *total_bytes (may be NULL) = offsets[S] of the most recent encode; ISSCABAC_ERR_CAPACITY if that exceeded its
 *     payload_cap (streams that did not fit were skipped by the gather, the offsets are complete);
 *   - decode: payload_bytes may be any upper bound of offsets[S] that the buffer really has (e.g. its capacity) --
 *     it only bounds memory accesses; streams are delimited by d_offsets.
 * Buffers must stay valid until isscabac_sync (or another synchronising call on this engine) has returned.  The
 * synchronous forms above are the async form + isscabac_sync. */
int isscabac_encode_batch_dev_async(isscabac_engine *e, const uint8_t *d_bins, const uint8_t *d_ctx,
                                    const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                    const uint8_t *ctx_init, int nctx, uint8_t *d_payload,
                                    uint64_t payload_cap, uint64_t *d_offsets);
int isscabac_decode_batch_dev_async(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes,
                                    const uint64_t *d_offsets, const uint8_t *d_ctx, const uint32_t *d_nbins,
                                    uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                    uint8_t *d_bins_out, uint32_t *d_status);
int isscabac_sync(isscabac_engine *e, uint64_t *total_bytes);

/* ---- batches of MORE than 255 contexts: 16-bit context indices ---------------------------------------
 * The reference's coder holds up to 999 context models (ref: CABAC/CommonDef.h:56, CABAC_ContextModelsInit.cpp:53,84) and
 * its per-bin commands take the context index as a double (SimpleCABACMex.cpp:223-224, 316).  These calls are the batch
 * form of that: as isscabac_encode_batch_dev / isscabac_decode_batch_dev, with d_ctx16 / ctx16 = [S * pitch] uint16
 * context indices (16-byte aligned), ISSCABAC_BYPASS16 (0xFFFF) = bypass bin, 1 <= nctx <= ISSCABAC_MAX_CTX16.  An index
 * in [nctx, 0xFFFF) is invalid: that stream's result is unspecified (memory-safe), the others are not affected.  No
 * fixed-probability bins here.  The state image (256 bytes per four contexts and 64 streams) lives in the LDS of a one-wave
 * workgroup: 270 Gbin/s per direction at 999 contexts, 450-670 at 256, against 1 400 for up to 64 contexts -- use the 8-bit
 * calls wherever they apply.  Synchronous. */
int isscabac_encode_batch16_dev(isscabac_engine *e, const uint8_t *d_bins, const uint16_t *d_ctx16, const uint32_t *d_nbins,
                                uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint64_t *total_bytes);
int isscabac_decode_batch16_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                                const uint16_t *d_ctx16, const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                const uint8_t *ctx_init, int nctx, uint8_t *d_bins_out, uint32_t *d_status);
/* the same on host buffers (plain copies around the device forms) */
int isscabac_encode_batch16(isscabac_engine *e, const uint8_t *bins, const uint16_t *ctx16, const uint32_t *nbins_per_stream,
                            uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                            uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes);
int isscabac_decode_batch16(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint16_t *ctx16,
                            const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init,
                            int nctx, uint8_t *bins_out, uint32_t *status);

/* ---- batch coding on host buffers (what a MEX / FFI caller holding host arrays uses) --------------
 * same semantics; the engine stages through its own device buffers (H2D, kernels, D2H).
 * payload_cap in bytes; on ISSCABAC_ERR_CAPACITY nothing is written but *total_bytes is set. */
int isscabac_encode_batch(isscabac_engine *e, const uint8_t *bins, const uint8_t *ctx,
                          const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins,
                          uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                          uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes);
int isscabac_decode_batch(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets,
                          const uint8_t *ctx, const uint32_t *nbins_per_stream, uint64_t S,
                          uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                          uint8_t *bins_out, uint32_t *status);

/* The same two calls with the bins as BITS on the caller's side -- an additive form for host callers, whose cost is the
 * PCIe link (ref: the bins the reference's callers pass are one value per bin, CABAC/SimpleCABACMex.cpp:210-247, 301-335):
 * row s of bins_bits / bins_bits_out is pitch / 8 bytes long and holds bin i of stream s at bit i % 8 of byte i / 8
 * (numpy.packbits(..., bitorder="little")).  ctx stays one byte per bin.  An eighth of the bins' bytes crosses the link;
 * they are spread to / gathered from bytes on the device.  Results are those of the byte forms. */
int isscabac_encode_batch_packed(isscabac_engine *e, const uint8_t *bins_bits, const uint8_t *ctx,
                                 const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins,
                                 uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                 uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes);
int isscabac_decode_batch_packed(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets,
                                 const uint8_t *ctx, const uint32_t *nbins_per_stream, uint64_t S,
                                 uint32_t nbins, uint32_t pitch, const uint8_t *ctx_init, int nctx,
                                 uint8_t *bins_bits_out, uint32_t *status);

/* ---- symbol-level batch coding: binariser + context selector fused into the coder kernels -----------------------
 * A stream is a column of integer symbols 0..nq-1 (uint8, stream-major [S][pitch] like the bin arrays).  The kernel
 * binarises (TU: ref CABAC/cabacBinarizer.m:30-37; EGk: :56-69), picks the context of every bin
 *   ISSCABAC_SEL_DEMO: 3 contexts from the first bin of the previous symbol   (ref: CABAC/cabacDemo.m:113-121)
 *   ISSCABAC_SEL_ISS : 7*nlbp+2 contexts from the bins of this symbol and of the symbol one row up
 *                      (ref: ISS/+coder/cabacContextSelection.m:24-67; `iss_types` = 1 cond0 | 2 cond1 | 4 conds0 |
 *                      8 conds1 | 16 condbinlft; the reference default is 15 with nlbp = 3, ISS/ISS.m:50-52)
 * and codes it; the decoder runs the same selector on the bins it has just decoded, detects the end of each symbol
 * (ref: CABAC/cabacDecodeSymbolFinished.m:12-30) and de-binarises (ref: CABAC/cabacDebinarizer.m:28-31,46-53).
 * One stream here == one column walked like the inner loops of ISS/+coder/cabacEncode.m:45-70 / cabacDecode.m:29-55
 * (the symbol "one row up" is the previous symbol of the same stream; the first symbol has none).
 * Limits: nctx <= 64, at most 32 bins per symbol (TU: nq <= 33). */
#define ISSCABAC_BIN_TU 0u
#define ISSCABAC_BIN_EGK 1u
/* encode side only -- the reference has no end-of-symbol rule for them (ref: CABAC/cabacDecodeSymbolFinished.m:12-30), so
 * isscabac_decode_symbols* refuses them: */
#define ISSCABAC_BIN_TRK 2u  /* truncated Rice, order k; suffix all ones from the maximum on (ref: CABAC/cabacBinarizer.m:39-54) */
#define ISSCABAC_BIN_FL32 3u /* 32 bins, MSB first (ref: CABAC/cabacBinarizer.m:71-75, 'DEC2FL32') */
#define ISSCABAC_SEL_DEMO 0u
#define ISSCABAC_SEL_ISS 1u
typedef struct {
  uint32_t binarization; /* ISSCABAC_BIN_* */
  uint32_t k;            /* EG order */
  uint32_t nq;           /* symbols are 0..nq-1 */
  uint32_t selector;     /* ISSCABAC_SEL_* */
  uint32_t iss_types;
  uint32_t nlbp;
  uint32_t col_rows;     /* 0: each stream is one column.  > 0: each stream is a whole matrix stored column after column
                            with col_rows rows; the "one row up" neighbour restarts at every column while coder and
                            context states carry on -- the reference's one-file-per-matrix mode,
                            ref: ISS/+coder/cabacEncode.m:45-70 */
} isscabac_symbol_params;

/* d_sym: device [S * pitch] uint8; d_nsym: per-stream symbol counts or NULL => nsym; d_nbins_out: optional S x u32,
 * bins coded per stream.  Outputs as isscabac_encode_batch_dev. */
int isscabac_encode_symbols_dev(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S,
                                uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                const uint8_t *ctx_init, int nctx, uint8_t *d_payload, uint64_t payload_cap,
                                uint64_t *d_offsets, uint64_t *total_bytes, uint32_t *d_nbins_out);
int isscabac_decode_symbols_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes,
                                const uint64_t *d_offsets, const uint32_t *d_nsym, uint64_t S, uint32_t nsym,
                                uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init,
                                int nctx, uint8_t *d_sym_out, uint32_t *d_status);
/* The encoder-side debugging statistics of +coder/cabacEncode.m (ref: ISS/+coder/cabacEncode.m:39-42,60-67), per
 * stream, all optional (NULL = not wanted):
 *   ctx_hist[s * ISSCABAC_STAT_CTX + c]  bins coded with context c                              (ctxHist)
 *   ctx_cost[s * ISSCABAC_STAT_CTX + c]  growth of getNumBits() while coding them               (ctxCost)
 *   sym_bits[s * pitch + i]              growth of getNumBits() over symbol i, the heat map     (H)
 * getNumBits() counts bits already written to the file (whole bytes, ref: CABAC/CABAC_BitstreamFile.h:70), so the
 * kernel emits bytes at the reference's own cadence (one write-out test per bin), as this symbol path always does. */
#define ISSCABAC_STAT_CTX 64
typedef struct isscabac_symbol_stats {
  uint32_t *ctx_hist;
  uint32_t *ctx_cost;
  uint16_t *sym_bits;
  /* optional state trace: bin_trace[s * bin_trace_pitch + i] = the 4-byte record of bin i of stream s (i < nbins_out[s]);
   * isscabac_trace_stats turns one stream's records into getEncoderStats' two outputs.  bin_trace_pitch (in records)
   * must be at least nsym x isscabac_symbol_max_bins. */
  uint32_t *bin_trace;
  uint32_t bin_trace_pitch;
} isscabac_symbol_stats;
/* Asynchronous forms (as isscabac_encode_batch_dev_async / isscabac_decode_batch_dev_async): the calls only enqueue on the
 * engine's stream, so that symbol batch after symbol batch -- or encode -> decode -- runs with no host wait in between;
 * isscabac_sync() waits and reports the payload size / a capacity overflow of the most recent asynchronous encode (batch or
 * symbol).  payload_bytes of the decode only bounds its memory accesses (the buffer's capacity will do). */
int isscabac_encode_symbols_dev_async(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S,
                                      uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                      const uint8_t *ctx_init, int nctx, uint8_t *d_payload, uint64_t payload_cap,
                                      uint64_t *d_offsets, uint32_t *d_nbins_out);
int isscabac_decode_symbols_dev_async(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes,
                                      const uint64_t *d_offsets, const uint32_t *d_nsym, uint64_t S, uint32_t nsym,
                                      uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init,
                                      int nctx, uint8_t *d_sym_out, uint32_t *d_status);
/* as isscabac_encode_symbols_dev, d_stats->* are device pointers */
int isscabac_encode_symbols_stats_dev(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S,
                                      uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                      const uint8_t *ctx_init, int nctx, uint8_t *d_payload, uint64_t payload_cap,
                                      uint64_t *d_offsets, uint64_t *total_bytes, uint32_t *d_nbins_out,
                                      const isscabac_symbol_stats *d_stats);
/* host-buffer forms (H2D, kernels, D2H inside) */
int isscabac_encode_symbols(isscabac_engine *e, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                            uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                            const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                            uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out);
/* stats->* are host arrays ([S * ISSCABAC_STAT_CTX] x 2 and [S * pitch]) */
int isscabac_encode_symbols_stats(isscabac_engine *e, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                                  uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                  const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                                  uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out,
                                  const isscabac_symbol_stats *stats);
int isscabac_decode_symbols(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets,
                            const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym, uint32_t pitch,
                            const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                            uint8_t *sym_out, uint32_t *status);
/* Empirical context initialisation for the ISS selector: p(0) of every one of the 7*nlbp+2 contexts, counted over the
 * binarised symbol matrix (one column per stream, all columns together) -- a histogram reduction on the device,
 * ratios in double on the host exactly as the reference forms them.
 * ref: ISS/+coder/cabacInitContextModel.m:1-130; p0_bytes_out = uint8(p0 * 255) as ISS/+coder/cabacEncode.m:30-31
 * transmits it (feed p0_bytes_out[c] / 255.0 to isscabac_init_by_prob, as cabacEncode.m:31 / cabacDecode.m:13 do).
 * Either output may be NULL. */
int isscabac_iss_init_context_model_dev(isscabac_engine *e, const uint8_t *d_sym, uint64_t S, uint32_t nsym,
                                        uint32_t pitch, const isscabac_symbol_params *params, double *p0_out,
                                        uint8_t *p0_bytes_out);
int isscabac_iss_init_context_model(isscabac_engine *e, const uint8_t *sym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                    const isscabac_symbol_params *params, double *p0_out, uint8_t *p0_bytes_out);

/* most bins one symbol can binarise to under `params` (0 if the parameters are not supported) */
uint32_t isscabac_symbol_max_bins(const isscabac_symbol_params *params);

/* ---- single-stream handle: the reference's per-bin commands, kept working ---------------------------------
 * One isscabac_stream = one reference handle (`class CABAC`, ref: CABAC/SimpleCABACMex.cpp:69-80): an encoder, a
 * decoder and one set of context models for each (ref: :175-176), all resident on the device, coded by one lane with
 * the reference's exact write-out cadence, so getNumBits between bins reports what the reference reports.  encodeBin
 * queues on the host; calls that need an answer at once (decodeBin, getNumBits behind up to 32 queued bins) are served
 * by a live kernel polling a pinned mailbox -- about 2.3 us per call, against 12.6 us for a launch and a synchronisation
 * (ISSCABAC_STREAM_LIVE=0 switches it off); it leaves after 200 us without requests, after 2 ms of residency whatever the
 * traffic (the next call starts the next one), and at once when anything else is enqueued for the engine.  Work of OTHER
 * engines or other HIP users in the process that the runtime maps onto the same hardware queue can wait behind it for up to
 * those 2 ms.  This is the compatibility path
 * (cabacWrapper.encodeBin / decodeBin one bin at a time, context chosen by the caller from already decoded bins);
 * throughput lives in the batch calls.  A handle is not thread safe and belongs to its engine: the engine's calls and its
 * handles' calls must not run concurrently (they share the engine's stream, and the engine stops its handles' live kernels
 * before it enqueues anything else); handles of different engines are independent.  Close a handle before isscabac_destroy
 * of its engine; a handle that outlives its engine is stopped by the destroy, answers ISSCABAC_ERR_STATE to every call and
 * can only be closed.
 *
 *   SimpleCABACMex('initByProb'|'initByState', fn, ctxInit)  -> isscabac_init_by_* + isscabac_stream_open
 *   ('encodeStart', h)        -> isscabac_stream_encode_start      (ref: :186-209)
 *   ('encodeBin', h, bin, ctx)-> isscabac_stream_encode_bins(n=1)  (ref: :210-247; ctx 0xFF = bypass, additive)
 *   ('getNumBits', h)         -> isscabac_stream_num_bits          (ref: :248-264)
 *   ('encodeFinish', h)       -> isscabac_stream_encode_finish     (ref: :265-279; returns the file's bytes)
 *   ('decodeStart', h)        -> isscabac_stream_decode_start      (ref: :280-300; caller passes the file's bytes)
 *   ('decodeBin', h, ctx)     -> isscabac_stream_decode_bins(n=1)  (ref: :301-335)
 *   ('decodeFinish', h)       -> isscabac_stream_decode_finish     (ref: :336-354; asserts become status flags)
 *   ('getEncoderStats'|'getDecoderStats', h, ctx) -> isscabac_stream_get_stats (ref: :356-466, Windows build only)
 * nctx may be up to ISSCABAC_STREAM_MAX_CTX.  The u8 forms below name contexts 0..254 (0xFF = bypass; in handles of at
 * most ISSCABAC_PROB_MAX_CTX contexts ISSCABAC_PROB(p) = fixed probability); the *_ops forms take 16-bit ops and reach
 * every context. */
#define ISSCABAC_OP_CTX(c) ((uint16_t)(c))                 /* context bin, c < nctx */
#define ISSCABAC_OP_BYPASS ((uint16_t)0xFFFFu)              /* encodeBinEP / decodeBinEP */
#define ISSCABAC_OP_PROB(p) ((uint16_t)(0x8000u | (p)))     /* encodeBinProb / decodeBinProb, p = 1..99 percent */
typedef struct isscabac_stream isscabac_stream;
int isscabac_stream_open(isscabac_engine *e, const uint8_t *ctx_init, int nctx, isscabac_stream **out);
void isscabac_stream_close(isscabac_stream *s);
int isscabac_stream_encode_start(isscabac_stream *s);
/* bins[i] in {0,1}; ctx[i] < nctx or ISSCABAC_BYPASS.  Bins are queued on the host and coded in order before any
 * call that observes the encoder (num_bits, encode_finish). */
int isscabac_stream_encode_bins(isscabac_stream *s, const uint8_t *bins, const uint8_t *ctx, uint32_t n);
int isscabac_stream_num_bits(isscabac_stream *s, uint64_t *bits);
/* terminate + flush; copies the complete bitstream (what the reference leaves in its file) to out; *nbytes = size.
 * ISSCABAC_ERR_CAPACITY if cap is too small (call again with a bigger buffer). */
int isscabac_stream_encode_finish(isscabac_stream *s, uint8_t *out, uint64_t cap, uint64_t *nbytes);
int isscabac_stream_decode_start(isscabac_stream *s, const uint8_t *payload, uint64_t nbytes);
int isscabac_stream_decode_bins(isscabac_stream *s, const uint8_t *ctx, uint32_t n, uint8_t *bins_out);
int isscabac_stream_decode_finish(isscabac_stream *s, uint32_t *status);
/* the same with 16-bit ops (ISSCABAC_OP_*) */
int isscabac_stream_encode_ops(isscabac_stream *s, const uint8_t *bins, const uint16_t *ops, uint32_t n);
int isscabac_stream_decode_ops(isscabac_stream *s, const uint16_t *ops, uint32_t n, uint8_t *bins_out);

/* ---- state traces: getEncoderStats / getDecoderStats of the reference's Windows build ---------------------------------
 * ref: CABAC/SimpleCABACMex.cpp:231-242,316-327 (one step per context bin), :356-466 (the two outputs),
 *      CABAC/ContextModel.cpp:126-134 (state numbering: mps 0 -> 63 - state, mps 1 -> state + 64).
 * Tracing is off by default (4 bytes of device memory per bin); switch it before encodeStart / decodeStart.
 *   trace5      : 5 bytes per step [codedBin, state before, mps before, state after, mps after] -- the 5 x N uint8 matrix;
 *                 on the decoder side codedBin is always 0, as in the reference (its gateway stores the variable before
 *                 decodeBin has written it)
 *   cap_steps   : steps trace5 has room for; *nsteps receives the number of steps there are (call again if larger)
 *   transitions : 128 x 128 uint32 counts [state before][state after], or NULL
 * isscabac_trace_stats is the host-side reduction itself, for trace records obtained elsewhere (the symbol kernels'
 * statistics variant writes the same 4-byte records: isscabac_symbol_stats.bin_trace). */
int isscabac_stream_set_trace(isscabac_stream *s, int enabled);
int isscabac_stream_get_stats(isscabac_stream *s, int decoder, int ctx, uint8_t *trace5, uint64_t cap_steps,
                              uint64_t *nsteps, uint32_t *transitions);
int isscabac_trace_stats(const uint32_t *records, uint64_t nrecords, int ctx, int decoder, uint8_t *trace5,
                         uint64_t cap_steps, uint64_t *nsteps, uint32_t *transitions);

/* ---- host twins of the SERIAL entry points (`cabac_cpu_*`) ------------------------------------------------------
 * The reference's own use of this engine is serial: one bin per MEX call (ref: CABAC/cabacWrapper.m:39-66), or a whole matrix
 * as ONE stream (ref: ISS/+coder/cabacEncode.m:45-70).  Neither has parallelism to give a GPU: a device handle answers a
 * decodeBin in 2.3 us and codes a 400 x 20 matrix in 1.3 ms, one host core needs ~15 ns per bin.  These twins are the engine's
 * own per-lane code (isscabac_amd/csrc/cabac_lane.h, cabac_symbols.h -- the functions the HIP kernels call) compiled for the
 * host: same arguments, same bytes, same getNumBits cadence, same status words, same trace records as their device
 * counterparts, and no engine argument -- they need no GPU.  They are explicitly named entry points, never a fallback: no
 * isscabac_* batch call routes here, and the batch calls still fail with ISSCABAC_ERR_HIP without a device.
 *   isscabac_cpu_stream_*          = isscabac_stream_*            (ref: CABAC/SimpleCABACMex.cpp:186-354, 356-466)
 *   isscabac_cpu_encode_symbols    = isscabac_encode_symbols_stats (stats may be NULL), streams coded one after the other
 *   isscabac_cpu_decode_symbols    = isscabac_decode_symbols
 * matlab/SimpleCABACMex.cpp serves the per-bin commands from these by default (ISSCABAC_MEX_PERBIN=gpu: the device handle). */
typedef struct isscabac_cpu_stream isscabac_cpu_stream;
int isscabac_cpu_stream_open(const uint8_t *ctx_init, int nctx, isscabac_cpu_stream **out);
void isscabac_cpu_stream_close(isscabac_cpu_stream *s);
const char *isscabac_cpu_stream_last_error(const isscabac_cpu_stream *s);
int isscabac_cpu_stream_encode_start(isscabac_cpu_stream *s);
int isscabac_cpu_stream_encode_bins(isscabac_cpu_stream *s, const uint8_t *bins, const uint8_t *ctx, uint32_t n);
int isscabac_cpu_stream_encode_ops(isscabac_cpu_stream *s, const uint8_t *bins, const uint16_t *ops, uint32_t n);
int isscabac_cpu_stream_num_bits(isscabac_cpu_stream *s, uint64_t *bits);
int isscabac_cpu_stream_encode_finish(isscabac_cpu_stream *s, uint8_t *out, uint64_t cap, uint64_t *nbytes);
int isscabac_cpu_stream_decode_start(isscabac_cpu_stream *s, const uint8_t *payload, uint64_t nbytes);
int isscabac_cpu_stream_decode_bins(isscabac_cpu_stream *s, const uint8_t *ctx, uint32_t n, uint8_t *bins_out);
int isscabac_cpu_stream_decode_ops(isscabac_cpu_stream *s, const uint16_t *ops, uint32_t n, uint8_t *bins_out);
int isscabac_cpu_stream_decode_finish(isscabac_cpu_stream *s, uint32_t *status);
int isscabac_cpu_stream_set_trace(isscabac_cpu_stream *s, int enabled);
int isscabac_cpu_stream_get_stats(isscabac_cpu_stream *s, int decoder, int ctx, uint8_t *trace5, uint64_t cap_steps,
                                  uint64_t *nsteps, uint32_t *transitions);
int isscabac_cpu_encode_symbols(const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym, uint32_t pitch,
                                const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out,
                                const isscabac_symbol_stats *stats);
int isscabac_cpu_decode_symbols(const uint8_t *payload, const uint64_t *offsets, const uint32_t *nsym_per_stream, uint64_t S,
                                uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init,
                                int nctx, uint8_t *sym_out, uint32_t *status);

/* ---- several GPUs behind one handle, one process (what a MEX / FFI caller can use) -------------------
 * isscabac_multi owns one engine per listed device.  A batch is split like the multi-process path splits it
 * (isscabac_amd/sharding.py): streams are independent, device g of n codes the contiguous block
 * [S*g/n, S*(g+1)/n) (isscabac_multi_shard), nothing is exchanged while coding, and the only communication is the final
 * bitstream gather.  One host thread per device drives its engine, so the devices work concurrently.  Results are
 * identical to a single engine's.  A device may be listed more than once (two engines sharing it). */
typedef struct isscabac_multi isscabac_multi;
int isscabac_create_multi(const int *device_ids, int n, isscabac_multi **out);
void isscabac_destroy_multi(isscabac_multi *m);
int isscabac_multi_device_count(const isscabac_multi *m);
/* how device g's shard reaches the root (device 0 of the list) in the final gather: 0 = same device, 1 = peer access
 * (hipDeviceCanAccessPeer was true and it is enabled: xGMI / P2P copy), 2 = no peer access, hipMemcpyPeerAsync stages through
 * the host; -1 = bad argument.  ISSCABAC_MULTI_LOG=1 prints one line per pair at isscabac_create_multi */
int isscabac_multi_peer_access(const isscabac_multi *m, int g);
isscabac_engine *isscabac_multi_engine(isscabac_multi *m, int g); /* engine g, e.g. for isscabac_dev_alloc / isscabac_set_tuning */
const char *isscabac_multi_last_error(const isscabac_multi *m);
void isscabac_multi_shard(const isscabac_multi *m, uint64_t S, int g, uint64_t *first, uint64_t *end);
/* host buffers: as isscabac_encode_batch / isscabac_decode_batch */
int isscabac_multi_encode_batch(isscabac_multi *m, const uint8_t *bins, const uint8_t *ctx,
                                const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                                uint64_t *offsets, uint64_t *total_bytes);
int isscabac_multi_decode_batch(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                                const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                const uint8_t *ctx_init, int nctx, uint8_t *bins_out, uint32_t *status);
/* the same with the bins as bits: as isscabac_encode_batch_packed / isscabac_decode_batch_packed */
int isscabac_multi_encode_batch_packed(isscabac_multi *m, const uint8_t *bins_bits, const uint8_t *ctx,
                                       const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins, uint32_t pitch,
                                       const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                                       uint64_t *offsets, uint64_t *total_bytes);
int isscabac_multi_decode_batch_packed(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets,
                                       const uint8_t *ctx, const uint32_t *nbins_per_stream, uint64_t S, uint32_t nbins,
                                       uint32_t pitch, const uint8_t *ctx_init, int nctx, uint8_t *bins_bits_out,
                                       uint32_t *status);
/* symbol-level calls over the shards: as isscabac_encode_symbols[_stats] / isscabac_decode_symbols (host buffers; stats may
 * be NULL).  Streams -- columns, or whole matrices with col_rows -- are split like the bin batches; a single stream is one
 * device's work (ref: the whole-matrix stream of ISS/+coder/cabacEncode.m:45-70 is serial by construction). */
int isscabac_multi_encode_symbols(isscabac_multi *m, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                                  uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                  const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                                  uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out);
int isscabac_multi_encode_symbols_stats(isscabac_multi *m, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                                        uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                        const uint8_t *ctx_init, int nctx, uint8_t *payload, uint64_t payload_cap,
                                        uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out,
                                        const isscabac_symbol_stats *stats);
int isscabac_multi_decode_symbols(isscabac_multi *m, const uint8_t *payload, const uint64_t *offsets,
                                  const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym, uint32_t pitch,
                                  const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                                  uint8_t *sym_out, uint32_t *status);
/* device-resident shards: d_bins[g] / d_ctx[g] (/ d_nbins[g], the array or its entries may be NULL) live on device g and
 * hold that device's block of streams; the payload and the S + 1 offsets are gathered on the FIRST listed device with
 * hipMemcpyPeerAsync, every device pushing its piece straight to the root (no ring). */
int isscabac_multi_encode_batch_dev(isscabac_multi *m, const uint8_t *const *d_bins, const uint8_t *const *d_ctx,
                                    const uint32_t *const *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                    const uint8_t *ctx_init, int nctx, uint8_t *d_payload_root, uint64_t payload_cap,
                                    uint64_t *d_offsets_root, uint64_t *total_bytes);
/* ... and back: payload and offsets on the FIRST listed device (where the call above leaves them), every device pulls its
 * slice from there (hipMemcpyPeerAsync) and decodes its block into its own d_bins_out[g] / d_status[g] (d_status or its
 * entries may be NULL).  The peer copies between DIFFERENT devices are unmeasured: no multi-GPU node was available to the
 * builder (the tests run several engines on one device). */
int isscabac_multi_decode_batch_dev(isscabac_multi *m, const uint8_t *d_payload_root, uint64_t payload_bytes,
                                    const uint64_t *d_offsets_root, const uint8_t *const *d_ctx,
                                    const uint32_t *const *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch,
                                    const uint8_t *ctx_init, int nctx, uint8_t *const *d_bins_out,
                                    uint32_t *const *d_status);

/* ---- on-disk container (host side, no GPU work) --------------------------------------------------------------------
 * The reference hands encoder output to the decoder through ONE FILE PER STREAM (ref: CABAC/SimpleCABACMex.cpp:195,288)
 * and sizes results by `dir(fn).bytes * 8` (ref: CABAC/cabacDemo.m:190-191, ISS/+coder/cabacEncode.m:100-101).  These two
 * calls convert between a batch (payload + offsets) and those files: stream s <-> paths[s], byte for byte what the
 * reference would have written / can read.  isscabac_read_stream_files: ISSCABAC_ERR_CAPACITY (offsets and
 * *total_bytes complete, nothing useful in payload) when cap is too small; payload may be NULL to ask for the sizes.
 * Errors of the calls in this section are reported through isscabac_last_error(NULL). */
int isscabac_write_stream_files(const uint8_t *payload, const uint64_t *offsets, uint64_t S, const char *const *paths);
int isscabac_read_stream_files(const char *const *paths, uint64_t S, uint8_t *payload, uint64_t cap, uint64_t *offsets,
                               uint64_t *total_bytes);
/* the name scheme the MEX gateway's 'writeStreamFiles' / 'readStreamFiles' use: "<prefix>%06llu.bin" */
int isscabac_stream_file_path(const char *prefix, uint64_t s, char *buf, uint64_t cap);

/* What ISS.m transmits next to its two CABAC streams (ref: ISS/ISS.m:197-201): the quantiser centroids cW and cH, the
 * matrix Q and -- unless equalProb -- the uint8 context initialisations xw, xh of the two streams (the p0_bytes_out of
 * isscabac_iss_init_context_model).  The reference only measures this through MATLAB's `save`; here it is a concrete
 * little-endian file holding the side information and the W and H bitstreams: everything a decoder needs. */
typedef struct isscabac_iss_side_info {
  const double *cW; /* ncW centroids */
  const double *cH; /* ncH centroids */
  const double *Q;  /* q_rows x q_cols, column-major */
  const uint8_t *xw, *xh; /* nxw / nxh context initialisation bytes (0 entries with equalProb) */
  uint32_t ncW, ncH, q_rows, q_cols, nxw, nxh;
} isscabac_iss_side_info;
typedef struct isscabac_iss_container_info {
  uint32_t ncW, ncH, q_rows, q_cols, nxw, nxh;
  uint64_t w_bytes, h_bytes;
} isscabac_iss_container_info;
int isscabac_iss_container_write(const char *path, const isscabac_iss_side_info *si, const uint8_t *w_stream,
                                 uint64_t w_bytes, const uint8_t *h_stream, uint64_t h_bytes);
int isscabac_iss_container_sizes(const char *path, isscabac_iss_container_info *out);
/* buffers sized from isscabac_iss_container_sizes */
int isscabac_iss_container_read(const char *path, double *cW, double *cH, double *Q, uint8_t *xw, uint8_t *xh,
                                uint8_t *w_stream, uint8_t *h_stream);

/* ---- measurement hooks --------------------------------------------------------------------------- */
/* when enabled, every batch call brackets each of its kernels with hipEvents on the engine's stream */
int isscabac_set_profiling(isscabac_engine *e, int enabled);
/* kernel times of the last batch call in ms: [0] encode kernel, [1] offsets scan, [2] payload gather,
 * [3] decode kernel; unused entries are 0 */
int isscabac_last_kernel_ms(isscabac_engine *e, float ms[4]);
/* mean kernel times (same four slots) over the batch calls since the previous call of this function, and how many
 * launches each mean covers; waits for the enqueued work first.  Up to 128 calls are kept between reads (a full ring is read out early, which waits for the stream). */
int isscabac_kernel_ms_mean(isscabac_engine *e, float ms[4], uint32_t launches[4]);
/* the individual launch times of one slot (0 encode kernel, 1 scan, 2 gather, 3 decode kernel) recorded since the
 * previous complete read, oldest first (at most 65536 are kept): *count = how many there are; the first min(cap, *count)
 * are copied to ms, and they are forgotten once cap covered them all.  Waits for the enqueued work first. */
int isscabac_kernel_ms_samples(isscabac_engine *e, int slot, float *ms, uint32_t cap, uint32_t *count);
/* launches of `slot` whose time was NOT kept because 65536 were already waiting to be read; reading resets the count.  A
 * non-zero value says the series of isscabac_kernel_ms_samples is truncated (the means of isscabac_kernel_ms_mean are not). */
int isscabac_kernel_ms_samples_dropped(isscabac_engine *e, int slot, uint64_t *dropped);

/* synthetic workload generator of SURVEY.md section 8(d) (counter based, identical to
 * oracle/cabac_oracle.c:orc_synth_fill), filling device buffers for global streams s0 .. s0+S-1 */
int isscabac_synth_fill_dev(isscabac_engine *e, uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins,
                            uint32_t pitch, uint32_t nctx, int ep_half, uint8_t *d_bins, uint8_t *d_ctx);

/* raw device memory helpers for callers without their own HIP allocator (MEX shim, tests) */
int isscabac_dev_alloc(isscabac_engine *e, uint64_t bytes, void **d_ptr);
int isscabac_dev_free(isscabac_engine *e, void *d_ptr);
int isscabac_copy_h2d(isscabac_engine *e, void *d_dst, const void *h_src, uint64_t bytes);
int isscabac_copy_d2h(isscabac_engine *e, void *h_dst, const void *d_src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
