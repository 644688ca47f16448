/*
 * cabac_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's HM-derived CABAC hot path
 * (christianrohlfing/ISScabac, CABAC/ directory).  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check
 * the HIP engine bit-for-bit.  Nothing under isscabac_amd/ may include, link
 * or call it.
 *
 * Parity pinning: every function here is checked (tests/test_oracle_vs_ref.py,
 * run in the build container) against the REAL reference classes compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/libcabac_ref.so, and
 * (tests/test_oracle_golden.py, runs anywhere) against the byte-exact golden
 * vectors in tests/golden/ that were produced by that library.
 *
 * Citations are relative to /root/reference/.
 */
#ifndef CABAC_ORACLE_H
#define CABAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BYPASS_CTX 0xFFu /* ctx byte that marks an equiprobable (EP) bin */
/* ctx bytes 156..254 mark a fixed-probability bin with P(1) = byte - 155 percent (the engine's additive encoding of
 * encodeBinProb's second argument) in batches of at most 128 contexts */
#define ORC_PROB_BASE 155u
#define ORC_PROB_MAX_NCTX 128

/* ---- bit/byte sink: CABAC/CABAC_BitstreamFile.cpp:95-151 (write side), in memory ---- */
typedef struct {
  uint8_t *buf;
  size_t len, cap;
  uint32_t held_bits;     /* msb-aligned in the low 8 bits, like m_held_bits */
  uint32_t num_held;      /* m_num_held_bits */
  uint64_t bits_written;  /* m_num_bits_written */
  int overflow;           /* set when cap was too small (oracle-only condition) */
} orc_sink;

void orc_sink_init(orc_sink *s, uint8_t *buf, size_t cap);
void orc_sink_write(orc_sink *s, uint32_t bits, uint32_t nbits);   /* :95-138 */
void orc_sink_align_zero(orc_sink *s);                             /* :140-151 */
uint64_t orc_sink_bits(const orc_sink *s);                         /* CABAC_BitstreamFile.h:70 */

/* ---- context model: CABAC/ContextModel.h:72-121, ContextModel.cpp:70-123,136-158 ---- */
/* packed byte p = (state << 1) | mps, exactly m_ucState */
uint8_t orc_ctx_pack(uint32_t mps, uint32_t state);  /* ContextModel.cpp:70-72 */
uint8_t orc_ctx_next_mps(uint8_t p);                 /* m_aucNextStateMPS */
uint8_t orc_ctx_next_lps(uint8_t p);                 /* m_aucNextStateLPS */
uint8_t orc_lps_table(uint32_t state, uint32_t q);   /* sm_aucLPSTable[state][q] */
uint8_t orc_renorm_table(uint32_t lps_shr3);         /* sm_aucRenormTable[i] */

/* p(0) -> (mps,state): CABAC/CABAC_ContextModelsInit.cpp:124-148 */
uint8_t orc_p0_to_packed(double p0);
/* triples [ctxIdx, mps, state] (ctxIdx ignored): CABAC_ContextModelsInit.cpp:51-80 */
void orc_init_by_state(const double *triples, int nctx, uint8_t *packed_out);
void orc_init_by_prob(const double *p0, int nctx, uint8_t *packed_out);

/* ---- encoder: CABAC/CABAC_ArithmeticEncoder.cpp ---- */
typedef struct {
  uint32_t low, range, buffered_byte;
  int32_t num_buffered, bits_left;
  uint32_t bins_coded;
  orc_sink sink;
} orc_encoder;

void orc_enc_start(orc_encoder *e, uint8_t *buf, size_t cap);          /* :54-68 */
void orc_enc_bin(orc_encoder *e, uint32_t bin, uint8_t *ctx_packed);   /* :113-178 */
void orc_enc_bin_ep(orc_encoder *e, uint32_t bin);                     /* :250-270 */
void orc_enc_bins_ep(orc_encoder *e, uint32_t bins, int nbins);        /* :278-319 */
void orc_enc_bin_prob(orc_encoder *e, uint32_t bin, uint32_t prob);    /* :180-243 (RWTH_CABAC_FIXED_PROBABILITY) */
uint8_t orc_prob_table(uint32_t row, uint32_t q);                      /* sm_aucLPSTProbTable[row][q], :494-547 */
void orc_enc_bin_trm(orc_encoder *e, uint32_t bin);                    /* :326-367 */
void orc_enc_finish(orc_encoder *e);                                   /* :70-105 */
size_t orc_enc_bytes(const orc_encoder *e); /* bytes in the sink so far */
uint64_t orc_enc_bits(const orc_encoder *e); /* getNumBits, SimpleCABACMex.cpp:248-264 */

/* event counters for test-case selection (not thread safe; oracle-only instrumentation) */
typedef struct {
  uint64_t writeouts, ff_leads, carry_resolves, finish_carries, max_num_buffered;
} orc_counters;
void orc_counters_reset(void);
const orc_counters *orc_counters_get(void);

/* ---- decoder: CABAC/CABAC_ArithmeticDecoder.cpp ---- */
typedef struct {
  const uint8_t *buf;
  size_t len, pos;
  uint32_t last_byte;   /* m_cLastCharRead */
  uint32_t overreads;   /* reads past len (the fstream would hand back 0xFF) */
  uint32_t range, value;
  int32_t bits_needed;
} orc_decoder;

void orc_dec_start(orc_decoder *d, const uint8_t *buf, size_t len);    /* :54-71 */
uint32_t orc_dec_bin(orc_decoder *d, uint8_t *ctx_packed);             /* :87-190 */
uint32_t orc_dec_bin_ep(orc_decoder *d);                               /* :288-331 */
uint32_t orc_dec_bins_ep(orc_decoder *d, int nbins);                   /* :333-421 */
uint32_t orc_dec_bin_prob(orc_decoder *d, uint32_t prob);              /* :192-286 (RWTH_CABAC_FIXED_PROBABILITY) */
uint32_t orc_dec_bin_trm(orc_decoder *d);                              /* :423-472 */
/* finish (:73-85) with the two asserts turned into a status word:
 * bit0 = terminate bin was not 1, bit1 = stop bit not where expected,
 * bit2 = decoder read past the end of the payload. 0 = stream intact. */
uint32_t orc_dec_finish(orc_decoder *d);

/* ---- whole-stream helpers (ctx byte 0xFF = EP bin) ---- */
/* returns payload length in bytes (0 on sink overflow); *bits_before_finish =
 * getNumBits right before finish() as SimpleCABACMex.cpp:248-264 would report */
size_t orc_encode_stream(const uint8_t *bins, const uint8_t *ctx, uint32_t nbins,
                         const uint8_t *ctx_init, int nctx,
                         uint8_t *out, size_t cap, uint64_t *bits_before_finish,
                         uint8_t *ctx_final /* nullable, nctx bytes */);
uint32_t orc_decode_stream(const uint8_t *payload, size_t len, const uint8_t *ctx,
                           uint32_t nbins, const uint8_t *ctx_init, int nctx,
                           uint8_t *bins_out, uint8_t *ctx_final /* nullable */);

/* stream-major batch: bins/ctx are [S][B]; nbins[s] (nullable => B) bins used.
 * payload is the tight concatenation, offsets has S+1 entries. returns total bytes,
 * or (uint64_t)-1 if cap is too small. */
uint64_t orc_encode_batch(const uint8_t *bins, const uint8_t *ctx, const uint32_t *nbins,
                          uint64_t S, uint32_t B, const uint8_t *ctx_init, int nctx,
                          uint8_t *payload, uint64_t cap, uint64_t *offsets);
void orc_decode_batch(const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                      const uint32_t *nbins, uint64_t S, uint32_t B,
                      const uint8_t *ctx_init, int nctx, uint8_t *bins_out,
                      uint32_t *status /* nullable, S entries */);

/* ---- binarisation: CABAC/cabacBinarizer.m:30-37,56-69; cabacDebinarizer.m:28-31,46-53;
 *      cabacDecodeSymbolFinished.m:12-30 ---- */
/* write the bin string of v, return its length */
int orc_binarize_tu(uint32_t v, uint32_t max_val, uint8_t *bins);
int orc_binarize_egk(uint32_t v, uint32_t k, uint8_t *bins);
int orc_binarize_trk(uint32_t v, uint32_t k, uint32_t max_val, uint8_t *bins); /* TRCode, cabacBinarizer.m:39-54 */
int orc_binarize_fl(uint32_t v, int nbits, uint8_t *bins);                     /* FLCode, :71-75 */
uint32_t orc_debinarize_tu(const uint8_t *bins, int n, uint32_t max_val);
uint32_t orc_debinarize_egk(const uint8_t *bins, int n, uint32_t k);

/* ---- context selectors ---- */
/* ISS selector: ISS/+coder/cabacContextSelection.m:24-67.  n is the 1-based bin
 * position; g_prev = already coded bins of this symbol (n-1 of them); g_up1 = bin
 * string of the symbol one row up (len_up1 = 0 when there is none).  types is a bit
 * mask: 1 cond0, 2 cond1, 4 conds0, 8 conds1, 16 condbinlft.  Returns the 1-based id. */
int orc_iss_ctx(int n, const uint8_t *g_prev, const uint8_t *g_up1, int len_up1,
                unsigned types, int nlbp);

/* encode a column of symbols the way ISS/+coder/cabacEncode.m:45-70 walks ONE column
 * (d = 1..rows), EGk or TU binarisation, ISS context selection; and decode it back the way
 * cabacDecode.m:29-55 does.  method: 0 = TU (max_val = nq-1), 1 = EGk; encode side only: 2 = TRk, 3 = FL32 (the
 * reference cannot decode those: cabacDecodeSymbolFinished.m:12-30 knows TU and EG0-2). */
size_t orc_iss_encode_column(const uint32_t *sym, int rows, uint32_t nq, int method, uint32_t k,
                             unsigned types, int nlbp, const uint8_t *ctx_init, int nctx,
                             uint8_t *out, size_t cap, uint32_t *nbins_out);
uint32_t orc_iss_decode_column(const uint8_t *payload, size_t len, int rows, uint32_t nq,
                               int method, uint32_t k, unsigned types, int nlbp,
                               const uint8_t *ctx_init, int nctx, uint32_t *sym_out);

/* the same loops over a whole column-major matrix coded as ONE stream (the reference's actual ISS mode) */
size_t orc_iss_encode_matrix(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                             unsigned types, int nlbp, const uint8_t *ctx_init, int nctx, uint8_t *out, size_t cap,
                             uint32_t *nbins_out);
size_t orc_iss_encode_matrix_stats(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                                   unsigned types, int nlbp, const uint8_t *ctx_init, int nctx, uint8_t *out, size_t cap,
                                   uint32_t *ctx_hist, uint32_t *ctx_cost, uint32_t *sym_bits);
uint32_t orc_iss_decode_matrix(const uint8_t *payload, size_t len, int rows, int cols, uint32_t nq, int method,
                               uint32_t k, unsigned types, int nlbp, const uint8_t *ctx_init, int nctx,
                               uint32_t *sym_out);

/* empirical p(0) per ISS context from the binarised matrix: ISS/+coder/cabacInitContextModel.m:1-130 (restated from
 * the MATLAB text; unpinned: no MATLAB available).  out: 7*nlbp+2 doubles. */
void orc_iss_init_context_model(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                                unsigned types, int nlbp, double *out);
uint8_t orc_matlab_uint8_of_p(double p); /* uint8(p*255), cabacEncode.m:30 */

/* demo selector + TU: CABAC/cabacDemo.m:92-131 (encode), :148-176 (decode) */
size_t orc_demo_encode(const uint32_t *sym, int nsym, uint32_t nq, const uint8_t *ctx_init,
                       uint8_t *out, size_t cap, uint32_t *nbins_out);
uint32_t orc_demo_decode(const uint8_t *payload, size_t len, int nsym, uint32_t nq,
                         const uint8_t *ctx_init, uint32_t *sym_out);

/* ---- state traces (getEncoderStats / getDecoderStats of the reference's Windows build) ---- */
uint8_t orc_trace_state(uint8_t packed); /* ContextModel.cpp:128-129 */
uint64_t orc_trace_stats(const uint8_t *bins, const uint8_t *ctx, uint64_t nbins, const uint8_t *ctx_init, int nctx,
                         int which_ctx, int decoder_side, uint8_t *trace5, uint64_t cap_steps, uint32_t *trans);

/* ---- synthetic workload generator of SURVEY.md section 8(d) ---- */
/* fills bins/ctx ([nS][B], stream-major) for global stream ids s0 .. s0+nS-1.
 * ep_half != 0 => config-5 style: bypass iff (h >> 31). */
void orc_synth_fill(uint32_t seed, uint64_t s0, uint64_t nS, uint32_t B, uint32_t C,
                    int ep_half, uint8_t *bins, uint8_t *ctx);

/* all-core CPU ceiling: the restatement, in-memory sink, native threads (bench.py cpu_baseline) */
int orc_roundtrip_threads(uint32_t seed, uint64_t s0, uint32_t B, uint32_t C, int ep_half, int nthreads, double seconds,
                          uint64_t *streams_done, double *wall_s);

#ifdef __cplusplus
}
#endif
#endif
