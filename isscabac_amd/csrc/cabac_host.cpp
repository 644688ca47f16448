// cabac_host.cpp -- host twins of the SERIAL entry points of include/isscabac.h: isscabac_cpu_stream_* (the reference's
// per-bin commands, one handle = one `class CABAC`, ref: CABAC/SimpleCABACMex.cpp:69-97, 186-354) and
// isscabac_cpu_encode_symbols / isscabac_cpu_decode_symbols (a handful of streams of symbols, e.g. a whole matrix as ONE
// stream: ref: ISS/+coder/cabacEncode.m:45-70, cabacDecode.m:29-55).
//
// Why they exist: these cases have no parallelism to give a GPU -- one bin per call, or one stream of 16 000 bins on one
// lane -- and a lone GPU lane issues an instruction every ~2 ns behind a PCIe round trip: 2.3 us per decodeBin and 1.3 / 2.5 ms
// per 400 x 20 matrix on the device against ~15 ns per bin on one host core (profiles/r05/per_bin_native.txt,
// iss_matrix_rate.txt).  SURVEY 8(b) asks for `cabac_cpu_*` twins for exactly that.
//
// What they are: the product's own per-lane code -- cabac_lane.h / cabac_symbols.h, the functions the HIP kernels call --
// compiled for the host.  They are separate, explicitly named entry points, not a fallback: no isscabac_* batch call ever
// routes here, the batch calls still fail with ISSCABAC_ERR_HIP without a GPU, and nothing here touches oracle/.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/isscabac.h"
#include "cabac_lane.h"
#include "cabac_symbols.h"

using namespace isscabac;

namespace {

const uint32_t kLpsRow[64] = {ISSCABAC_LPS_ROWS};
const uint8_t kTransLps[64] = {ISSCABAC_TRANS_LPS};
const uint32_t kProbRow[49] = {ISSCABAC_PROB_ROWS};
struct HostNextTab {
  uint32_t e[64];
  HostNextTab() {
    for (uint32_t s = 0; s < 64; s++) e[s] = next_state_entry(s, kTransLps[s]);
  }
};
const HostNextTab kNext;

constexpr int kStreamCtx = 1000;  // RWTH_MAX_NUM_CONTEXTS, ref: CommonDef.h:56

// States policy of cabac_lane.h over a plain array of doubled states
struct HostStates {
  uint8_t *q;
  uint32_t load(uint32_t c) const { return q[c]; }
  void store(uint32_t c, uint32_t v) const { q[c] = (uint8_t)v; }
  uint32_t addr(uint32_t c) const { return c; }
  uint32_t load_at(uint32_t a) const { return q[a]; }
  void store_at(uint32_t a, uint32_t v) const { q[a] = (uint8_t)v; }
  uint32_t lps_row(uint32_t v) const { return kLpsRow[v >> 2]; }
  uint32_t next_entry(uint32_t v) const { return kNext.e[v >> 2]; }
};
struct VecSink {
  std::vector<uint8_t> *v;
  void put(uint32_t b) { v->push_back((uint8_t)b); }
};
struct CountSink {  // a stream's bytes into a caller's buffer, counting on past its end
  uint8_t *buf;
  uint64_t len, cap;
  void put(uint32_t b) {
    if (len < cap) buf[len] = (uint8_t)b;
    len++;
  }
};

// trace record of one op, as the device handle writes it (cabac_stream.hip: trace_record)
inline uint32_t trace_record(uint32_t ctx, uint32_t bin, uint32_t q_before, uint32_t q_after) {
  return ctx | (bin << 10) | (1u << 11) | (packed_from_q(q_before) << 16) | (packed_from_q(q_after) << 24);
}

inline void host_enc_op(EncCore &e, HostStates &cs, uint32_t op, uint32_t bin, uint32_t *rec) {
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    enc_bin_ep(e, bin);
    *rec = 0;
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    enc_bin_prob(e, bin, p, kProbRow[prob_row_index(p)]);
    *rec = 0;
  } else {
    const uint32_t q0 = cs.load(op);
    enc_bin(e, bin << 1, op, cs);
    *rec = trace_record(op, bin, q0, cs.load(op));
  }
}
inline uint32_t host_dec_op(RefDec &d, HostStates &cs, RefReader &rd, uint32_t op, uint32_t *rec) {
  uint32_t bin;
  *rec = 0;
  if (op == ISSCABAC_OP_BYPASS || op == ISSCABAC_OP_PROB(50)) {
    bin = ref_dec_bin_ep(d, rd);
  } else if (op & 0x8000u) {
    const uint32_t p = op & 0x7Fu;
    bin = ref_dec_bin_prob(d, p, kProbRow[prob_row_index(p)], rd);
  } else {
    const uint32_t q0 = cs.load(op);
    bin = ref_dec_bin(d, op, cs, rd);
    *rec = trace_record(op, bin, q0, cs.load(op));
  }
  return bin;
}

}  // namespace

struct isscabac_cpu_stream {
  int nctx = 0;
  uint8_t q_init[kStreamCtx], q_enc[kStreamCtx], q_dec[kStreamCtx];  // doubled states; one set per side, ref: SimpleCABACMex.cpp:175-176
  std::string err;
  // encoder side
  bool encoding = false, enc_done = false;
  EncCore enc;
  std::vector<uint8_t> out;
  // decoder side
  bool decoding = false;
  RefDec dec;  // the reference's own formulation, like the device handle: exact on any input
  RefReader rd;
  std::vector<uint8_t> payload;
  // state traces (off by default): one record per op and side
  bool tracing = false;
  std::vector<uint32_t> trace[2];
};

namespace {
int sfail(isscabac_cpu_stream *s, int code, const char *msg) {
  s->err = msg;
  return code;
}
// the u8 forms' context byte -> 16-bit op, as the device handle maps it (cabac_stream.hip: op_from_u8)
inline uint16_t op_from_u8(const isscabac_cpu_stream *s, uint8_t c) {
  if (c == ISSCABAC_BYPASS) return ISSCABAC_OP_BYPASS;
  if (s->nctx <= ISSCABAC_PROB_MAX_CTX && c > ISSCABAC_PROB_BASE) return (uint16_t)ISSCABAC_OP_PROB(c - ISSCABAC_PROB_BASE);
  return c;
}
const char *op_problem(const isscabac_cpu_stream *s, uint16_t op) {
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
}  // namespace

extern "C" {

int isscabac_cpu_stream_open(const uint8_t *ctx_init, int nctx, isscabac_cpu_stream **out) {
  if (!out) return ISSCABAC_ERR_ARG;
  *out = nullptr;
  if (!ctx_init || nctx < 1 || nctx > ISSCABAC_STREAM_MAX_CTX) return ISSCABAC_ERR_ARG;
  for (int c = 0; c < nctx; c++)
    if ((ctx_init[c] >> 1) > 62) return ISSCABAC_ERR_ARG;  // state 63: refused like everywhere (isscabac_init_by_state)
  isscabac_cpu_stream *s = new isscabac_cpu_stream;
  s->nctx = nctx;
  for (int c = 0; c < kStreamCtx; c++) s->q_init[c] = (uint8_t)q_from_packed(1);  // ContextModel() default (mps 1, state 0), ref: ContextModel.cpp:54
  for (int c = 0; c < nctx; c++) s->q_init[c] = (uint8_t)q_from_packed(ctx_init[c]);
  memcpy(s->q_enc, s->q_init, sizeof s->q_init);
  memcpy(s->q_dec, s->q_init, sizeof s->q_init);
  *out = s;
  return ISSCABAC_OK;
}

void isscabac_cpu_stream_close(isscabac_cpu_stream *s) { delete s; }

const char *isscabac_cpu_stream_last_error(const isscabac_cpu_stream *s) { return s ? s->err.c_str() : ""; }

int isscabac_cpu_stream_set_trace(isscabac_cpu_stream *s, int enabled) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (s->encoding || s->decoding) return sfail(s, ISSCABAC_ERR_STATE, "isscabac_cpu_stream_set_trace: switch tracing before encodeStart / decodeStart");
  s->tracing = enabled != 0;
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_encode_start(isscabac_cpu_stream *s) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (s->encoding) return sfail(s, ISSCABAC_ERR_STATE, "encodeStart: the bitstream is already open for writing");  // ref asserts, CABAC_BitstreamFile.cpp:52
  enc_start(s->enc);
  s->out.clear();
  s->trace[0].clear();
  s->encoding = true;
  s->enc_done = false;
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_encode_ops(isscabac_cpu_stream *s, const uint8_t *bins, const uint16_t *ops, uint32_t n) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->encoding) return sfail(s, ISSCABAC_ERR_STATE, "encodeBin before encodeStart");
  if (n && (!bins || !ops)) return sfail(s, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++) {
    if (bins[i] > 1) return sfail(s, ISSCABAC_ERR_ARG, "Error: invalid input 3, bin to be encoded should either be 1 or 0");  // ref: :227
    if (const char *why = op_problem(s, ops[i])) return sfail(s, ISSCABAC_ERR_ARG, why);
  }
  HostStates cs{s->q_enc};
  VecSink sink{&s->out};
  for (uint32_t i = 0; i < n; i++) {
    uint32_t rec;
    host_enc_op(s->enc, cs, ops[i], bins[i], &rec);
    enc_flush(s->enc, sink);  // after EVERY bin, like testAndWriteOut (ref: CABAC_ArithmeticEncoder.cpp:369-375): getNumBits is the reference's
    if (s->tracing) s->trace[0].push_back(rec);
  }
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_encode_bins(isscabac_cpu_stream *s, const uint8_t *bins, const uint8_t *ctx, uint32_t n) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (n && !ctx) return sfail(s, ISSCABAC_ERR_ARG, "encodeBin: null buffer");
  uint16_t small[64];
  for (uint32_t i0 = 0; i0 < n || i0 == 0; i0 += 64) {
    const uint32_t m = std::min<uint32_t>(64, n - i0);
    for (uint32_t i = 0; i < m; i++) small[i] = op_from_u8(s, ctx[i0 + i]);
    const int rc = isscabac_cpu_stream_encode_ops(s, bins ? bins + i0 : nullptr, small, m);
    if (rc || n == 0) return rc;
  }
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_num_bits(isscabac_cpu_stream *s, uint64_t *bits) {
  if (!s || !bits) return ISSCABAC_ERR_ARG;
  *bits = 8ull * s->out.size();  // CABAC_BitstreamFile::getNumberOfWrittenBits: no held bits before finish()
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_encode_finish(isscabac_cpu_stream *s, uint8_t *out, uint64_t cap, uint64_t *nbytes) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->encoding && !s->enc_done) return sfail(s, ISSCABAC_ERR_STATE, "encodeFinish before encodeStart");
  if (!s->enc_done) {
    VecSink sink{&s->out};
    enc_finish(s->enc, sink);
    s->enc_done = true;
    s->encoding = false;
  }
  if (nbytes) *nbytes = s->out.size();
  if (s->out.size() > cap || !out) return sfail(s, ISSCABAC_ERR_CAPACITY, "encodeFinish: the caller's buffer is too small");
  memcpy(out, s->out.data(), s->out.size());
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_decode_start(isscabac_cpu_stream *s, const uint8_t *payload, uint64_t nbytes) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!payload && nbytes) return sfail(s, ISSCABAC_ERR_ARG, "decodeStart: null payload");
  if (nbytes > 0xFFFFFFF0ull) return sfail(s, ISSCABAC_ERR_ARG, "decodeStart: payload too large");
  s->payload.assign(payload, payload + nbytes);
  s->rd = RefReader{s->payload.data(), (uint32_t)nbytes, 0, 0};
  ref_dec_start(s->dec, s->rd);
  s->trace[1].clear();
  s->decoding = true;
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_decode_ops(isscabac_cpu_stream *s, const uint16_t *ops, uint32_t n, uint8_t *bins_out) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->decoding) return sfail(s, ISSCABAC_ERR_STATE, "decodeBin before decodeStart");
  if (n == 0) return ISSCABAC_OK;
  if (!ops || !bins_out) return sfail(s, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  for (uint32_t i = 0; i < n; i++)
    if (const char *why = op_problem(s, ops[i])) return sfail(s, ISSCABAC_ERR_ARG, why);
  HostStates cs{s->q_dec};
  for (uint32_t i = 0; i < n; i++) {
    uint32_t rec;
    bins_out[i] = (uint8_t)host_dec_op(s->dec, cs, s->rd, ops[i], &rec);
    if (s->tracing) s->trace[1].push_back(rec);
  }
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_decode_bins(isscabac_cpu_stream *s, const uint8_t *ctx, uint32_t n, uint8_t *bins_out) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (n && !ctx) return sfail(s, ISSCABAC_ERR_ARG, "decodeBin: null buffer");
  uint16_t small[64];
  for (uint32_t i0 = 0; i0 < n; i0 += 64) {
    const uint32_t m = std::min<uint32_t>(64, n - i0);
    for (uint32_t i = 0; i < m; i++) small[i] = op_from_u8(s, ctx[i0 + i]);
    const int rc = isscabac_cpu_stream_decode_ops(s, small, m, bins_out ? bins_out + i0 : nullptr);
    if (rc) return rc;
  }
  return s->decoding ? ISSCABAC_OK : sfail(s, ISSCABAC_ERR_STATE, "decodeBin before decodeStart");
}

int isscabac_cpu_stream_decode_finish(isscabac_cpu_stream *s, uint32_t *status) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->decoding) return sfail(s, ISSCABAC_ERR_STATE, "decodeFinish before decodeStart");
  const uint32_t st = ref_dec_finish(s->dec, s->rd, (uint32_t)s->payload.size());
  if (status) *status = st;
  s->decoding = false;
  return ISSCABAC_OK;
}

int isscabac_cpu_stream_get_stats(isscabac_cpu_stream *s, int decoder, int ctx, uint8_t *trace5, uint64_t cap_steps, uint64_t *nsteps,
                                  uint32_t *transitions) {
  if (!s) return ISSCABAC_ERR_ARG;
  if (!s->tracing) return sfail(s, ISSCABAC_ERR_STATE, "getEncoderStats: tracing is off (isscabac_cpu_stream_set_trace)");
  if (ctx < 0 || ctx >= s->nctx) return sfail(s, ISSCABAC_ERR_ARG, "Error: invalid context index");
  const std::vector<uint32_t> &t = s->trace[decoder ? 1 : 0];
  return isscabac_trace_stats(t.data(), t.size(), ctx, decoder, trace5, cap_steps, nsteps, transitions);
}

// ---- a few streams of symbols on one host core ---------------------------------------------------------------------------
// Same arguments and results as isscabac_encode_symbols_stats / isscabac_decode_symbols (host-buffer forms) without the engine:
// binarisation (cabac_symbols.h: binarize), context selection (SymWalk::ctx) and the bin coder of cabac_lane.h, stream after
// stream.  Bytes leave `low` after every bin, so the statistics are getNumBits differences as +coder/cabacEncode.m forms them
// (ref: ISS/+coder/cabacEncode.m:39-42, 60-67).
static bool sym_params_ok(const isscabac_symbol_params *p, SymParams *P, bool decode) {
  if (!p) return false;
  P->method = p->binarization;
  P->k = p->k;
  P->nq = p->nq;
  P->selector = p->selector;
  P->types = p->iss_types;
  P->nlbp = p->nlbp;
  P->col_rows = p->col_rows;
  if (isscabac_symbol_max_bins(p) == 0) return false;
  if (decode && (P->method == ISSCABAC_BIN_TRK || P->method == ISSCABAC_BIN_FL32)) return false;  // no end-of-symbol rule in the reference
  return true;
}

int isscabac_cpu_encode_symbols(const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym, uint32_t pitch,
                                const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *payload,
                                uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out,
                                const isscabac_symbol_stats *stats) {
  SymParams P;
  if (!sym || !payload || !offsets || !ctx_init || S == 0 || nsym > pitch || nctx < 1 || nctx > ISSCABAC_MAX_CTX || !sym_params_ok(params, &P, false))
    return ISSCABAC_ERR_ARG;
  for (int c = 0; c < nctx; c++)
    if ((ctx_init[c] >> 1) > 62) return ISSCABAC_ERR_ARG;
  uint64_t pos = 0;
  for (uint64_t s = 0; s < S; s++) {
    uint8_t q[256];
    for (int c = 0; c < 256; c++) q[c] = (uint8_t)q_from_packed(c < nctx ? ctx_init[c] : 1u);
    HostStates cs{q};
    EncCore e;
    enc_start(e);
    CountSink sink{payload + std::min(pos, payload_cap), 0, payload_cap - std::min(pos, payload_cap)};
    const uint32_t n = std::min(nsym_per_stream ? nsym_per_stream[s] : nsym, nsym);
    const uint8_t *row = sym + s * pitch;
    SymWalk w;
    w.start();
    uint32_t nbins = 0;
    uint32_t *hist = stats && stats->ctx_hist ? stats->ctx_hist + s * ISSCABAC_STAT_CTX : nullptr;
    uint32_t *cost = stats && stats->ctx_cost ? stats->ctx_cost + s * ISSCABAC_STAT_CTX : nullptr;
    if (hist) memset(hist, 0, ISSCABAC_STAT_CTX * sizeof(uint32_t));
    if (cost) memset(cost, 0, ISSCABAC_STAT_CTX * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t v = std::min<uint32_t>(row[i], P.nq - 1);  // (symbols beyond the alphabet are clamped, as the kernels clamp them)
      const BinString g = binarize(v, P);
      const uint64_t sym_before = sink.len;
      w.cur = g;  // the selector may look at the symbol's own earlier bins; bins from n on are never read (iss_ctx)
      for (w.n = 1; w.n <= g.len; w.n++) {
        const uint32_t c = w.ctx(P);
        const uint32_t bin = (g.bits >> (w.n - 1)) & 1u;
        const uint64_t before = sink.len;
        const uint32_t q0 = cs.load(c);
        enc_bin(e, bin << 1, c, cs);
        enc_flush(e, sink);
        if (c < ISSCABAC_STAT_CTX) {
          if (hist) hist[c]++;
          if (cost) cost[c] += (uint32_t)(8u * (sink.len - before));
        }
        if (stats && stats->bin_trace && nbins < stats->bin_trace_pitch)
          stats->bin_trace[s * stats->bin_trace_pitch + nbins] = trace_record(c, bin, q0, cs.load(c));
        nbins++;
      }
      if (stats && stats->sym_bits) stats->sym_bits[s * pitch + i] = (uint16_t)(8u * (sink.len - sym_before));
      w.cur = g;
      w.next_symbol(P);
    }
    enc_finish(e, sink);
    offsets[s] = pos;
    pos += sink.len;
    if (nbins_out) nbins_out[s] = nbins;
  }
  offsets[S] = pos;
  if (total_bytes) *total_bytes = pos;
  return pos > payload_cap ? ISSCABAC_ERR_CAPACITY : ISSCABAC_OK;
}

int isscabac_cpu_decode_symbols(const uint8_t *payload, const uint64_t *offsets, const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym,
                                uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *sym_out,
                                uint32_t *status) {
  SymParams P;
  if (!payload || !offsets || !sym_out || !ctx_init || S == 0 || nsym > pitch || nctx < 1 || nctx > ISSCABAC_MAX_CTX || !sym_params_ok(params, &P, true))
    return ISSCABAC_ERR_ARG;
  for (uint64_t s = 0; s < S; s++)
    if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] > 0xFFFFFFF0ull) return ISSCABAC_ERR_ARG;
  const uint32_t maxbins = isscabac_symbol_max_bins(params);
  for (uint64_t s = 0; s < S; s++) {
    uint8_t q[256];
    for (int c = 0; c < 256; c++) q[c] = (uint8_t)q_from_packed(c < nctx ? ctx_init[c] : 1u);
    HostStates cs{q};
    RefReader rd{payload + offsets[s], (uint32_t)(offsets[s + 1] - offsets[s]), 0, 0};
    RefDec d;
    ref_dec_start(d, rd);
    const uint32_t n = std::min(nsym_per_stream ? nsym_per_stream[s] : nsym, nsym);
    uint8_t *row = sym_out + s * pitch;
    SymWalk w;
    w.start();
    for (uint32_t i = 0; i < n; i++) {
      // bins until the end-of-symbol rule fires (ref: cabacDecodeSymbolFinished.m:12-30); a corrupt stream cannot run on: a symbol
      // has at most maxbins bins, and the string is cut there like the kernels cut it
      for (;;) {
        const uint32_t c = w.ctx(P);
        const uint32_t bin = ref_dec_bin(d, c, cs, rd);
        w.cur = append_bin(w.cur, w.n, bin);
        if (symbol_finished(w.cur, P) || w.n >= maxbins) break;
        w.n++;
      }
      row[i] = (uint8_t)std::min<uint32_t>(debinarize(w.cur, P), 255u);
      w.next_symbol(P);
    }
    const uint32_t st = ref_dec_finish(d, rd, rd.len);
    if (status) status[s] = st;
  }
  return ISSCABAC_OK;
}

}  // extern "C"
