// ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" driver around the REAL reference classes.  oracle/Makefile compiles it together
// with the reference's own sources, taken where they lie under /root/reference/CABAC (nothing is
// copied into this repo), into oracle/_ref/libcabac_ref.so.  It is used to
//   * validate oracle/cabac_oracle.c bin by bin (tests/test_oracle_vs_ref.py),
//   * generate the golden vectors in tests/golden/ (tests/golden/make_golden.py),
//   * time the reference itself as bench.py's cpu_baseline (kind "reference").
// oracle/Makefile builds it twice: _ref/libcabac_ref.so as the reference ships on Linux, and _ref/libcabac_ref_full.so
// with the reference's two compile-time switches on -- RWTH_TRACE_CABAC_STATES (CommonDef.h:39-43: on under _WIN32, so
// the build passes -D_WIN32) and RWTH_CABAC_FIXED_PROBABILITY (CommonDef.h:50: hard-defined 0, so the Makefile derives
// a flipped copy of that one header with sed at build time and force-includes it; the include guard then skips the
// original).  The second build pins the fixed-probability bins and the state traces.
// The reference bit sink is an std::fstream (CABAC_BitstreamFile.cpp:50-86), so every stream goes
// through a scratch file whose path the caller supplies (put it on /dev/shm).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "CABAC_ArithmeticDecoder.h"
#include "CABAC_ArithmeticEncoder.h"
#include "CABAC_BitstreamFile.h"
#include "ContextModel.h"

namespace {

// expose the protected coder registers for step-by-step comparison with the restatement
struct EncPeek : CABAC_ArithmeticEncoder {
  using CABAC_ArithmeticEncoder::CABAC_ArithmeticEncoder;
  void snapshot(uint32_t *o) const {
    o[0] = m_uiLow; o[1] = m_uiRange; o[2] = m_bufferedByte;
    o[3] = (uint32_t)m_numBufferedBytes; o[4] = (uint32_t)m_bitsLeft;
  }
};
struct DecPeek : CABAC_ArithmeticDecoder {
  using CABAC_ArithmeticDecoder::CABAC_ArithmeticDecoder;
  void snapshot(uint32_t *o) const { o[0] = m_uiRange; o[1] = m_uiValue; o[2] = (uint32_t)m_bitsNeeded; }
  int bitsNeeded() const { return m_bitsNeeded; }
};

void init_models(std::vector<ContextModel> &cm, const uint8_t *packed, int nctx) {
  cm.resize(nctx > 0 ? nctx : 1);
  for (int c = 0; c < nctx; c++) cm[c].init(packed[c] & 1u, packed[c] >> 1);
}

#if RWTH_TRACE_CABAC_STATES
// the context models of the most recent ref_encode_ops / ref_decode_ops run, for ref_get_trace
std::vector<ContextModel> g_last_cm[2];
#endif

long slurp(const char *path, uint8_t *out, size_t cap) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return -1;
  long n = (long)std::fread(out, 1, cap, f);
  int more = std::fgetc(f);
  std::fclose(f);
  return more == EOF ? n : -2;  // -2: cap too small
}

}  // namespace

extern "C" {

// ops: kind 0 = context bin (val = bin, arg = ctx), 1 = encodeBinEP(val), 2 = encodeBinsEP(val, arg)
// trace (nullable): 5 u32 per op = encoder registers after the op; bits (nullable): getNumberOfWrittenBits
// after each op.  Returns the file length, <0 on error.
long ref_encode_ops(const uint8_t *kind, const uint32_t *val, const uint32_t *arg, uint32_t nops,
                    const uint8_t *ctx_init, int nctx, const char *scratch, uint8_t *out, size_t cap,
                    uint32_t *trace, uint32_t *bits, uint8_t *ctx_final) {
  std::vector<ContextModel> cm;
  init_models(cm, ctx_init, nctx);
  CABAC_BitstreamFile bs;
  if (!bs.openOutputFile(scratch)) return -1;
  EncPeek enc(&bs);
  enc.start();
  for (uint32_t i = 0; i < nops; i++) {
    switch (kind[i]) {
      case 0: {
#if RWTH_TRACE_CABAC_STATES
        // exactly what the MEX gateway does around encodeBin, SimpleCABACMex.cpp:231-242
        ContextModel *m = &cm[arg[i]];
        uint8_t bin = (uint8_t)val[i];
        uint8_t mps_p = m->getMps();
        uint8_t state_p = m->getState();
#endif
        enc.encodeBin(val[i], &cm[arg[i]]);
#if RWTH_TRACE_CABAC_STATES
        uint8_t mps_a = m->getMps();
        uint8_t state_a = m->getState();
        m->addCabacStep(bin, state_p, mps_p, state_a, mps_a);
#endif
        break;
      }
      case 1: enc.encodeBinEP(val[i]); break;
      case 2: enc.encodeBinsEP(val[i], (int)arg[i]); break;
#if RWTH_CABAC_FIXED_PROBABILITY
      case 3: enc.encodeBinProb(val[i], arg[i]); break;  // arg = probability of a 1 in percent
#endif
      default: return -3;
    }
    if (trace) enc.snapshot(trace + 5 * (size_t)i);
    if (bits) bits[i] = bs.getNumberOfWrittenBits();
  }
  enc.finish();
  bs.closeFile();
  if (ctx_final)
    for (int c = 0; c < nctx; c++) ctx_final[c] = (uint8_t)((cm[c].getState() << 1) | cm[c].getMps());
#if RWTH_TRACE_CABAC_STATES
  g_last_cm[0] = cm;
#endif
  return slurp(scratch, out, cap);
}

// Decode the same op script from `payload` (written to the scratch file first).  result[i] = decoded
// bin (kind 0/1) or bin pattern (kind 2).  Returns a status word: bit0 terminate bin != 1,
// bit1 stop bit misplaced (the two asserts of CABAC_ArithmeticDecoder.cpp:78-81, evaluated by hand so
// that a malformed stream does not abort the test process).
long ref_decode_ops(const uint8_t *payload, size_t len, const uint8_t *kind, const uint32_t *arg,
                    uint32_t nops, const uint8_t *ctx_init, int nctx, const char *scratch,
                    uint32_t *result, uint32_t *trace, uint8_t *ctx_final) {
  {
    FILE *f = std::fopen(scratch, "wb");
    if (!f) return -1;
    std::fwrite(payload, 1, len, f);
    std::fclose(f);
  }
  std::vector<ContextModel> cm;
  init_models(cm, ctx_init, nctx);
  CABAC_BitstreamFile bs;
  if (!bs.openInputFile(scratch)) return -1;
  DecPeek dec(&bs);
  dec.start();
  for (uint32_t i = 0; i < nops; i++) {
    unsigned int r = 0;
    switch (kind[i]) {
      case 0: {
#if RWTH_TRACE_CABAC_STATES
        // SimpleCABACMex.cpp:316-328 -- note that the gateway records the bin variable BEFORE decoding (always 0)
        ContextModel *m = &cm[arg[i]];
        uint8_t bin = (uint8_t)r;
        uint8_t mps_p = m->getMps();
        uint8_t state_p = m->getState();
#endif
        dec.decodeBin(r, &cm[arg[i]]);
#if RWTH_TRACE_CABAC_STATES
        uint8_t mps_a = m->getMps();
        uint8_t state_a = m->getState();
        m->addCabacStep(bin, state_p, mps_p, state_a, mps_a);
#endif
        break;
      }
      case 1: dec.decodeBinEP(r); break;
      case 2: dec.decodeBinsEP(r, (int)arg[i]); break;
#if RWTH_CABAC_FIXED_PROBABILITY
      case 3: dec.decodeBinProb(r, arg[i]); break;
#endif
      default: return -3;
    }
    result[i] = r;
    if (trace) dec.snapshot(trace + 3 * (size_t)i);
  }
  unsigned int trm = 0;
  dec.decodeBinTrm(trm);
  long status = 0;
  if (trm != 1) status |= 1;
  if (((bs.getLastByteRead() << (8 + dec.bitsNeeded())) & 0xff) != 0x80) status |= 2;
  bs.closeFile();
  if (ctx_final)
    for (int c = 0; c < nctx; c++) ctx_final[c] = (uint8_t)((cm[c].getState() << 1) | cm[c].getMps());
#if RWTH_TRACE_CABAC_STATES
  g_last_cm[1] = cm;
#endif
  return status;
}

// which of the reference's compile-time switches this build has on
int ref_has_fixed_prob(void) { return RWTH_CABAC_FIXED_PROBABILITY; }
int ref_has_trace(void) { return RWTH_TRACE_CABAC_STATES; }

#if RWTH_TRACE_CABAC_STATES
// What getEncoderStats / getDecoderStats hand to MATLAB for context `ctx` (SimpleCABACMex.cpp:356-466) after the most
// recent ref_encode_ops (decoder = 0) / ref_decode_ops (decoder = 1): trace5 = 5 bytes per step [codedBin, state_p,
// mps_p, state_a, mps_a] (states remapped by ContextModel::addCabacStep), trans = 128 x 128 counts [state_p][state_a].
// Returns the number of steps (the caller's trace5 holds cap_steps of them), -1 for a bad context index.
long ref_get_trace(int decoder, int ctx, uint8_t *trace5, size_t cap_steps, uint32_t *trans) {
  std::vector<ContextModel> &cm = g_last_cm[decoder ? 1 : 0];
  if (ctx < 0 || (size_t)ctx >= cm.size()) return -1;
  std::vector<CABACStep> *steps = cm[ctx].getCabacSteps();
  std::vector<std::vector<unsigned int>> *tt = cm[ctx].getCabactTransitions();
  for (size_t e = 0; e < steps->size() && e < cap_steps; e++) {
    CABACStep t = steps->at(e);
    trace5[e * 5 + 0] = t.codedBin;
    trace5[e * 5 + 1] = t.state_p;
    trace5[e * 5 + 2] = t.mps_p;
    trace5[e * 5 + 3] = t.state_a;
    trace5[e * 5 + 4] = t.mps_a;
  }
  if (trans)
    for (int p = 0; p < RWTH_TRACE_CABAC_STATES_NUM_STATES; p++)
      for (int a = 0; a < RWTH_TRACE_CABAC_STATES_NUM_STATES; a++)
        trans[p * RWTH_TRACE_CABAC_STATES_NUM_STATES + a] = (*tt)[p][a];
  return (long)steps->size();
}
#endif

// Stream-major batch round trip with one scratch file per stream (what the reference does, one file per
// handle: SimpleCABACMex.cpp:195,288).  ctx byte 0xFF = EP bin.  Used only for timing the reference.
// mode bit0 = encode, bit1 = decode (decode needs encode in the same call).  Returns payload bytes
// summed over streams, or <0 on error / round-trip mismatch.
long ref_roundtrip_batch(const uint8_t *bins, const uint8_t *ctx, uint64_t S, uint32_t B,
                         const uint8_t *ctx_init, int nctx, const char *scratch, int mode,
                         double *t_enc, double *t_dec);

// raw tables, for the exhaustive table check
uint8_t ref_next_state(uint8_t packed, int lps) {
  ContextModel m;
  m.init(packed & 1u, packed >> 1);
  if (lps) m.updateLPS(); else m.updateMPS();
  return (uint8_t)((m.getState() << 1) | m.getMps());
}

}  // extern "C"

#include <chrono>

extern "C" long ref_roundtrip_batch(const uint8_t *bins, const uint8_t *ctx, uint64_t S, uint32_t B,
                                    const uint8_t *ctx_init, int nctx, const char *scratch, int mode,
                                    double *t_enc, double *t_dec) {
  using clk = std::chrono::steady_clock;
  std::vector<ContextModel> cm;
  long total = 0;
  double te = 0, td = 0;
  for (uint64_t s = 0; s < S; s++) {
    const uint8_t *b = bins + s * B, *c = ctx + s * B;
    auto t0 = clk::now();
    {
      init_models(cm, ctx_init, nctx);
      CABAC_BitstreamFile bs;
      if (!bs.openOutputFile(scratch)) return -1;
      CABAC_ArithmeticEncoder enc(&bs);
      enc.start();
      for (uint32_t i = 0; i < B; i++) {
        if (c[i] == 0xFF) enc.encodeBinEP(b[i]);
        else enc.encodeBin(b[i], &cm[c[i]]);
      }
      enc.finish();
      total += (long)(bs.getNumberOfWrittenBits() / 8);
      bs.closeFile();
    }
    auto t1 = clk::now();
    te += std::chrono::duration<double>(t1 - t0).count();
    if (mode & 2) {
      init_models(cm, ctx_init, nctx);
      CABAC_BitstreamFile bs;
      if (!bs.openInputFile(scratch)) return -1;
      CABAC_ArithmeticDecoder dec(&bs);
      dec.start();
      int bad = 0;
      for (uint32_t i = 0; i < B; i++) {
        unsigned int r = 0;
        if (c[i] == 0xFF) dec.decodeBinEP(r);
        else dec.decodeBin(r, &cm[c[i]]);
        bad |= (r != b[i]);
      }
      dec.finish();
      bs.closeFile();
      if (bad) return -4;
      td += std::chrono::duration<double>(clk::now() - t1).count();
    }
  }
  if (t_enc) *t_enc = te;
  if (t_dec) *t_dec = td;
  return total;
}
