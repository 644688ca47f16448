/*
 * cabac_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cabac_oracle.h).
 *
 * CPU restatement of the reference CABAC path.  Written from the algorithm, not
 * copied: tables are generated from the H.264/H.265 definitions (rangeTabLps
 * ITU-T H.265 Table 9-46, transIdxLps Table 9-45) and cross-checked against the
 * compiled reference (tests/test_oracle_vs_ref.py).  Citations: /root/reference/.
 */
#include "cabac_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------- */

/* rangeTabLps[state][q], one little-endian dword per state (byte q = column q).
 * Same numbers as sm_aucLPSTable, CABAC/CABAC_ArithmeticEncoder.cpp:414-480 and its
 * byte-identical copy CABAC_ArithmeticDecoder.cpp:474-540. */
static const uint32_t k_lps_row[64] = {
    0xF0D0B080u, 0xE3C5A780u, 0xD8BB9E80u, 0xCDB2967Bu, 0xC3A98E74u, 0xB9A0876Fu, 0xAF988069u,
    0xA6907A64u, 0x9E89745Fu, 0x96826E5Au, 0x8E7B6855u, 0x87756351u, 0x806F5E4Du, 0x7A695949u,
    0x74645545u, 0x6E5F5042u, 0x685A4C3Eu, 0x6356483Bu, 0x5E514538u, 0x594D4135u, 0x55493E33u,
    0x50453B30u, 0x4C42382Eu, 0x483F352Bu, 0x453B3229u, 0x41383027u, 0x3E362D25u, 0x3B332B23u,
    0x38302921u, 0x352E2720u, 0x322B251Eu, 0x3029231Du, 0x2D27211Bu, 0x2B251F1Au, 0x29231E18u,
    0x27211C17u, 0x25201B16u, 0x231E1A15u, 0x211D1814u, 0x1F1B1713u, 0x1E1A1612u, 0x1C191511u,
    0x1B171410u, 0x1916130Fu, 0x1815120Eu, 0x1714110Eu, 0x1613100Du, 0x15120F0Cu, 0x14110E0Cu,
    0x13100E0Bu, 0x120F0D0Bu, 0x110F0C0Au, 0x100E0C0Au, 0x0F0D0B09u, 0x0E0C0B09u, 0x0E0C0A08u,
    0x0D0B0908u, 0x0C0B0907u, 0x0C0A0907u, 0x0B0A0807u, 0x0B090806u, 0x0A090706u, 0x09080706u,
    0x02020202u};

/* transIdxLps[state]; state 63 maps to itself (the reference's 128-entry table keeps
 * packed 126/127 fixed, ContextModel.cpp:156-157). */
static const uint8_t k_trans_lps[64] = {
    0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
    18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
    31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

uint8_t orc_lps_table(uint32_t state, uint32_t q) {
  return (uint8_t)(k_lps_row[state & 63] >> (8 * (q & 3)));
}

/* sm_aucRenormTable (CABAC_ArithmeticEncoder.cpp:482-492): shift that brings an LPS
 * range back into [256,512); index is lps >> 3.  6 for index 0, else 5 - floor(log2(i)). */
uint8_t orc_renorm_table(uint32_t i) {
  i &= 31;
  if (i == 0) return 6;
  int lg = 0;
  while ((i >> (lg + 1)) != 0) lg++;
  return (uint8_t)(5 - lg);
}

uint8_t orc_ctx_pack(uint32_t mps, uint32_t state) { return (uint8_t)((state << 1) + mps); }

/* m_aucNextStateMPS (ContextModel.cpp:136-146): state+1, saturating at 62; 63 stays. */
uint8_t orc_ctx_next_mps(uint8_t p) {
  uint32_t state = p >> 1, mps = p & 1;
  if (state < 62) state++;
  return (uint8_t)((state << 1) | mps);
}

/* m_aucNextStateLPS (ContextModel.cpp:148-158): transIdxLps, MPS flips only at state 0. */
uint8_t orc_ctx_next_lps(uint8_t p) {
  uint32_t state = p >> 1, mps = p & 1;
  if (state == 0) mps ^= 1;
  return (uint8_t)((k_trans_lps[state] << 1) | mps);
}

/* ------------------------------------------------------------------------------------------
 * p(0) -> state   (CABAC_ContextModelsInit.cpp:124-148)
 * ---------------------------------------------------------------------------------------- */
uint8_t orc_p0_to_packed(double p0) {
  double p_lps;
  uint32_t mps;
  if (p0 >= 0.5) { p_lps = 1.0 - p0; mps = 0; }
  else           { p_lps = p0;       mps = 1; }
  if (p_lps < 0.01875) p_lps = 0.01875;
  int state = (int)round(62.0 * log10(2.0 * p_lps) / log10(2.0 * 0.01875));
  if (state < 0) state = 0;
  if (state > 62) state = 62;
  return orc_ctx_pack(mps, (uint32_t)state);
}

void orc_init_by_prob(const double *p0, int nctx, uint8_t *out) {
  for (int c = 0; c < nctx; c++) out[c] = orc_p0_to_packed(p0[c]);
}

void orc_init_by_state(const double *t, int nctx, uint8_t *out) {
  /* CABAC_ContextModelsInit.cpp:70-74: slot 3c is ignored, 3c+1 = mps, 3c+2 = state */
  for (int c = 0; c < nctx; c++)
    out[c] = orc_ctx_pack((uint32_t)t[3 * c + 1], (uint32_t)t[3 * c + 2]);
}

/* ------------------------------------------------------------------------------------------
 * Sink  (CABAC_BitstreamFile.cpp:95-151)
 * ---------------------------------------------------------------------------------------- */
void orc_sink_init(orc_sink *s, uint8_t *buf, size_t cap) {
  memset(s, 0, sizeof *s);
  s->buf = buf;
  s->cap = cap;
}

static void sink_put(orc_sink *s, uint32_t byte) {
  if (s->len < s->cap) s->buf[s->len] = (uint8_t)byte;
  else s->overflow = 1;
  s->len++;
  s->bits_written += 8;
}

void orc_sink_write(orc_sink *s, uint32_t bits, uint32_t nbits) {
  uint32_t total = nbits + s->num_held;
  uint32_t next_held = total & 7;
  /* the new low-order bits that do not complete a byte, msb-aligned in a byte */
  uint32_t next_held_bits = (bits << (8 - next_held)) & 0xFF;
  uint32_t nbytes = total >> 3;
  if (nbytes == 0) {
    s->held_bits |= next_held_bits;
    s->num_held = next_held;
    return;
  }
  /* held bits go in front of the (nbits - next_held) bits that are written now */
  uint32_t shift_in = (nbits - next_held) & ~7u;
  uint32_t word = (s->held_bits << shift_in) | (bits >> next_held);
  for (uint32_t b = nbytes; b-- > 0;) sink_put(s, (word >> (8 * b)) & 0xFF);
  s->held_bits = next_held_bits;
  s->num_held = next_held;
}

void orc_sink_align_zero(orc_sink *s) {
  if (s->num_held == 0) return;
  sink_put(s, s->held_bits);
  s->held_bits = 0;
  s->num_held = 0;
}

uint64_t orc_sink_bits(const orc_sink *s) { return s->bits_written + s->num_held; }

/* ------------------------------------------------------------------------------------------
 * Encoder  (CABAC_ArithmeticEncoder.cpp)
 * ---------------------------------------------------------------------------------------- */
void orc_enc_start(orc_encoder *e, uint8_t *buf, size_t cap) {
  e->low = 0;
  e->range = 510;
  e->bits_left = 23;
  e->num_buffered = 0;
  e->buffered_byte = 0xFF;
  e->bins_coded = 0;
  orc_sink_init(&e->sink, buf, cap);
}

static _Thread_local orc_counters g_cnt; /* per thread: the all-core baseline must not share a cache line */
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
const orc_counters *orc_counters_get(void) { return &g_cnt; }

/* writeOut, :380-412 */
static void enc_write_out(orc_encoder *e) {
  uint32_t lead = e->low >> (24 - e->bits_left);
  e->bits_left += 8;
  e->low &= 0xFFFFFFFFu >> e->bits_left;
  g_cnt.writeouts++;
  if (lead == 0xFF) {
    e->num_buffered++;
    g_cnt.ff_leads++;
    if ((uint64_t)e->num_buffered > g_cnt.max_num_buffered) g_cnt.max_num_buffered = (uint64_t)e->num_buffered;
    return;
  }
  if (e->num_buffered > 0) {
    uint32_t carry = lead >> 8;
    g_cnt.carry_resolves += carry;
    uint32_t byte = e->buffered_byte + carry;
    e->buffered_byte = lead & 0xFF;
    orc_sink_write(&e->sink, byte, 8);
    byte = (0xFF + carry) & 0xFF;
    while (e->num_buffered > 1) {
      orc_sink_write(&e->sink, byte, 8);
      e->num_buffered--;
    }
  } else {
    e->num_buffered = 1;
    e->buffered_byte = lead;
  }
}

static void enc_test_write_out(orc_encoder *e) { /* :369-375 */
  if (e->bits_left < 12) enc_write_out(e);
}

void orc_enc_bin(orc_encoder *e, uint32_t bin, uint8_t *ctx) {
  uint32_t p = *ctx;
  e->bins_coded++;
  uint32_t lps = orc_lps_table(p >> 1, (e->range >> 6) & 3);
  e->range -= lps;
  if (bin != (p & 1)) {
    int n = orc_renorm_table(lps >> 3);
    e->low = (e->low + e->range) << n;
    e->range = lps << n;
    *ctx = orc_ctx_next_lps((uint8_t)p);
    e->bits_left -= n;
  } else {
    *ctx = orc_ctx_next_mps((uint8_t)p);
    if (e->range >= 256) return;
    e->low <<= 1;
    e->range <<= 1;
    e->bits_left--;
  }
  enc_test_write_out(e);
}

void orc_enc_bin_ep(orc_encoder *e, uint32_t bin) {
  e->bins_coded++;
  e->low <<= 1;
  if (bin) e->low += e->range;
  e->bits_left--;
  enc_test_write_out(e);
}

/* sm_aucLPSTProbTable[49][4], CABAC_ArithmeticEncoder.cpp:494-547 (= CABAC_ArithmeticDecoder.cpp:554-): one dword per
 * row, byte q = column q.  Row i serves an MPS probability of 51 + i percent. */
static const uint32_t k_prob_row[49] = {
  0xEACAAB8Bu, 0xE5C6A788u, 0xE0C2A486u, 0xDCBEA083u, 0xD7BA9D80u, 0xD2B6997Du, 0xCDB2967Au,
  0xC9AE9277u, 0xC4A98F74u, 0xBFA58B72u, 0xBAA1886Fu, 0xB59D846Cu, 0xB1998169u, 0xAC957E66u,
  0xA7917A63u, 0xA28C7761u, 0x9E88735Eu, 0x9984705Bu, 0x94806C58u, 0x8F7C6955u, 0x8A786552u,
  0x86746250u, 0x81705E4Du, 0x7C6B5B4Au, 0x77675747u, 0x73635444u, 0x6E5F5041u, 0x695B4D3Eu,
  0x6457493Cu, 0x5F534639u, 0x5B4E4236u, 0x564A3F33u, 0x51463B30u, 0x4C42382Du, 0x483E342Bu,
  0x433A3128u, 0x3E362D25u, 0x39322A22u, 0x352D261Fu, 0x3029231Cu, 0x2B251F1Au, 0x26211C17u,
  0x211D1814u, 0x1D191511u, 0x1815110Eu, 0x13110E0Bu, 0x0E0C0A09u, 0x0A080706u, 0x05040404u};
uint8_t orc_prob_table(uint32_t row, uint32_t q) { return (uint8_t)(k_prob_row[row] >> (8 * q)); }

/* encodeBinProb, CABAC_ArithmeticEncoder.cpp:180-243 (compiled only under RWTH_CABAC_FIXED_PROBABILITY).
 * prob = probability of a ONE in percent, 1..99; 50 is an EP bin. */
void orc_enc_bin_prob(orc_encoder *e, uint32_t bin, uint32_t prob) {
  if (prob == 50) {
    orc_enc_bin_ep(e, bin);
    return;
  }
  e->bins_coded++;
  uint32_t mps = prob > 50 ? 1 : 0;
  uint32_t pm = prob < 50 ? 50 - prob : prob - 50;
  uint32_t lps = orc_prob_table(pm - 1, (e->range >> 6) & 3);
  e->range -= lps;
  if (bin != mps) {
    int n = orc_renorm_table(lps >> 3);
    e->low = (e->low + e->range) << n;
    e->range = lps << n;
    e->bits_left -= n;
  } else {
    if (e->range >= 256) return;
    e->low <<= 1;
    e->range <<= 1;
    e->bits_left--;
  }
  enc_test_write_out(e);
}

void orc_enc_bins_ep(orc_encoder *e, uint32_t bins, int nbins) {
  e->bins_coded += (uint32_t)nbins;
  while (nbins > 8) {
    nbins -= 8;
    uint32_t pattern = bins >> nbins;
    e->low <<= 8;
    e->low += e->range * pattern;
    bins -= pattern << nbins;
    e->bits_left -= 8;
    enc_test_write_out(e);
  }
  e->low <<= nbins;
  e->low += e->range * bins;
  e->bits_left -= nbins;
  enc_test_write_out(e);
}

void orc_enc_bin_trm(orc_encoder *e, uint32_t bin) {
  e->bins_coded++;
  e->range -= 2;
  if (bin) {
    e->low += e->range;
    e->low <<= 7;
    e->range = 2 << 7;
    e->bits_left -= 7;
  } else if (e->range >= 256) {
    return;
  } else {
    e->low <<= 1;
    e->range <<= 1;
    e->bits_left--;
  }
  enc_test_write_out(e);
}

void orc_enc_finish(orc_encoder *e) {
  orc_enc_bin_trm(e, 1);
  if (e->low >> (32 - e->bits_left)) {
    g_cnt.finish_carries++;
    orc_sink_write(&e->sink, e->buffered_byte + 1, 8);
    while (e->num_buffered > 1) {
      orc_sink_write(&e->sink, 0x00, 8);
      e->num_buffered--;
    }
    e->low -= 1u << (32 - e->bits_left);
  } else {
    if (e->num_buffered > 0) orc_sink_write(&e->sink, e->buffered_byte, 8);
    while (e->num_buffered > 1) {
      orc_sink_write(&e->sink, 0xFF, 8);
      e->num_buffered--;
    }
  }
  orc_sink_write(&e->sink, e->low >> 8, (uint32_t)(24 - e->bits_left));
  orc_sink_write(&e->sink, 1, 1);
  orc_sink_align_zero(&e->sink);
}

size_t orc_enc_bytes(const orc_encoder *e) { return e->sink.len; }
uint64_t orc_enc_bits(const orc_encoder *e) { return orc_sink_bits(&e->sink); }

/* ------------------------------------------------------------------------------------------
 * Decoder  (CABAC_ArithmeticDecoder.cpp; byte source = CABAC_BitstreamFile.cpp:153-158)
 * ---------------------------------------------------------------------------------------- */
static uint32_t dec_read_byte(orc_decoder *d) {
  uint32_t b;
  if (d->pos < d->len) b = d->buf[d->pos];
  else { b = 0xFF; d->overreads++; } /* fstream::get() at EOF -> (unsigned char)EOF */
  d->pos++;
  d->last_byte = b;
  return b;
}

void orc_dec_start(orc_decoder *d, const uint8_t *buf, size_t len) {
  d->buf = buf;
  d->len = len;
  d->pos = 0;
  d->last_byte = 0;
  d->overreads = 0;
  d->range = 510;
  d->bits_needed = -8;
  d->value = dec_read_byte(d) << 8;
  d->value |= dec_read_byte(d);
}

uint32_t orc_dec_bin(orc_decoder *d, uint8_t *ctx) {
  uint32_t p = *ctx, bin;
  uint32_t lps = orc_lps_table(p >> 1, (d->range >> 6) - 4);
  d->range -= lps;
  uint32_t scaled = d->range << 7;
  if (d->value < scaled) {
    bin = p & 1;
    *ctx = orc_ctx_next_mps((uint8_t)p);
    if (scaled >= (256u << 7)) return bin;
    d->range = scaled >> 6;
    d->value += d->value;
    if (++d->bits_needed == 0) {
      d->bits_needed = -8;
      d->value += dec_read_byte(d);
    }
  } else {
    int n = orc_renorm_table(lps >> 3);
    d->value = (d->value - scaled) << n;
    d->range = lps << n;
    bin = 1 - (p & 1);
    *ctx = orc_ctx_next_lps((uint8_t)p);
    d->bits_needed += n;
    if (d->bits_needed >= 0) {
      d->value += dec_read_byte(d) << d->bits_needed;
      d->bits_needed -= 8;
    }
  }
  return bin;
}

static uint32_t orc_dec_bin_ep_fwd(orc_decoder *d);
/* decodeBinProb, CABAC_ArithmeticDecoder.cpp:192-286 */
uint32_t orc_dec_bin_prob(orc_decoder *d, uint32_t prob) {
  if (prob == 50) return orc_dec_bin_ep_fwd(d);
  uint32_t mps = prob > 50 ? 1 : 0, bin;
  uint32_t pm = prob < 50 ? 50 - prob : prob - 50;
  uint32_t lps = orc_prob_table(pm - 1, (d->range >> 6) & 3);
  d->range -= lps;
  uint32_t scaled = d->range << 7;
  if (d->value < scaled) {
    bin = mps;
    if (scaled >= (256u << 7)) return bin;
    d->range = scaled >> 6;
    d->value += d->value;
    if (++d->bits_needed == 0) {
      d->bits_needed = -8;
      d->value += dec_read_byte(d);
    }
  } else {
    int n = orc_renorm_table(lps >> 3);
    d->value = (d->value - scaled) << n;
    d->range = lps << n;
    bin = 1 - mps;
    d->bits_needed += n;
    if (d->bits_needed >= 0) {
      d->value += dec_read_byte(d) << d->bits_needed;
      d->bits_needed -= 8;
    }
  }
  return bin;
}

uint32_t orc_dec_bin_ep(orc_decoder *d) {
  d->value += d->value;
  if (++d->bits_needed >= 0) {
    d->bits_needed = -8;
    d->value += dec_read_byte(d);
  }
  uint32_t scaled = d->range << 7;
  if (d->value >= scaled) {
    d->value -= scaled;
    return 1;
  }
  return 0;
}
static uint32_t orc_dec_bin_ep_fwd(orc_decoder *d) { return orc_dec_bin_ep(d); }

uint32_t orc_dec_bins_ep(orc_decoder *d, int nbins) {
  uint32_t bins = 0;
  while (nbins > 8) {
    d->value = (d->value << 8) + (dec_read_byte(d) << (8 + d->bits_needed));
    uint32_t scaled = d->range << 15;
    for (int i = 0; i < 8; i++) {
      bins += bins;
      scaled >>= 1;
      if (d->value >= scaled) { bins++; d->value -= scaled; }
    }
    nbins -= 8;
  }
  d->bits_needed += nbins;
  d->value <<= nbins;
  if (d->bits_needed >= 0) {
    d->value += dec_read_byte(d) << d->bits_needed;
    d->bits_needed -= 8;
  }
  uint32_t scaled = d->range << (nbins + 7);
  for (int i = 0; i < nbins; i++) {
    bins += bins;
    scaled >>= 1;
    if (d->value >= scaled) { bins++; d->value -= scaled; }
  }
  return bins;
}

uint32_t orc_dec_bin_trm(orc_decoder *d) {
  d->range -= 2;
  uint32_t scaled = d->range << 7;
  if (d->value >= scaled) return 1;
  if (scaled < (256u << 7)) {
    d->range = scaled >> 6;
    d->value += d->value;
    if (++d->bits_needed == 0) {
      d->bits_needed = -8;
      d->value += dec_read_byte(d);
    }
  }
  return 0;
}

uint32_t orc_dec_finish(orc_decoder *d) {
  uint32_t status = 0;
  if (orc_dec_bin_trm(d) != 1) status |= 1;
  if (((d->last_byte << (8 + d->bits_needed)) & 0xFF) != 0x80) status |= 2;
  if (d->overreads) status |= 4;
  return status;
}

/* ------------------------------------------------------------------------------------------
 * Whole streams and batches
 * ---------------------------------------------------------------------------------------- */
size_t orc_encode_stream(const uint8_t *bins, const uint8_t *ctx, uint32_t nbins,
                         const uint8_t *ctx_init, int nctx, uint8_t *out, size_t cap,
                         uint64_t *bits_before_finish, uint8_t *ctx_final) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm); /* ContextModel() default, ContextModel.cpp:54 */
  if (nctx > 0) memcpy(cm, ctx_init, (size_t)nctx);
  orc_encoder e;
  orc_enc_start(&e, out, cap);
  for (uint32_t i = 0; i < nbins; i++) {
    if (ctx[i] == ORC_BYPASS_CTX) orc_enc_bin_ep(&e, bins[i]);
    else if (nctx <= ORC_PROB_MAX_NCTX && ctx[i] > ORC_PROB_BASE) orc_enc_bin_prob(&e, bins[i], ctx[i] - ORC_PROB_BASE);
    else orc_enc_bin(&e, bins[i], &cm[ctx[i]]);
  }
  if (bits_before_finish) *bits_before_finish = orc_sink_bits(&e.sink);
  orc_enc_finish(&e);
  if (ctx_final && nctx > 0) memcpy(ctx_final, cm, (size_t)nctx);
  return e.sink.overflow ? 0 : e.sink.len;
}

uint32_t orc_decode_stream(const uint8_t *payload, size_t len, const uint8_t *ctx,
                           uint32_t nbins, const uint8_t *ctx_init, int nctx,
                           uint8_t *bins_out, uint8_t *ctx_final) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  if (nctx > 0) memcpy(cm, ctx_init, (size_t)nctx);
  orc_decoder d;
  orc_dec_start(&d, payload, len);
  for (uint32_t i = 0; i < nbins; i++) {
    if (ctx[i] == ORC_BYPASS_CTX) bins_out[i] = (uint8_t)orc_dec_bin_ep(&d);
    else if (nctx <= ORC_PROB_MAX_NCTX && ctx[i] > ORC_PROB_BASE) bins_out[i] = (uint8_t)orc_dec_bin_prob(&d, ctx[i] - ORC_PROB_BASE);
    else bins_out[i] = (uint8_t)orc_dec_bin(&d, &cm[ctx[i]]);
  }
  if (ctx_final && nctx > 0) memcpy(ctx_final, cm, (size_t)nctx);
  return orc_dec_finish(&d);
}

uint64_t orc_encode_batch(const uint8_t *bins, const uint8_t *ctx, const uint32_t *nbins,
                          uint64_t S, uint32_t B, const uint8_t *ctx_init, int nctx,
                          uint8_t *payload, uint64_t cap, uint64_t *offsets) {
  uint64_t off = 0;
  for (uint64_t s = 0; s < S; s++) {
    uint32_t n = nbins ? nbins[s] : B;
    offsets[s] = off;
    size_t len = orc_encode_stream(bins + s * B, ctx + s * B, n, ctx_init, nctx,
                                   payload + off, (size_t)(cap - off), NULL, NULL);
    if (len == 0) return (uint64_t)-1;
    off += len;
  }
  offsets[S] = off;
  return off;
}

void orc_decode_batch(const uint8_t *payload, const uint64_t *offsets, const uint8_t *ctx,
                      const uint32_t *nbins, uint64_t S, uint32_t B, const uint8_t *ctx_init,
                      int nctx, uint8_t *bins_out, uint32_t *status) {
  for (uint64_t s = 0; s < S; s++) {
    uint32_t n = nbins ? nbins[s] : B;
    uint32_t st = orc_decode_stream(payload + offsets[s], (size_t)(offsets[s + 1] - offsets[s]),
                                    ctx + s * B, n, ctx_init, nctx, bins_out + s * B, NULL);
    if (status) status[s] = st;
  }
}

/* ------------------------------------------------------------------------------------------
 * Binarisation  (cabacBinarizer.m / cabacDebinarizer.m)
 * ---------------------------------------------------------------------------------------- */
int orc_binarize_tu(uint32_t v, uint32_t max_val, uint8_t *bins) {
  /* TUCode, cabacBinarizer.m:30-37: v ones, then a zero unless v == max_val */
  int n = 0;
  if (v == max_val) {
    for (uint32_t i = 0; i < max_val; i++) bins[n++] = 1;
  } else {
    for (uint32_t i = 0; i < v; i++) bins[n++] = 1;
    bins[n++] = 0;
  }
  return n;
}

int orc_binarize_egk(uint32_t v, uint32_t k, uint8_t *bins) {
  /* EGCode, cabacBinarizer.m:56-69 */
  uint32_t t = (v >> k) + 1; /* floor(v / 2^k) + 1 ; floor(log2(v/2^k + 1)) == floor(log2(t)) */
  int lg = 0;
  while ((t >> (lg + 1)) != 0) lg++;
  int n_p = lg + 1;
  int n_s = (int)k + n_p - 1;
  uint32_t v_s = v - ((((uint32_t)1 << (n_p - 1)) - 1) << k);
  int n = 0;
  for (int i = 0; i < n_p - 1; i++) bins[n++] = 1;
  bins[n++] = 0;
  for (int i = n_s - 1; i >= 0; i--) bins[n++] = (uint8_t)((v_s >> i) & 1); /* FLCode, :71-75 */
  return n;
}

int orc_binarize_fl(uint32_t v, int nbits, uint8_t *bins) {
  /* FLCode, cabacBinarizer.m:71-75: bit i = floor(v / 2^(nbits-i)) mod 2, MSB first ('DEC2FL32': nbits = 32) */
  for (int i = 0; i < nbits; i++) bins[i] = (uint8_t)((nbits - 1 - i) < 32 ? (v >> (nbits - 1 - i)) & 1 : 0);
  return nbits;
}

int orc_binarize_trk(uint32_t v, uint32_t k, uint32_t max_val, uint8_t *bins) {
  /* TRCode, cabacBinarizer.m:39-54: floor(v / 2^k) ones, a zero, then k suffix bits -- the remainder MSB first, or all
   * ones when v >= max_val (the reference's escape is a TODO that leaves the suffix as initialised) */
  uint32_t n_p = (v >> k) + 1;
  int n = 0;
  for (uint32_t i = 0; i + 1 < n_p; i++) bins[n++] = 1;
  bins[n++] = 0;
  if (v >= max_val) {
    for (uint32_t i = 0; i < k; i++) bins[n++] = 1;
  } else {
    uint32_t v_s = v - ((n_p - 1) << k);
    n += orc_binarize_fl(v_s, (int)k, bins + n);
  }
  return n;
}

uint32_t orc_debinarize_tu(const uint8_t *bins, int n, uint32_t max_val) {
  /* rTUCode, cabacDebinarizer.m:28-31 */
  for (int i = 0; i < n; i++)
    if (!bins[i]) return (uint32_t)i;
  return max_val;
}

uint32_t orc_debinarize_egk(const uint8_t *bins, int n, uint32_t k) {
  /* rEGCode, cabacDebinarizer.m:46-53 */
  int n_p = 0;
  while (n_p < n && bins[n_p]) n_p++;
  n_p++; /* 1-based index of the first zero */
  uint32_t suffix = 0;
  for (int i = n_p; i < n; i++) suffix = (suffix << 1) | bins[i];
  return ((((uint32_t)1 << (n_p - 1)) - 1) << k) + suffix;
}

/* incremental end-of-symbol test, cabacDecodeSymbolFinished.m:12-30.
 * g[n-1] is the bin just decoded (n is 1-based). */
typedef struct { int n_p, n_s; } sym_state;

static int symbol_finished(const uint8_t *g, int n, uint32_t nq, int method, uint32_t k,
                           sym_state *st) {
  if (method == 0) return g[n - 1] == 0 || (uint32_t)n == nq - 1;
  if (st->n_s == -1) {
    if (g[n - 1] == 0) {
      st->n_p += n;
      st->n_s = (int)k + st->n_p - 1;
      if (st->n_s == 0) return 1;
    }
    return 0;
  }
  if (st->n_s == 1) return 1;
  st->n_s--;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Context selectors
 * ---------------------------------------------------------------------------------------- */
static int first_zero_1based(const uint8_t *g, int len) {
  for (int i = 0; i < len; i++)
    if (!g[i]) return i + 1;
  return 0; /* "isempty" */
}

int orc_iss_ctx(int n, const uint8_t *g_prev, const uint8_t *g_up1, int len_up1,
                unsigned types, int N) {
  /* cabacContextSelection.m:24-67 */
  int np_prev = first_zero_1based(g_prev, n - 1);
  int prefix_prev = !(np_prev && n > np_prev);
  int np_up1 = first_zero_1based(g_up1, len_up1);
  int prefix_up1 = !(np_up1 && n > np_up1);
  int id;
  if (prefix_prev) {
    if (n <= N) {
      id = n;
      if (len_up1 >= n && prefix_up1) {
        if (g_up1[n - 1] == 0 && (types & 1)) id = N + n;
        else if (g_up1[n - 1] == 1 && (types & 2)) id = 2 * N + n;
      } else if (n > 1 && g_prev[n - 2] == 1 && (types & 16)) {
        id = 3 * N + n - 1;
      }
    } else {
      id = 7 * N + 1;
    }
  } else {
    int j = n - np_prev;
    if (j <= N) {
      id = 4 * N + j;
      if (len_up1 >= n && !prefix_up1) {
        if (g_up1[n - 1] == 0 && (types & 4)) id = 5 * N + j;
        else if (g_up1[n - 1] == 1 && (types & 8)) id = 6 * N + j;
      }
    } else {
      id = 7 * N + 2;
    }
  }
  return id;
}

#define ORC_MAX_SYM_BINS 320 /* TR0 of 255 has 256 bins */
/* decoding: the reference has no limit on a symbol's bin string (MATLAB keeps appending), but a string longer than 32
 * bins cannot come from a uint8 symbol (EG2 of 255 has 17 bins); on corrupt streams the engine cuts a symbol at 32 bins
 * (its bin string is one 32-bit word) and the restatement does the same so that both stay in step afterwards */
#define ORC_DEC_SYM_CAP 32

static int binarize(uint32_t v, uint32_t nq, int method, uint32_t k, uint8_t *g) {
  /* method: 0 TU, 1 EGk, 2 TRk, 3 FL32 (cabacBinarizer.m:10-27) */
  if (method == 2) return orc_binarize_trk(v, k, nq - 1, g);
  if (method == 3) return orc_binarize_fl(v, 32, g);
  return method == 0 ? orc_binarize_tu(v, nq - 1, g) : orc_binarize_egk(v, k, g);
}

size_t orc_iss_encode_column(const uint32_t *sym, int rows, uint32_t nq, int method, uint32_t k,
                             unsigned types, int nlbp, const uint8_t *ctx_init, int nctx,
                             uint8_t *out, size_t cap, uint32_t *nbins_out) {
  /* one column of the (d, cnt) loops of cabacEncode.m:45-70 */
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  memcpy(cm, ctx_init, (size_t)nctx);
  uint8_t g[ORC_MAX_SYM_BINS], up[ORC_MAX_SYM_BINS];
  int len_up = 0;
  uint32_t total = 0;
  orc_encoder e;
  orc_enc_start(&e, out, cap);
  for (int d = 0; d < rows; d++) {
    int len = binarize(sym[d], nq, method, k, g);
    for (int cnt = 1; cnt <= len; cnt++) {
      int id = orc_iss_ctx(cnt, g, up, len_up, types, nlbp);
      orc_enc_bin(&e, g[cnt - 1], &cm[id - 1]); /* cabacEncode.m:64 passes id-1 */
      total++;
    }
    memcpy(up, g, (size_t)len);
    len_up = len;
  }
  orc_enc_finish(&e);
  if (nbins_out) *nbins_out = total;
  return e.sink.overflow ? 0 : e.sink.len;
}

uint32_t orc_iss_decode_column(const uint8_t *payload, size_t len, int rows, uint32_t nq,
                               int method, uint32_t k, unsigned types, int nlbp,
                               const uint8_t *ctx_init, int nctx, uint32_t *sym_out) {
  /* one column of cabacDecode.m:29-55 */
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  memcpy(cm, ctx_init, (size_t)nctx);
  uint8_t g[ORC_MAX_SYM_BINS], up[ORC_MAX_SYM_BINS];
  int len_up = 0;
  orc_decoder dec;
  orc_dec_start(&dec, payload, len);
  for (int d = 0; d < rows; d++) {
    sym_state st = {0, -1};
    int n = 1, done = 0;
    while (!done && n <= ORC_DEC_SYM_CAP) {
      int id = orc_iss_ctx(n, g, up, len_up, types, nlbp);
      g[n - 1] = (uint8_t)orc_dec_bin(&dec, &cm[id - 1]);
      done = symbol_finished(g, n, nq, method, k, &st);
      n++;
    }
    int glen = n - 1;
    sym_out[d] = !done ? 0 /* cut at the cap: see ORC_DEC_SYM_CAP */
                       : method == 0 ? orc_debinarize_tu(g, glen, nq - 1) : orc_debinarize_egk(g, glen, k);
    memcpy(up, g, (size_t)glen);
    len_up = glen;
  }
  return orc_dec_finish(&dec);
}

/* whole matrix as ONE stream: cabacEncode.m:34-70 / cabacDecode.m:29-55 (k outer, d inner); sym is column-major */
size_t orc_iss_encode_matrix(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                             unsigned types, int nlbp, const uint8_t *ctx_init, int nctx, uint8_t *out, size_t cap,
                             uint32_t *nbins_out) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  memcpy(cm, ctx_init, (size_t)nctx);
  uint8_t g[ORC_MAX_SYM_BINS], up[ORC_MAX_SYM_BINS];
  uint32_t total = 0;
  orc_encoder e;
  orc_enc_start(&e, out, cap);
  for (int c = 0; c < cols; c++) {
    int len_up = 0; /* d == 1: no upper neighbour, cabacEncode.m:52 */
    for (int d = 0; d < rows; d++) {
      int len = binarize(sym[(size_t)c * rows + d], nq, method, k, g);
      for (int cnt = 1; cnt <= len; cnt++) {
        int id = orc_iss_ctx(cnt, g, up, len_up, types, nlbp);
        orc_enc_bin(&e, g[cnt - 1], &cm[id - 1]);
        total++;
      }
      memcpy(up, g, (size_t)len);
      len_up = len;
    }
  }
  orc_enc_finish(&e);
  if (nbins_out) *nbins_out = total;
  return e.sink.overflow ? 0 : e.sink.len;
}

/* the same walk with the debugging statistics of cabacEncode.m:39-42,60-67: ctx_hist[id-1] = bins coded with context
 * id, ctx_cost[id-1] = getNumBits() growth while coding them, sym_bits[c*rows+d] = H(d,k) = getNumBits() growth over the
 * symbol (getNumBits counts the bits already written to the file, CABAC_BitstreamFile.h:70: whole bytes) */
size_t orc_iss_encode_matrix_stats(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                                   unsigned types, int nlbp, const uint8_t *ctx_init, int nctx, uint8_t *out, size_t cap,
                                   uint32_t *ctx_hist, uint32_t *ctx_cost, uint32_t *sym_bits) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  memcpy(cm, ctx_init, (size_t)nctx);
  uint8_t g[ORC_MAX_SYM_BINS], up[ORC_MAX_SYM_BINS];
  orc_encoder e;
  orc_enc_start(&e, out, cap);
  for (int c = 0; c < cols; c++) {
    int len_up = 0;
    for (int d = 0; d < rows; d++) {
      const uint64_t sym_start = orc_enc_bits(&e);
      int len = binarize(sym[(size_t)c * rows + d], nq, method, k, g);
      for (int cnt = 1; cnt <= len; cnt++) {
        int id = orc_iss_ctx(cnt, g, up, len_up, types, nlbp);
        const uint64_t before = orc_enc_bits(&e);
        orc_enc_bin(&e, g[cnt - 1], &cm[id - 1]);
        ctx_hist[id - 1]++;
        ctx_cost[id - 1] += (uint32_t)(orc_enc_bits(&e) - before);
      }
      sym_bits[(size_t)c * rows + d] = (uint32_t)(orc_enc_bits(&e) - sym_start);
      memcpy(up, g, (size_t)len);
      len_up = len;
    }
  }
  orc_enc_finish(&e);
  return e.sink.overflow ? 0 : e.sink.len;
}

uint32_t orc_iss_decode_matrix(const uint8_t *payload, size_t len, int rows, int cols, uint32_t nq, int method,
                               uint32_t k, unsigned types, int nlbp, const uint8_t *ctx_init, int nctx,
                               uint32_t *sym_out) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  memcpy(cm, ctx_init, (size_t)nctx);
  uint8_t g[ORC_MAX_SYM_BINS], up[ORC_MAX_SYM_BINS];
  orc_decoder dec;
  orc_dec_start(&dec, payload, len);
  for (int c = 0; c < cols; c++) {
    int len_up = 0;
    for (int d = 0; d < rows; d++) {
      sym_state st = {0, -1};
      int n = 1, done = 0;
      while (!done && n <= ORC_DEC_SYM_CAP) {
        int id = orc_iss_ctx(n, g, up, len_up, types, nlbp);
        g[n - 1] = (uint8_t)orc_dec_bin(&dec, &cm[id - 1]);
        done = symbol_finished(g, n, nq, method, k, &st);
        n++;
      }
      int glen = n - 1;
      sym_out[(size_t)c * rows + d] = !done ? 0 : method == 0 ? orc_debinarize_tu(g, glen, nq - 1) : orc_debinarize_egk(g, glen, k);
      memcpy(up, g, (size_t)glen);
      len_up = glen;
    }
  }
  return orc_dec_finish(&dec);
}

/* cabacDemo.m:113-121 (and :158-166): bin 1 of symbol it>1 uses ctx 1 / 2 when the first bin
 * of the previous symbol was 1 / 0; everything else uses ctx 0. */
static int demo_ctx(int n, int have_prev, uint8_t prev_first) {
  if (n == 1 && have_prev) return prev_first == 1 ? 1 : 2;
  return 0;
}

size_t orc_demo_encode(const uint32_t *sym, int nsym, uint32_t nq, const uint8_t *ctx_init,
                       uint8_t *out, size_t cap, uint32_t *nbins_out) {
  uint8_t cm[3];
  memcpy(cm, ctx_init, 3);
  uint8_t g[ORC_MAX_SYM_BINS], prev_first = 0;
  uint32_t total = 0;
  orc_encoder e;
  orc_enc_start(&e, out, cap);
  for (int it = 0; it < nsym; it++) {
    int len = orc_binarize_tu(sym[it], nq - 1, g);
    for (int n = 1; n <= len; n++) {
      orc_enc_bin(&e, g[n - 1], &cm[demo_ctx(n, it > 0, prev_first)]);
      total++;
    }
    prev_first = g[0];
  }
  orc_enc_finish(&e);
  if (nbins_out) *nbins_out = total;
  return e.sink.overflow ? 0 : e.sink.len;
}

uint32_t orc_demo_decode(const uint8_t *payload, size_t len, int nsym, uint32_t nq,
                         const uint8_t *ctx_init, uint32_t *sym_out) {
  uint8_t cm[3];
  memcpy(cm, ctx_init, 3);
  uint8_t g[ORC_MAX_SYM_BINS], prev_first = 0;
  orc_decoder d;
  orc_dec_start(&d, payload, len);
  for (int it = 0; it < nsym; it++) {
    sym_state st = {0, -1};
    int n = 1, done = 0;
    while (!done) {
      g[n - 1] = (uint8_t)orc_dec_bin(&d, &cm[demo_ctx(n, it > 0, prev_first)]);
      done = symbol_finished(g, n, nq, 0, 0, &st);
      n++;
    }
    sym_out[it] = orc_debinarize_tu(g, n - 1, nq - 1);
    prev_first = g[0];
  }
  return orc_dec_finish(&d);
}

/* ------------------------------------------------------------------------------------------
 * Empirical context initialisation, ISS/+coder/cabacInitContextModel.m:1-130 (MATLAB only: restated from the source
 * text, not pinned against a run of the reference -- there is no MATLAB here).  sym is the column-major rows x cols
 * matrix; out has 7*nlbp+2 entries in the order of the reference's "Concatenate" line (:128).
 * MATLAB details kept: the row above the first row is a NaN cell (length 1, np_up1 = 0, compares false with 0 and 1);
 * np = position of the first zero, or the string length if there is none; the norms of pcs0/pcs1 test y(n), not
 * y(n + np_up1) (:96,:108); empty selections give 0, a zero norm count gives norm = 1.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t b[ORC_MAX_SYM_BINS]; int len, np; int is_nan; } bin_cell;

static double ratio_or_zero(uint64_t num, uint64_t den) { return den ? (double)num / (double)den : 0.0; }

void orc_iss_init_context_model(const uint32_t *sym, int rows, int cols, uint32_t nq, int method, uint32_t k,
                                unsigned types, int N, double *out) {
  const int total = 7 * N + 2;
  for (int i = 0; i < total; i++) out[i] = 0.0;
  double *p_pre = out, *pc0 = out + N, *pc1 = out + 2 * N, *p_lft = out + 3 * N, *p_suf = out + 4 * N,
         *pcs0 = out + 5 * N, *pcs1 = out + 6 * N;
  for (int n = 1; n <= N; n++) {
    uint64_t a_pre = 0, z_pre = 0, d_c = 0, j_c0 = 0, y0_c = 0, j_c1 = 0, y1_c = 0, d_b = 0, j_b = 0, x1_b = 0,
             d_s = 0, z_s = 0, d_cs = 0, j_cs0 = 0, y0_cs = 0, j_cs1 = 0, y1_cs = 0;
    for (int c = 0; c < cols; c++) {
      bin_cell up, cur;
      up.is_nan = 1; up.len = 1; up.np = 0; up.b[0] = 0xEE;
      for (int d = 0; d < rows; d++) {
        cur.is_nan = 0;
        cur.len = binarize(sym[(size_t)c * rows + d], nq, method, k, cur.b);
        cur.np = first_zero_1based(cur.b, cur.len);
        if (!cur.np) cur.np = cur.len;
        /* y(i) of the upper cell: NaN compares false with everything */
#define UPBIT_IS(i, val) (!up.is_nan && up.b[(i) - 1] == (val))
        if (n <= cur.np) { a_pre++; z_pre += cur.b[n - 1] == 0; }
        if (n <= cur.np && n <= up.np) {
          d_c++;
          j_c0 += cur.b[n - 1] == 0 && UPBIT_IS(n, 0);
          y0_c += UPBIT_IS(n, 0);
          j_c1 += cur.b[n - 1] == 0 && UPBIT_IS(n, 1);
          y1_c += UPBIT_IS(n, 1);
        }
        if (n + 1 <= cur.np && up.np < n + 1) {
          d_b++;
          j_b += cur.b[n] == 0 && cur.b[n - 1] == 1;
          x1_b += cur.b[n - 1] == 1;
        }
        if (n + cur.np <= cur.len) { d_s++; z_s += cur.b[n + cur.np - 1] == 0; }
        if (n + cur.np <= cur.len && n + up.np <= up.len) {
          d_cs++;
          j_cs0 += cur.b[n + cur.np - 1] == 0 && UPBIT_IS(n + up.np, 0);
          y0_cs += UPBIT_IS(n, 0);
          j_cs1 += cur.b[n + cur.np - 1] == 0 && UPBIT_IS(n + up.np, 1);
          y1_cs += UPBIT_IS(n, 1);
        }
#undef UPBIT_IS
        up = cur;
      }
    }
    p_pre[n - 1] = ratio_or_zero(z_pre, a_pre);
    if (types & 1) pc0[n - 1] = ratio_or_zero(j_c0, d_c) / ((d_c && y0_c) ? (double)y0_c / (double)d_c : 1.0);
    if (types & 2) pc1[n - 1] = ratio_or_zero(j_c1, d_c) / ((d_c && y1_c) ? (double)y1_c / (double)d_c : 1.0);
    if (types & 16) p_lft[n - 1] = ratio_or_zero(j_b, d_b) / ((d_b && x1_b) ? (double)x1_b / (double)d_b : 1.0);
    p_suf[n - 1] = ratio_or_zero(z_s, d_s);
    if (types & 4) pcs0[n - 1] = ratio_or_zero(j_cs0, d_cs) / ((d_cs && y0_cs) ? (double)y0_cs / (double)d_cs : 1.0);
    if (types & 8) pcs1[n - 1] = ratio_or_zero(j_cs1, d_cs) / ((d_cs && y1_cs) ? (double)y1_cs / (double)d_cs : 1.0);
  }
  /* the rest contexts, :112-126 */
  uint64_t t_rp = 0, z_rp = 0, t_rs = 0, z_rs = 0;
  const int n = N + 1;
  for (int c = 0; c < cols; c++)
    for (int d = 0; d < rows; d++) {
      uint8_t g[ORC_MAX_SYM_BINS];
      int len = binarize(sym[(size_t)c * rows + d], nq, method, k, g);
      int np = first_zero_1based(g, len);
      if (!np) np = len;
      if (n <= np)
        for (int i = n; i <= np; i++) { t_rp++; z_rp += g[i - 1] == 0; }
      if (n > np && n <= len)
        for (int i = n; i <= len; i++) { t_rs++; z_rs += g[i - 1] == 0; }
    }
  out[7 * N] = ratio_or_zero(z_rp, t_rp);
  out[7 * N + 1] = ratio_or_zero(z_rs, t_rs);
}

/* uint8(x * 255) as MATLAB converts: round to nearest, ties away from zero, saturate (cabacEncode.m:30) */
uint8_t orc_matlab_uint8_of_p(double p) {
  double x = p * 255.0;
  if (!(x == x)) return 0;
  double r = floor(x + 0.5);
  if (r < 0) r = 0;
  if (r > 255) r = 255;
  return (uint8_t)r;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic generator, SURVEY.md section 8(d)
 * ---------------------------------------------------------------------------------------- */
static uint32_t fmix32(uint32_t h) { /* murmur3 finaliser */
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

void orc_synth_fill(uint32_t seed, uint64_t s0, uint64_t nS, uint32_t B, uint32_t C,
                    int ep_half, uint8_t *bins, uint8_t *ctx) {
  int pow2 = (C & (C - 1)) == 0;
  for (uint64_t r = 0; r < nS; r++) {
    uint32_t s = (uint32_t)(s0 + r);
    for (uint32_t i = 0; i < B; i++) {
      uint32_t h = fmix32(seed ^ (s * 0x9E3779B1u) ^ (i * 0x85EBCA77u));
      uint32_t h2 = fmix32(h + 0x6A09E667u);
      uint32_t c = pow2 ? (h & (C - 1)) : (((h & 0xFFFFu) * C) >> 16);
      uint32_t thr = ((2 * c + 1) << 24) / (2 * C);
      uint8_t bin = (uint8_t)((h2 >> 8) < thr);
      uint8_t cb = (uint8_t)c;
      if (ep_half && (h >> 31)) {
        cb = ORC_BYPASS_CTX;
        bin = (uint8_t)(h2 & 1);
      }
      bins[r * B + i] = bin;
      ctx[r * B + i] = cb;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * State traces of the reference's Windows build (RWTH_TRACE_CABAC_STATES): what getEncoderStats /
 * getDecoderStats return for one context, SimpleCABACMex.cpp:356-466.
 *   trace state   ContextModel::addCabacStep, ContextModel.cpp:126-134: mps == 0 ? 63 - state : state + 64
 *   step          [codedBin, trace state before, mps before, trace state after, mps after] for every bin coded with
 *                 the context (SimpleCABACMex.cpp:231-242); on the DECODER side the gateway stores the bin variable
 *                 before decodeBin has written it (:316-327), so codedBin is always 0 there
 *   transitions   counts [trace state before][trace state after], 128 x 128
 * The context state evolves with the bins alone (updateMPS / updateLPS), so no arithmetic coder is needed here.
 * Returns the number of steps; trace5 receives the first cap_steps of them.
 * ---------------------------------------------------------------------------------------- */
uint8_t orc_trace_state(uint8_t p) { return (p & 1) == 0 ? (uint8_t)(63 - (p >> 1)) : (uint8_t)((p >> 1) + 64); }

uint64_t orc_trace_stats(const uint8_t *bins, const uint8_t *ctx, uint64_t nbins, const uint8_t *ctx_init, int nctx,
                         int which_ctx, int decoder_side, uint8_t *trace5, uint64_t cap_steps, uint32_t *trans) {
  uint8_t cm[256];
  memset(cm, orc_ctx_pack(1, 0), sizeof cm);
  if (nctx > 0) memcpy(cm, ctx_init, (size_t)nctx);
  if (trans) memset(trans, 0, 128 * 128 * sizeof(uint32_t));
  uint64_t steps = 0;
  for (uint64_t i = 0; i < nbins; i++) {
    const uint32_t c = ctx[i];
    if (c == ORC_BYPASS_CTX || (nctx <= ORC_PROB_MAX_NCTX && c > ORC_PROB_BASE)) continue; /* no context model involved */
    const uint8_t p = cm[c];
    const uint8_t a = bins[i] == (p & 1u) ? orc_ctx_next_mps(p) : orc_ctx_next_lps(p);
    cm[c] = a;
    if ((int)c != which_ctx) continue;
    const uint8_t tp = orc_trace_state(p), ta = orc_trace_state(a);
    if (trace5 && steps < cap_steps) {
      uint8_t *t = trace5 + 5 * steps;
      t[0] = decoder_side ? 0 : bins[i];
      t[1] = tp;
      t[2] = p & 1u;
      t[3] = ta;
      t[4] = a & 1u;
    }
    if (trans) trans[(size_t)tp * 128 + ta]++;
    steps++;
  }
  return steps;
}

/* ------------------------------------------------------------------------------------------
 * All-core CPU ceiling for bench.py's cpu_baseline: the restatement above with its in-memory sink on
 * native threads, streams split evenly, no files and nothing interpreted between streams.  Every thread generates its
 * streams with the counter-based generator, encodes, decodes and compares.  *wall_s = time from the first thread's
 * start to the last one's end.  Returns 1 when every round trip reproduced its bins, else 0.
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
  uint32_t seed, B, C;
  int ep, nthreads, t;
  uint64_t s0, done;
  double seconds;
  int ok;
} orc_rt_job;

static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *orc_rt_worker(void *arg) {
  orc_rt_job *j = (orc_rt_job *)arg;
  const size_t cap = (6 * (size_t)j->B + 9 + 7) / 8 + 16;
  uint8_t *bins = (uint8_t *)malloc(j->B), *ctx = (uint8_t *)malloc(j->B), *dec = (uint8_t *)malloc(j->B);
  uint8_t *out = (uint8_t *)malloc(cap);
  uint8_t init[256];
  memset(init, 0, sizeof init); /* (mps 0, state 0): the workload's initialisation */
  j->ok = bins && ctx && dec && out;
  const double deadline = orc_now() + j->seconds;
  /* thread t takes streams s0 + t, s0 + t + nthreads, ... until the time is up */
  for (uint64_t k = 0; j->ok && orc_now() < deadline; k++) {
    orc_synth_fill(j->seed, j->s0 + (uint64_t)j->t + k * (uint64_t)j->nthreads, 1, j->B, j->C, j->ep, bins, ctx);
    const size_t len = orc_encode_stream(bins, ctx, j->B, init, (int)j->C, out, cap, NULL, NULL);
    const uint32_t st = orc_decode_stream(out, len, ctx, j->B, init, (int)j->C, dec, NULL);
    if (len == 0 || st != 0 || memcmp(dec, bins, j->B) != 0) j->ok = 0;
    j->done++;
  }
  free(bins);
  free(ctx);
  free(dec);
  free(out);
  return NULL;
}

/* runs for about `seconds`; *streams_done = round trips completed by all threads, *wall_s = first start to last end */
int orc_roundtrip_threads(uint32_t seed, uint64_t s0, uint32_t B, uint32_t C, int ep_half, int nthreads, double seconds,
                          uint64_t *streams_done, double *wall_s) {
  if (nthreads < 1 || nthreads > 1024 || C > 255) return 0;
  static pthread_t th[1024];
  static orc_rt_job job[1024];
  const double t0 = orc_now();
  for (int t = 0; t < nthreads; t++) {
    job[t] = (orc_rt_job){seed, B, C, ep_half, nthreads, t, s0, 0, seconds, 0};
    if (pthread_create(&th[t], NULL, orc_rt_worker, &job[t]) != 0) return 0;
  }
  int ok = 1;
  uint64_t done = 0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    ok &= job[t].ok;
    done += job[t].done;
  }
  if (wall_s) *wall_s = orc_now() - t0;
  if (streams_done) *streams_done = done;
  return ok;
}
