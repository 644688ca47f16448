// cabac_symbols.h -- symbol-level front end of the engine: TU / EGk (de)binarisation and the two context
// selectors of the reference, per lane, fused in front of (behind) the arithmetic coder so that a caller hands over
// integer symbols instead of pre-binarised bins.  On decode this has to live in the kernel anyway: the next context
// depends on bins that were just decoded (ref: CABAC/cabacDemo.m:158-166, ISS/+coder/cabacDecode.m:38).
//
// Restated from (ref: = /root/reference):
//   binarize / debinarize   CABAC/cabacBinarizer.m:30-37 (TU), :56-69 (EGk), :71-75 (FL suffix);
//                           CABAC/cabacDebinarizer.m:28-31, :46-53
//   symbol_finished         CABAC/cabacDecodeSymbolFinished.m:12-30
//   iss_ctx                 ISS/+coder/cabacContextSelection.m:24-67
//   demo_ctx                CABAC/cabacDemo.m:113-121
// A bin string is kept as packed bits (bit i = bin i+1) plus a length, at most 32 bins per symbol.
#ifndef ISSCABAC_SYMBOLS_H
#define ISSCABAC_SYMBOLS_H

#include "cabac_lane.h"

namespace isscabac {

struct SymParams {
  uint32_t method;    // 0 = TU (max value nq-1), 1 = EGk
  uint32_t k;         // EG order
  uint32_t nq;        // number of quantisation levels (symbols are 0..nq-1)
  uint32_t selector;  // 0 = demo (3 contexts), 1 = ISS (7*nlbp+2 contexts)
  uint32_t types;     // ISS: 1 cond0, 2 cond1, 4 conds0, 8 conds1, 16 condbinlft
  uint32_t nlbp;
  uint32_t col_rows;  // > 0: the stream is a whole matrix walked column by column (rows = col_rows); the symbol "one
                      // row up" does not exist for the first symbol of every column (ref: cabacEncode.m:52), while
                      // the coder and the adaptive context states run on across columns.  0: one column per stream.
};

struct BinString {
  uint32_t bits, len;
};

CABAC_HD uint32_t low_mask(uint32_t n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }
CABAC_HD uint32_t first_zero_1based(uint32_t bits, uint32_t len) {  // 0 = none
  const uint32_t z = ~bits & low_mask(len);
  return z ? (uint32_t)__builtin_ctz(z) + 1u : 0u;
}
CABAC_HD uint32_t bitrev(uint32_t v, uint32_t n) {  // reverse the low n bits (n <= 32)
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
  return r;
}

CABAC_HD BinString binarize(uint32_t v, const SymParams &P) {
  BinString g;
  if (P.method == 0) {  // TUCode: v ones, then a zero unless v == max
    const uint32_t mx = P.nq - 1;
    g.bits = low_mask(v);
    g.len = v == mx ? mx : v + 1;
  } else {  // EGCode
    const uint32_t t = (v >> P.k) + 1;
    const uint32_t n_p = 32u - clz32(t);  // floor(log2(t)) + 1
    const uint32_t n_s = P.k + n_p - 1;
    const uint32_t v_s = v - (low_mask(n_p - 1) << P.k);
    g.bits = low_mask(n_p - 1) | (n_s ? bitrev(v_s, n_s) << n_p : 0u);  // suffix MSB first
    g.len = n_p + n_s;
  }
  return g;
}

CABAC_HD uint32_t debinarize(BinString g, const SymParams &P) {
  const uint32_t fz = first_zero_1based(g.bits, g.len);
  if (P.method == 0) return fz ? fz - 1 : P.nq - 1;
  const uint32_t n_p = fz;  // an EG string always has its prefix zero
  const uint32_t n_s = g.len - n_p;
  const uint32_t suffix = n_s ? bitrev(g.bits >> n_p, n_s) : 0u;
  return (low_mask(n_p - 1) << P.k) + suffix;
}

// has the symbol ended with bin n (1-based) = the bin just decoded?  g holds bins 1..n.
CABAC_HD bool symbol_finished(BinString g, const SymParams &P) {
  const uint32_t n = g.len;
  if (P.method == 0) return ((g.bits >> (n - 1)) & 1u) == 0 || n == P.nq - 1;
  const uint32_t n_p = first_zero_1based(g.bits, n);
  if (n_p == 0) return false;            // still in the prefix
  return n == n_p + (P.k + n_p - 1);     // prefix + suffix complete (n_s == 0: done at the zero)
}

// 0-based context index for bin n (1-based) of the current symbol; cur holds at least bins 1..n-1, up is the bin
// string of the symbol one row up (len 0 = none)
CABAC_HD uint32_t iss_ctx(uint32_t n, BinString cur, BinString up, const SymParams &P) {
  const uint32_t N = P.nlbp;
  const uint32_t np_prev = first_zero_1based(cur.bits, n - 1);
  const uint32_t np_up = first_zero_1based(up.bits, up.len);
  const bool prefix_up = !(np_up && n > np_up);
  const bool has_up = up.len >= n;
  const uint32_t upbit = (up.bits >> (n - 1)) & 1u;
  uint32_t id;
  if (np_prev == 0) {  // prefix
    if (n <= N) {
      id = n;
      if (has_up && prefix_up) {
        if (upbit == 0 && (P.types & 1u)) id = N + n;
        else if (upbit == 1 && (P.types & 2u)) id = 2 * N + n;
      } else if (n > 1 && ((cur.bits >> (n - 2)) & 1u) && (P.types & 16u)) {
        id = 3 * N + n - 1;
      }
    } else {
      id = 7 * N + 1;
    }
  } else {  // suffix
    const uint32_t j = n - np_prev;
    if (j <= N) {
      id = 4 * N + j;
      if (has_up && !prefix_up) {
        if (upbit == 0 && (P.types & 4u)) id = 5 * N + j;
        else if (upbit == 1 && (P.types & 8u)) id = 6 * N + j;
      }
    } else {
      id = 7 * N + 2;
    }
  }
  return id - 1;  // cabacEncode.m:64 passes id-1 to the engine
}

CABAC_HD uint32_t demo_ctx(uint32_t n, bool have_prev, uint32_t prev_first) {
  return (n == 1 && have_prev) ? (prev_first == 1 ? 1u : 2u) : 0u;
}

// per-lane walk over one stream's symbols, one bin per step
struct SymWalk {
  uint32_t idx;        // symbol index
  uint32_t n;          // 1-based position of the next bin inside the current symbol
  BinString cur, up;
  uint32_t prev_first;
  CABAC_HD void start() {
    idx = 0;
    n = 1;
    cur.bits = cur.len = 0;
    up.bits = up.len = 0;
    prev_first = 0;
  }
  CABAC_HD uint32_t ctx(const SymParams &P) const {
    return P.selector == 0 ? demo_ctx(n, idx > 0, prev_first) : iss_ctx(n, cur, up, P);
  }
  CABAC_HD void next_symbol(const SymParams &P) {
    up = cur;
    prev_first = cur.bits & 1u;
    idx++;
    n = 1;
    cur.bits = cur.len = 0;
    if (P.col_rows && idx % P.col_rows == 0) up.bits = up.len = 0;  // first row of the next column
  }
};

}  // namespace isscabac
#endif
