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
  uint32_t method;    // 0 = TU (max value nq-1), 1 = EGk; encode only: 2 = TRk, 3 = FL32 (cabacBinarizer.m:39-54,71-75;
                      // the reference cannot decode those either: cabacDecodeSymbolFinished.m:12-30)
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
  uint32_t np;  // 1-based position of the first 0 bin (the end of a TU string / of an EG prefix), 0 = none so far
};

CABAC_HD uint32_t low_mask(uint32_t n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }
CABAC_HD uint32_t first_zero_1based(uint32_t bits, uint32_t len) {  // 0 = none
  const uint32_t z = ~bits & low_mask(len);
  return z ? (uint32_t)__builtin_ctz(z) + 1u : 0u;
}
CABAC_HD uint32_t bitrev(uint32_t v, uint32_t n) {  // reverse the low n bits (1 <= n <= 32)
#if defined(__clang__)
  return __builtin_bitreverse32(v) >> (32u - n);
#else
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
  return r;
#endif
}

CABAC_HD BinString binarize(uint32_t v, const SymParams &P) {
  BinString g;
  if (P.method == 0) {  // TUCode: v ones, then a zero unless v == max
    const uint32_t mx = P.nq - 1;
    g.bits = low_mask(v);
    g.len = v == mx ? mx : v + 1;
    g.np = v == mx ? 0u : v + 1;
  } else if (P.method == 2) {  // TRCode: floor(v / 2^k) ones, a zero, k suffix bits (all ones from the maximum on: the
                               // reference's escape branch is a TODO that leaves the suffix as initialised)
    const uint32_t n_p = (v >> P.k) + 1;
    const uint32_t v_s = v >= P.nq - 1 ? low_mask(P.k) : v - ((n_p - 1) << P.k);
    g.bits = low_mask(n_p - 1) | (P.k ? bitrev(v_s, P.k) << n_p : 0u);
    g.len = n_p + P.k;
    g.np = n_p;
  } else if (P.method == 3) {  // FLCode(v, 32): 32 bins, MSB first
    g.bits = bitrev(v, 32);
    g.len = 32;
    g.np = first_zero_1based(g.bits, 32);
  } else {  // EGCode
    const uint32_t t = (v >> P.k) + 1;
    const uint32_t n_p = 32u - clz32(t);  // floor(log2(t)) + 1
    const uint32_t n_s = P.k + n_p - 1;
    const uint32_t v_s = v - (low_mask(n_p - 1) << P.k);
    g.bits = low_mask(n_p - 1) | (n_s ? bitrev(v_s, n_s) << n_p : 0u);  // suffix MSB first
    g.len = n_p + n_s;
    g.np = n_p;
  }
  return g;
}

// bin n (1-based) of a symbol being decoded
CABAC_HD BinString append_bin(BinString g, uint32_t n, uint32_t bin) {
  g.bits |= bin << (n - 1);
  g.len = n;
  g.np = (g.np == 0 && bin == 0) ? n : g.np;
  return g;
}

CABAC_HD uint32_t debinarize(BinString g, const SymParams &P) {
  if (P.method == 0) return g.np ? g.np - 1 : P.nq - 1;
  const uint32_t n_p = g.np;  // an EG string always has its prefix zero
  const uint32_t n_s = g.len - n_p;
  const uint32_t suffix = n_s ? bitrev(g.bits >> n_p, n_s) : 0u;
  return (low_mask(n_p - 1) << P.k) + suffix;
}

// has the symbol ended with bin n (1-based) = the bin just decoded?  g holds bins 1..n.
CABAC_HD bool symbol_finished(BinString g, const SymParams &P) {
  const uint32_t n = g.len;
  if (P.method == 0) return g.np == n || n == P.nq - 1;
  return g.np != 0 && n == g.np + (P.k + g.np - 1);  // prefix + suffix complete (n_s == 0: done at the zero)
}

// 0-based context index for bin n (1-based) of the current symbol; cur holds at least bins 1..n-1 (cur.np may describe
// the whole symbol: only a zero before bin n counts), up is the bin string of the symbol one row up (len 0 = none)
CABAC_HD uint32_t iss_ctx(uint32_t n, BinString cur, BinString up, const SymParams &P) {
  const uint32_t N = P.nlbp;
  const bool in_prefix = cur.np == 0 || cur.np >= n;           // no zero among bins 1..n-1
  const bool prefix_up = up.np == 0 || n <= up.np;              // the upper symbol is still in its prefix at n
  const bool has_up = up.len >= n;
  const uint32_t upbit = (up.bits >> (n - 1)) & 1u;
  const uint32_t j = in_prefix ? n : n - cur.np;                // position inside the prefix / the suffix
  const bool cond = has_up && (prefix_up == in_prefix);         // the upper bin n belongs to the same part
  // the conditioning types: prefix cond0 / cond1 = bits 0 / 1, suffix conds0 / conds1 = bits 2 / 3
  const uint32_t tbit = (in_prefix ? 0u : 2u) + upbit;
  const bool use_up = cond && ((P.types >> tbit) & 1u);
  const uint32_t base = in_prefix ? 0u : 4u * N;
  uint32_t id = base + j;
  id = use_up ? base + (1u + upbit) * N + j : id;
  // prefix only: condbinlft (bit 4) when the upper symbol does not condition: previous bin of this symbol is 1 (always
  // true inside a prefix for n > 1), ref: cabacContextSelection.m:44-45
  const bool lft = in_prefix && !(has_up && prefix_up) && n > 1 && ((cur.bits >> (n - 2)) & 1u) && (P.types & 16u);
  id = lft ? 3u * N + n - 1u : id;
  id = j <= N ? id : 7u * N + (in_prefix ? 1u : 2u);
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
  uint32_t row;        // idx % col_rows, kept by counting
  CABAC_HD void start() {
    idx = 0;
    row = 0;
    n = 1;
    cur.bits = cur.len = cur.np = 0;
    up.bits = up.len = up.np = 0;
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
    cur.bits = cur.len = cur.np = 0;
    row++;
    if (row == P.col_rows) {  // first row of the next column (never true for col_rows == 0)
      row = 0;
      up.bits = up.len = up.np = 0;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The binarisation as a code tree with the selectors folded in (small alphabets: nq <= 16).
// A symbol is a path from the root; the context of the bin coded AT a node depends on the node and on the symbol one row
// up only -- iss_ctx / demo_ctx look at nothing else -- so both become one byte look-up per bin:
//   ctx[node * kTrieUp + up]   up = value of the symbol one row up, kTrieNoUp = there is none
//   child[node * 2 + bin]      next node (< kTrieNodes), 0x80 | symbol at the end of a code word, 0xFF = no code word goes on
// Built on the host from the functions above (sym_trie_build), so it cannot drift from them; a walk that leaves the tree
// (0xFF: corrupt stream) hands the stream to the general kernel, which parses any bin string.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTrieNodes = 64, kTrieUp = 17, kTrieNoUp = 16, kTrieNone = 0xFFu;
struct SymTrie {
  uint8_t child[kTrieNodes * 2];
  uint8_t ctx[kTrieNodes * kTrieUp];
  uint32_t nodes;
};
// false: the alphabet does not fit (nq > 16, more than kTrieNodes inner nodes) -- the caller keeps the general kernels
inline bool sym_trie_build(const SymParams &P, SymTrie *T) {
  if (P.nq < 2 || P.nq > 16) return false;
  for (uint32_t i = 0; i < kTrieNodes * 2; i++) T->child[i] = (uint8_t)kTrieNone;
  for (uint32_t i = 0; i < kTrieNodes * kTrieUp; i++) T->ctx[i] = 0;
  bool seen[kTrieNodes] = {};
  T->nodes = 1;
  for (uint32_t v = 0; v < P.nq; v++) {
    const BinString g = binarize(v, P);
    if (g.len == 0 || g.len > 29) return false;  // (the encoder keeps a bin string << 3 in 32 bits)
    uint32_t node = 0;
    for (uint32_t n = 1; n <= g.len; n++) {
      const uint32_t bit = (g.bits >> (n - 1)) & 1u;
      for (uint32_t up = 0; up < kTrieUp; up++) {
        BinString ub;
        ub.bits = ub.len = ub.np = 0;
        const bool has = up < P.nq;
        if (has) ub = binarize(up, P);
        if (!has && up != kTrieNoUp) continue;
        const uint32_t c = P.selector == 0 ? demo_ctx(n, has, ub.bits & 1u) : iss_ctx(n, g, ub, P);
        if (c > 255u) return false;
        if (seen[node] && T->ctx[node * kTrieUp + up] != (uint8_t)c) return false;  // two symbols disagree: not a tree code
        T->ctx[node * kTrieUp + up] = (uint8_t)c;
      }
      seen[node] = true;
      uint8_t &slot = T->child[node * 2 + bit];
      if (n == g.len) {
        if (slot != kTrieNone) return false;  // a code word that is a prefix of another
        slot = (uint8_t)(0x80u | v);
      } else {
        if (slot == kTrieNone) {
          if (T->nodes == kTrieNodes) return false;
          slot = (uint8_t)T->nodes++;
        } else if (slot & 0x80u) {
          return false;
        }
        node = slot;
      }
    }
  }
  return true;
}

}  // namespace isscabac
#endif
