// lane_harness.cpp -- host build of the engine's per-lane CABAC code (isscabac_amd/csrc/cabac_lane.h)
// for the CPU test-suite.  It drives the SAME inline functions the HIP kernels call, with the same
// cadence of flush / top-up / boundary calls, over plain arrays, so arithmetic bugs are caught (and
// ASan/UBSan can run) without a GPU.  It is a test build of product code, never a runtime fallback.
//
//   g++ -O1 -g -shared -fPIC [-fsanitize=address,undefined] -I isscabac_amd/csrc tests/lane_harness.cpp
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <vector>

#include "cabac_lane.h"
#include "cabac_symbols.h"

using namespace isscabac;

namespace {

const uint32_t kLpsRow[64] = {ISSCABAC_LPS_ROWS};
const uint8_t kTransLps[64] = {ISSCABAC_TRANS_LPS};
const uint32_t kProbRow[49] = {ISSCABAC_PROB_ROWS};
struct NextTab {
  uint32_t e[64];
  NextTab() {
    for (uint32_t s = 0; s < 64; s++) e[s] = next_state_entry(s, kTransLps[s]);
  }
};
const NextTab kNext;

struct HostStates {
  uint8_t cm[257];  // doubled states q, like the kernels keep them in LDS; [256]: the decoder's dummy context of the mixed step
  void init(const uint8_t *packed, int nctx) {
    std::memset(cm, q_from_packed(1), 256);  // ContextModel() default: (mps 1, state 0)
    for (int c = 0; c < nctx; c++) cm[c] = (uint8_t)q_from_packed(packed[c]);
    cm[256] = (uint8_t)kDecBypassRaw;
  }
  void final_packed(uint8_t *out, int nctx) const {
    for (int c = 0; c < nctx; c++) out[c] = (uint8_t)packed_from_q(cm[c]);
  }
  uint32_t load(uint32_t c) const { return cm[c]; }
  void store(uint32_t c, uint32_t q) { cm[c] = (uint8_t)q; }
  uint32_t addr(uint32_t c) const { return c; }
  uint32_t load_at(uint32_t a) const { return cm[a]; }
  void store_at(uint32_t a, uint32_t q) { cm[a] = (uint8_t)q; }
  uint32_t lps_row(uint32_t q) const { return kLpsRow[q >> 2]; }
  uint32_t next_entry(uint32_t q) const { return kNext.e[q >> 2]; }
  // the byte tables of the kernels' fast paths (cabac_kernels.hip: LdsStates::lps_mem / next_mem / next_lps_mem)
  uint32_t lps_mem(uint32_t q, uint32_t range) const { return (kLpsRow[q >> 2] >> (8 * ((range >> 6) & 3u))) & 0xFFu; }
  uint32_t next_mem(uint32_t q_or_bin) const { return next_state_by_bin(q_or_bin >> 1, q_or_bin & 1u, kTransLps[q_or_bin >> 2]); }
  uint32_t path_index(uint32_t q) const { return q; }
  // ... and of the single-lookup steps (LdsStates::entry_mem): row | successors << 32, by bin (encoder) or by path (decoder)
  bool by_path = false;
  uint64_t entry_mem(uint32_t q) const {
    if (!by_path && q >= 0xFEu) return (uint64_t)kEncBypassHi << 32;  // k_encode's table: packed state 127 is the bypass entry (enc_bin_mixed3_acc)
    if (!by_path && q >= kEncBypassQ) return 0;                       // ... and 126 the all-zero one (enc_bin_mixed_acc)
    const uint32_t b0 = by_path ? next_path_mem(q) : next_mem(q), b1 = by_path ? next_path_mem(q + 1u) : next_mem(q | 1u);
    return (uint64_t)kLpsRow[q >> 2] | (uint64_t)(by_path ? (b0 | b1 << 8) : enc_entry_hi(q >> 1, b0, b1)) << 32;
  }
  // the decoder's entry by RAW state byte (LdsStates::entry_raw): bit 0 of the byte is a stale bin
  uint64_t entry_raw(uint32_t qb) const {
    if (qb >= kDecBypassRaw) return (uint64_t)kDecBypassHi << 32;  // k_decode's table: raw bytes 254 / 255 hold the bypass entry
    const uint32_t q = qb & ~1u;
    return (uint64_t)kLpsRow[q >> 2] | (uint64_t)dec_entry_hi_out(q >> 1, next_path_mem(q), next_path_mem(q + 1u)) << 32;
  }
  void store_at_hi(uint32_t a, uint32_t x) { cm[a] = (uint8_t)(x >> 16); }
  uint32_t next_path_mem(uint32_t q_plus_path) const {  // path 1 = MPS
    return (next_state_entry16(q_plus_path >> 1, kTransLps[q_plus_path >> 2]) >> (8 * (1u - (q_plus_path & 1u)))) & 0xFFu;
  }
};
struct HostBin {  // Bin policy of enc_bin_at_mem
  uint32_t bin;
  uint32_t xor_q(uint32_t q) const { return (bin << 1) ^ q; }
  uint32_t or_q(uint32_t q) const { return q | bin; }
  uint32_t next_of(uint32_t hi) const { return hi >> (8 * bin); }  // the store takes the low byte
  uint32_t lps_mask(uint32_t, uint32_t t) const { return (uint32_t)((int32_t)(t << 15) >> 31); }  // bit 16 of the shifted entry
  // the kernel policy's per-bin byte of a per-word word (SDWA there); the harness hands in the bin's byte alone
  uint32_t add_byte(uint32_t x, uint32_t y) const { return x + (y & 0xFFu); }
  uint32_t mul_byte(uint32_t x, uint32_t y) const { return x * (y & 0xFFu); }
  uint32_t lps_mask_sext(uint32_t t) const { return (uint32_t)((int32_t)(t << 8) >> 24); }  // byte 2 of the shifted entry, sign-extended
};

struct HostSink {
  uint8_t *buf;
  size_t len, cap;
  void put(uint32_t b) {
    if (len < cap) buf[len] = (uint8_t)b;
    len++;
  }
  void put16(uint32_t x) {
    put(x >> 8);
    put(x & 0xFFu);
  }
};

// payload bytes of one stream inside a bigger buffer, with the device's bounds rule
struct HostMem {
  const uint8_t *base;    // first byte of this stream
  uint64_t avail;         // bytes from base to the end of the whole payload buffer
  uint32_t *fast, *slow;  // counters
  void load16(uint32_t off, uint64_t &lo, uint64_t &hi) const {
    uint8_t tmp[16];
    if ((uint64_t)off + 16 <= avail) {
      std::memcpy(tmp, base + off, 16);
      if (fast) (*fast)++;
    } else {
      for (uint32_t i = 0; i < 16; i++) tmp[i] = (uint64_t)off + i < avail ? base[off + i] : 0;
      if (slow) (*slow)++;
    }
    std::memcpy(&lo, tmp, 8);
    std::memcpy(&hi, tmp + 8, 8);
  }
};

}  // namespace

static uint32_t g_word_shift_max = 0;  // largest total shift of one input word seen by mode 18

extern "C" {

uint32_t lane_word_shift_max(void) { return g_word_shift_max; }

// Largest total renormalisation shift of k consecutive context bins, over every start range, state (0..62) and path:
// the bound behind EncWordAcc (k = 4 must give 22) and behind the 4-bin flush check of the batch encoder.
uint32_t lane_max_shift_sum(int k) {
  static uint32_t best[2][511];
  for (uint32_t r = 256; r <= 510; r++) best[0][r] = 0;
  int cur = 0;
  for (int step = 0; step < k; step++, cur ^= 1) {
    for (uint32_t r = 256; r <= 510; r++) {
      uint32_t b = 0;
      for (uint32_t state = 0; state <= 62; state++) {
        const uint32_t lps = (kLpsRow[state] >> (8 * ((r >> 6) & 3u))) & 0xFFu, rmps = r - lps;
        const uint32_t nl = clz32(lps) - 23u, nm = clz32(rmps) - 23u;
        const uint32_t vl = nl + best[cur][lps << nl], vm = nm + best[cur][rmps << nm];
        b = vl > b ? vl : b;
        b = vm > b ? vm : b;
      }
      best[cur ^ 1][r] = b;
    }
  }
  uint32_t m = 0;
  for (uint32_t r = 256; r <= 510; r++) m = best[cur][r] > m ? best[cur][r] : m;
  return m;
}

// ... the same with bypass bins allowed anywhere in the run (one bit each, the range stays): the bound behind the decode
// kernel's unguarded mixed words
uint32_t lane_max_shift_sum_ep(int k) {
  static uint32_t best[2][511];
  for (uint32_t r = 256; r <= 510; r++) best[0][r] = 0;
  int cur = 0;
  for (int step = 0; step < k; step++, cur ^= 1) {
    for (uint32_t r = 256; r <= 510; r++) {
      uint32_t b = 1 + best[cur][r];  // a bypass bin
      for (uint32_t state = 0; state <= 62; state++) {
        const uint32_t lps = (kLpsRow[state] >> (8 * ((r >> 6) & 3u))) & 0xFFu, rmps = r - lps;
        const uint32_t nl = clz32(lps) - 23u, nm = clz32(rmps) - 23u;
        const uint32_t vl = nl + best[cur][lps << nl], vm = nm + best[cur][rmps << nm];
        b = vl > b ? vl : b;
        b = vm > b ? vm : b;
      }
      best[cur ^ 1][r] = b;
    }
  }
  uint32_t m = 0;
  for (uint32_t r = 256; r <= 510; r++) m = best[cur][r] > m ? best[cur][r] : m;
  return m;
}

// flush_every: bins between enc_flush calls (1 = reference cadence, 4 = kernel cadence)
// flush_every < 0: same cadence, but every bin goes through the combined enc_bin_any step
long lane_encode_stream(const uint8_t *bins, const uint8_t *ctx, uint32_t n, const uint8_t *ctx_init, int nctx,
                        int flush_every, uint8_t *out, size_t cap, uint8_t *ctx_final, int *max_w) {
  const bool any = flush_every < 0;
  if (any) flush_every = -flush_every;
  // 20: the select-based step for words that mix context-coded and bypass bins (enc_bin_mixed_acc), every word it can take
  // 21: ... the step on a dummy context that holds the bypass state (enc_bin_mixed3_acc: <= 64 contexts, the kernel's 5-bit group mask
  //     puts a bypass byte's state address on the dummy -- cm[256] here)
  const bool mixed3 = flush_every == 21;
  const bool mixedw = flush_every == 20 || mixed3;
  if (mixedw) flush_every = 18;
  // 19: the latency-oriented instance of that path: state read one bin ahead, successor forwarded for a repeated context
  const bool lat = flush_every == 19;
  if (lat) flush_every = 18;
  // 18: the kernel's plain path -- an aligned word of four context bins adds to `low` once (EncWordAcc / enc_word_close),
  //     each bin on the single-lookup step (enc_bin_row_acc)
  const bool wordacc = flush_every == 18;
  if (wordacc) flush_every = 17;
  // 22: ... with k_encode's lane-by-lane step on the single-lookup entries (enc_bin_at_row)
  const bool atrow = flush_every == 22;
  if (atrow) flush_every = 16;
  const bool memtab = flush_every == 17;  // ... with the byte-table bin step (enc_bin_at_mem: the symbol kernels')
  if (memtab) flush_every = 16;
  const bool digits16 = flush_every == 16;  // the batch kernel's output path: 16-bit digits, checked every 4 bins
  if (digits16) flush_every = 4;
  HostStates st;
  st.init(ctx_init, nctx);
  HostSink sink{out, 0, cap};
  EncCore e;
  enc_start(e);
  int mw = 0;
  for (uint32_t i = 0; i < n; i++) {
    auto ctx_or_ep = [&](uint32_t j) { return ctx[j] < 128 || ctx[j] == kBypass; };
    auto ctx64_or_ep = [&](uint32_t j) { return ctx[j] < 64 || ctx[j] == kBypass; };
    if (mixedw && nctx <= (mixed3 ? 64 : 128) && i % 4 == 0 && i + 4 <= n &&
        (mixed3 ? ctx64_or_ep(i) && ctx64_or_ep(i + 1) && ctx64_or_ep(i + 2) && ctx64_or_ep(i + 3)
                : ctx_or_ep(i) && ctx_or_ep(i + 1) && ctx_or_ep(i + 2) && ctx_or_ep(i + 3))) {
      EncWordAcc acc;
      for (int k = 0; k < 4; k++) {  // the kernel's view: bit 7 = bypass, the state address masked into the image
        const uint32_t c = ctx[i + k], mb = (c & 0x80u) ? ~0u : 0u, bm = (bins[i + k] && mb) ? ~0u : 0u;
        if (mixed3) enc_bin_mixed3_acc(e, acc, k == 0, HostBin{bins[i + k]}, mb & 1u, bm & 1u, (c & 0x80u) ? 256u : st.addr(c), st);
        else enc_bin_mixed_acc(e, acc, k == 0, HostBin{bins[i + k]}, mb, bm, st.addr(c & (nctx <= 64 ? 0x3Fu : 0x7Fu)), st);
      }
      if (mixed3 && st.cm[256] != 0xFEu) return -3;  // the dummy keeps the bypass state
      if (acc.S > g_word_shift_max) g_word_shift_max = acc.S;
      enc_word_close(e, acc);
      if (e.W > mw) mw = e.W;
      i += 3;
      if (e.W >= 39) enc_flush16_wave(e, sink);
      if (e.W > 62) return -2;
      continue;
    }
    if (wordacc && i % 4 == 0 && i + 4 <= n && ctx[i] < 128 && ctx[i + 1] < 128 && ctx[i + 2] < 128 && ctx[i + 3] < 128) {
      EncWordAcc acc;
      if (lat) {  // the kernel's order: read bin k + 1's state, then store bin k's successor, forward on equal addresses
        uint32_t at = st.addr(ctx[i]), at_prev = 0, nx_prev = 0, qld = st.load_at(at);
        for (int k = 0; k < 4; k++) {
          const uint32_t q = (k > 0 && at == at_prev) ? nx_prev : qld;
          uint32_t at_next = 0;
          if (k < 3) {
            at_next = st.addr(ctx[i + k + 1]);
            qld = st.load_at(at_next);
          }
          nx_prev = enc_bin_lat_acc(e, acc, k == 0, HostBin{bins[i + k]}, at, q, st);
          at_prev = at;
          at = at_next;
        }
      } else {
        for (int k = 0; k < 4; k++) enc_bin_row_acc(e, acc, k == 0, HostBin{bins[i + k]}, st.addr(ctx[i + k]), st);
      }
      if (acc.S > g_word_shift_max) g_word_shift_max = acc.S;
      enc_word_close(e, acc);
      if (e.W > mw) mw = e.W;
      i += 3;
      if (e.W >= 39) enc_flush16_wave(e, sink);
      if (e.W > 62) return -2;
      continue;
    }
    const bool prob = nctx <= 128 && is_prob_byte(ctx[i]);  // the batch kernels' rule
    if (prob && ctx[i] != kProbBase + 50u) {
      if (prob_valid(ctx[i] - kProbBase)) enc_bin_prob(e, bins[i], ctx[i] - kProbBase, kProbRow[prob_row_index(ctx[i] - kProbBase)]);
    } else if (prob) enc_bin_ep(e, bins[i]);
    else if (any) enc_bin_any(e, (uint32_t)bins[i] << 1, ctx[i], st);
    else if (ctx[i] == kBypass) enc_bin_ep(e, bins[i]);
    else if (memtab) enc_bin_at_mem(e, HostBin{bins[i]}, st.addr(ctx[i]), st);
    else if (atrow) enc_bin_at_row(e, HostBin{bins[i]}, st.addr(ctx[i]), st);
    else enc_bin(e, ((uint32_t)bins[i] << 1) | 0xA5A5A5A1u, ctx[i], st);  // only bit 1 may matter
    if (e.W > mw) mw = e.W;
    if ((i + 1) % (uint32_t)flush_every == 0) {
      if (digits16) {  // the batch kernel's cadence: every 4 bins, only when `low` could overflow within 4 more
        if (e.W >= 39) enc_flush16_wave(e, sink);
        if (e.W > 62) return -2;
      } else {
        enc_flush(e, sink);
      }
    }
  }
  if (digits16) {
    enc_flush16(e, sink);
    enc_finish16(e, sink);
  } else {
    enc_flush(e, sink);
    enc_finish(e, sink);
  }
  if (ctx_final) st.final_packed(ctx_final, nctx);
  if (max_w) *max_w = mw;
  return sink.len <= cap ? (long)sink.len : -1;
}

// The wave-pair encoder (cabac_pair.hip: k_encode_pair) on the host: wave A's record production restated over plain arrays
// (the device code is LDS plumbing and inline assembly around exactly these rules), wave B's steps THE SAME functions the
// kernel calls (cabac_lane.h: enc_rec_plain / enc_rec_mixed / enc_rec_general), with the kernel's chunk classification and
// flush cadence.  nmin: the shortest stream of the lane's wave (a chunk is plain / mixed only if it lies inside every lane's
// stream; pass n for a uniform batch, 0 to force the general records everywhere).  kinds[3] (nullable) counts the chunks by
// kind: plain, general, mixed.
long lane_encode_stream_pair(const uint8_t *bins, const uint8_t *ctx, uint32_t n, uint32_t nmin, const uint8_t *ctx_init, int nctx,
                             uint8_t *out, size_t cap, uint32_t *kinds) {
  if (nctx > 64) return -3;  // (the pair serves batches of up to 64 contexts)
  HostStates st;
  st.init(ctx_init, nctx);
  uint8_t dummy = 0;  // the bypass bins' context in mixed chunks: only ever receives the all-zero entry's successor
  HostSink sink{out, 0, cap};
  EncCore e;
  enc_start(e);
  for (uint32_t i0 = 0; i0 < n; i0 += 16) {
    uint8_t cb[16] = {0}, bb[16] = {0};
    for (uint32_t k = 0; k < 16 && i0 + k < n; k++) {
      cb[k] = ctx[i0 + k];
      bb[k] = bins[i0 + k] & 1u;
    }
    const bool full = i0 + 16 <= nmin;
    bool hi = false, odd = false;  // chunk_not_plain<1>: some byte >= 64; odd_high_bytes: bit 7 set without being 0xFF
    for (int k = 0; k < 16; k++) {
      hi |= (cb[k] & 0xC0u) != 0;
      odd |= (cb[k] & 0x80u) && cb[k] != 0xFFu;
    }
    uint32_t row[16], m[16], ext[16] = {0}, addm[16] = {0};
    int kind;
    if (full && (!hi || !odd)) {
      kind = hi ? 2 : 0;
      for (int k = 0; k < 16; k++) {
        const bool byp = hi && (cb[k] & 0x80u);
        // A's lookup: the state byte (a bypass bin: the dummy's, or-ed with 0xFE -> the all-zero entry of state 63)
        const uint32_t q = byp ? (dummy | 0xFEu) : st.load(cb[k] & 0x3Fu);
        const uint64_t ent = st.entry_mem(q);
        const uint32_t t = enc_rec_t((uint32_t)(ent >> 32), bb[k]);
        if (byp) dummy = (uint8_t)t;
        else st.store(cb[k] & 0x3Fu, t);
        row[k] = (uint32_t)ent;
        m[k] = (t & 0x00010000u) ? ~0u : 0u;
        ext[k] = byp ? 1u : 0u;
        addm[k] = (byp && bb[k]) ? ~0u : 0u;
      }
    } else {
      kind = 1;
      for (uint32_t k = 0; k < 16; k++) {
        row[k] = kEncRecNone;
        m[k] = 0;
        if (!(full || i0 + k < n)) continue;
        const uint32_t c = cb[k];
        if (c < 64u) {
          const uint64_t ent = st.entry_mem(st.load(c));
          const uint32_t t = enc_rec_t((uint32_t)(ent >> 32), bb[k]);
          st.store(c, t);
          row[k] = (uint32_t)ent;
          m[k] = (t & 0x00010000u) ? ~0u : 0u;
        } else if (c == kBypass || c == kProbBase + 50u) {
          row[k] = kEncRecBypass | (uint32_t)bb[k] << 1;
        } else if (is_prob_byte(c) && prob_valid(c - kProbBase)) {
          row[k] = kProbRow[prob_row_index(c - kProbBase)];
          m[k] = bb[k] != (c - kProbBase > 50u ? 1u : 0u) ? ~0u : 0u;
        }
      }
    }
    if (kinds) kinds[kind]++;
    // wave B
    for (int g = 0; g < 4; g++) {
      EncWordAcc acc{0, 0};
      for (int k = 0; k < 4; k++) {
        const int i = 4 * g + k;
        if (kind == 0) enc_rec_plain(e, acc, k == 0, row[i], m[i]);
        else if (kind == 2) enc_rec_mixed(e, acc, k == 0, row[i], m[i], ext[i], addm[i]);
        else enc_rec_general(e, row[i], m[i]);
      }
      if (kind != 1) {
        if (acc.S > g_word_shift_max) g_word_shift_max = acc.S;
        enc_word_close(e, acc);
      }
      if (e.W >= 39) enc_flush16_wave(e, sink);
      if (e.W > 62) return -2;
    }
  }
  enc_flush16(e, sink);
  enc_finish16(e, sink);
  return sink.len <= cap ? (long)sink.len : -1;
}

// the stream's payload is buf[start, start+len) inside a buffer of `total` bytes.
// chunk = bins per boundary() (kernel: 16), group = bins per top-up (kernel: 4)
uint32_t lane_decode_stream(const uint8_t *buf, uint64_t total, uint64_t start, uint32_t len, const uint8_t *ctx,
                            uint32_t n, const uint8_t *ctx_init, int nctx, int chunk, int group, uint8_t *bins_out,
                            uint8_t *ctx_final, uint32_t *counters /* [fast, slow, slow_refills] */) {
  const bool any = group < 0;
  if (any) group = -group;
  // kernel fast path: one bit-granular top-up per 4 bins (no availability check), only the 4th bin guarded
  const bool mixedw = group == 6;  // ... with the select-based step for context-coded and bypass bins alike (dec_bin_mixed)
  if (mixedw) group = 3;
  const bool lat = group == 5;  // ... in the latency-oriented instance's order (state one bin ahead, forwarding)
  if (lat) group = 3;
  bool memtab = group == 3;  // ... with the byte-table bin step and (mps, lps) pair collection of the kernel
  if (memtab) group = 2;
  uint32_t lat_at_prev = 0, lat_nx_prev = 0, lat_qld = 0;
  bool lat_have_prev = false, lat_have_qld = false;
  const bool noguard = group == 2;
  if (noguard) group = 4;
  uint32_t pairs = 0;
  bool has_bypass = any;  // the kernel guards with 8 instead of 7 when a chunk holds bypass bins
  // the batch kernels' rule for fixed-probability bins: ctx byte kProbBase + p in batches of at most 128 contexts
  auto prob_of = [&](uint32_t i) -> uint32_t { return nctx <= 128 && is_prob_byte(ctx[i]) ? ctx[i] - kProbBase : 0u; };
  auto is_ep = [&](uint32_t i) { return ctx[i] == kBypass || prob_of(i) == 50u; };
  for (uint32_t i = 0; i < n; i++) has_bypass |= is_ep(i);
  const int32_t min_sh = has_bypass ? 8 : 7;
  HostStates st;
  st.init(ctx_init, nctx);
  st.by_path = true;
  uint32_t fast = 0, slow = 0, slow_refills = 0;
  // like the kernel: the window works from the 4-byte aligned address below the stream's first byte
  const uint64_t abase = start & ~3ull;
  const uint32_t skew = (uint32_t)(start - abase);
  WindowSrc<HostMem> src;
  src.mem = HostMem{buf + abase, total - abase, &fast, &slow};
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  for (uint32_t i = 0; i < n; i++) {
    if (i % (uint32_t)chunk == 0) src.boundary();
    else if (noguard && i % 16 == 8) src.ensure(47);  // 64 bits after boundary() cover two top-ups; one check for the next two
    if (i % (uint32_t)group == 0) {
      if (noguard) {
        if (src.avail_bits() < 23 - d.sh) return 0xDEAD1000u | i;  // the no-check top-up would be wrong
        dec_topup_nocheck(d, src);
      } else {
        dec_topup(d, src);
      }
    }
    if (noguard && i % 4 == 3) dec_guard(d, src, min_sh);
    if (noguard) {
      if (d.sh < (is_ep(i) ? 8 : 7)) return 0xDEAD0000u | i;  // the no-guard argument would be wrong
    } else if (d.sh < 8) {
      dec_refill(d, src);
      slow_refills++;
    }
    if (memtab) {  // bypass bins join the pair format as (bin, 0), like the kernel's mixed chunks
      if (mixedw && nctx <= 128 && (ctx[i] < 128 || ctx[i] == kBypass)) {
        lat_have_prev = lat_have_qld = false;
        const uint32_t c = ctx[i], mb = (c & 0x80u) ? ~0u : 0u;
        pairs = collect_bin(pairs, dec_bin_mixed(d, mb, st.addr(c & (nctx <= 64 ? 0x3Fu : 0x7Fu)), st));
      } else if (prob_of(i) && !is_ep(i)) {  // fixed-probability bin: no context model, joins the pair format like a bypass bin
        lat_have_prev = lat_have_qld = false;
        const uint32_t pp = prob_of(i);
        pairs = collect_bin(pairs, prob_valid(pp) ? pair_of_bin(dec_bin_prob(d, pp, kProbRow[prob_row_index(pp)])) : pair_of_bin(0u));
      } else if (lat && !is_ep(i)) {
        const uint32_t at = st.addr(ctx[i]);
        if (!lat_have_qld) lat_qld = st.load_at(at);  // first bin of a run: nothing was read ahead
        const uint32_t q = (lat_have_prev && at == lat_at_prev) ? lat_nx_prev : lat_qld;
        lat_have_qld = i + 1 < n && ctx[i + 1] < 128 && (i + 1) % 16 != 0;  // the kernel reads ahead inside a chunk only
        if (lat_have_qld) lat_qld = st.load_at(st.addr(ctx[i + 1]));             // before this bin's store
        pairs = collect_bin(pairs, dec_bin_lat(d, at, q, st, lat_nx_prev));
        lat_at_prev = at;
        lat_have_prev = lat_have_qld;  // forwarding only ever pairs with a read-ahead value
      } else {
        lat_have_prev = lat_have_qld = false;
        pairs = collect_bin(pairs, is_ep(i) ? pair_of_bin(dec_bin_ep(d)) : dec_bin_at_mem(d, st.addr(ctx[i]), st));
      }
      if (i % 16 == 15 || i + 1 == n) {  // a chunk's worth: pairs -> bins -> bytes, like the kernel's store path
        const uint32_t done = i % 16 + 1;
        const uint32_t bw = pairs_to_bins(pairs) >> (2 * (16 - done));  // a short last chunk sits at the top
        for (uint32_t w = 0; w < 4; w++) {
          const uint32_t b4 = collected_bytes(bw, (int)w);
          for (uint32_t k = 0; k < 4; k++)
            if (4 * w + k < done) bins_out[i + 1 - done + 4 * w + k] = (uint8_t)(b4 >> (8 * k));
        }
        pairs = 0;
      }
      continue;
    }
    if (prob_of(i) && !is_ep(i)) bins_out[i] = prob_valid(prob_of(i)) ? (uint8_t)dec_bin_prob(d, prob_of(i), kProbRow[prob_row_index(prob_of(i))]) : 0;
    else if (is_ep(i)) bins_out[i] = (uint8_t)dec_bin_ep(d);
    else if (any) bins_out[i] = (uint8_t)dec_bin_any(d, ctx[i], st);
    else bins_out[i] = (uint8_t)dec_bin(d, ctx[i], st);
  }
  const uint8_t *p = buf + start;
  uint32_t status = dec_finish(d, src.bpos - 8u * skew, len, [p](uint32_t i) { return (uint32_t)p[i]; });
  if (ctx_final) st.final_packed(ctx_final, nctx);
  if (counters) {
    counters[0] = fast;
    counters[1] = slow;
    counters[2] = slow_refills;
  }
  return status;
}


// The decode kernel's chunk structure (cabac_kernels.hip: k_decode) on one stream: 16-bin chunks classified like the
// kernel does -- plain (16 context bins: dec_bin_out, no guard), mixed (context + bypass bins: dec_bin_mixed_nx, no guard),
// general (anything else: per-bin branches, the fourth bin of a word guarded) --, state bytes with the stale bin in bit 0,
// output bytes masked with 1 at the end like store_out.  lat: the latency-oriented instance's order (state one bin ahead,
// successor forwarded for a repeated context).  Returns the status word, or 0xDEADxxxx if an unguarded bin ran short.
}  // extern "C"
// Src: the payload source -- PieceSrc on 16-byte aligned pieces (k_decode) or WindowSrc on 4-byte aligned ones (the pair, wide
// and symbol kernels)
template <class Src, uint32_t ALIGN>
static uint32_t decode_stream_out_t(const uint8_t *buf, uint64_t total, uint64_t start, uint32_t len, const uint8_t *ctx,
                                    uint32_t n, const uint8_t *ctx_init, int nctx, int lat, uint8_t *bins_out,
                                    uint8_t *ctx_final, uint32_t *path_counts /* [plain, mixed, general] chunks */) {
  const int cm = nctx <= 64 ? 1 : nctx <= 128 ? 3 : 2;
  const uint32_t lim = std::min(4u * (((uint32_t)nctx + 3u) / 4u), 255u);  // cm 2: contexts of the state image (k_decode: ctx_lim)
  auto is_ctx = [&](uint32_t c) { return cm == 1 ? c < 64u : cm == 3 ? c < 128u : c < lim; };
  auto is_prob = [&](uint32_t c) { return cm != 2 && is_prob_byte(c); };
  HostStates st;
  st.init(ctx_init, nctx);
  const uint64_t abase = start & ~(uint64_t)(ALIGN - 1u);
  const uint32_t skew = (uint32_t)(start - abase);
  Src src;
  src.mem = HostMem{buf + abase, total - abase, nullptr, nullptr};
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  DecAbs da = dec_to_abs(d);  // the kernel's loop-carried core: R = range << sh
  uint32_t bad = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += 16) {
    src.boundary();
    uint8_t cb[16];
    for (uint32_t k = 0; k < 16; k++) cb[k] = i0 + k < n ? ctx[i0 + k] : 0;  // (the kernel reads whatever the row holds)
    const bool full = i0 + 16 <= n;
    bool hi = false, odd = false;
    uint32_t inv64 = 0;  // (through the kernel's own SWAR test, four bytes at a time)
    for (uint32_t k = 0; k < 16; k++) {
      hi |= cm == 2 ? cb[k] >= lim : (cb[k] & (cm == 1 ? 0xC0u : 0x80u)) != 0;
      odd |= (cb[k] & 0x80u) && cb[k] != 0xFFu;
    }
    for (uint32_t k = 0; k < 16; k += 4) inv64 |= not_ctx64_nor_bypass((uint32_t)cb[k] | (uint32_t)cb[k + 1] << 8 | (uint32_t)cb[k + 2] << 16 | (uint32_t)cb[k + 3] << 24);
    {
      bool want = false;
      for (uint32_t k = 0; k < 16; k++) want |= cb[k] >= 64u && cb[k] != 0xFFu;
      if (want != (inv64 != 0)) bad = 0xDEAD2000u;  // the SWAR test must say what the definition says
    }
    uint32_t ow[4] = {0, 0, 0, 0}, oa = 0, ob = 0;
    if (full && !hi) {
      if (path_counts) path_counts[0]++;
      uint32_t at = cb[0], at_prev = 0, nx_prev = 0, qld = lat ? st.load_at(at) : 0;
      for (int g = 0; g < 4; g++) {
        if (g == 2) src.ensure(47);
        if (src.avail_bits() < (int32_t)clz32(da.R)) bad = 0xDEAD1000u;
        dec_topup_abs(da, src);
        for (int k = 0; k < 4; k++) {
          const int i = 4 * g + k;
          if (lat) {
            const uint32_t q = (i > 0 && at == at_prev) ? nx_prev : qld;
            uint32_t at_next = 0;
            if (i < 15) {
              at_next = cb[i + 1];
              qld = st.load_at(at_next);
            }
            nx_prev = k == 0 ? dec_bin_nx_abs<true>(da, at, q, st) : dec_bin_nx_abs<false>(da, at, q, st);
            if (k == 0) put_out_byte<0>(oa, ob, nx_prev);
            else if (k == 1) put_out_byte<1>(oa, ob, nx_prev);
            else if (k == 2) put_out_byte<2>(oa, ob, nx_prev);
            else put_out_byte<3>(oa, ob, nx_prev);
            at_prev = at;
            at = at_next;
          } else {
            const uint32_t a = cb[i];
            if (k == 0) dec_bin_out_abs<0, true>(da, a, st.load_at(a), st, oa, ob);
            else if (k == 1) dec_bin_out_abs<1, false>(da, a, st.load_at(a), st, oa, ob);
            else if (k == 2) dec_bin_out_abs<2, false>(da, a, st.load_at(a), st, oa, ob);
            else dec_bin_out_abs<3, false>(da, a, st.load_at(a), st, oa, ob);
          }
          if (da.R < 512u) bad = 0xDEAD0000u | (i0 + i);  // sh >= 1 (the range is 256..510): R >> 1 stays exact
        }
        ow[g] = out_word(oa, ob);
      }
    } else if (cm == 1 && full && inv64 == 0) {
      // <= 64 contexts: the mixed step on a dummy context (dec_bin_mixed2_nx_abs); a bypass byte's state address is the dummy
      if (path_counts) path_counts[1]++;
      for (int g = 0; g < 4; g++) {
        if (g == 2) src.ensure(47);
        if (src.avail_bits() < (int32_t)clz32(da.R)) bad = 0xDEAD1000u;
        dec_topup_abs(da, src);
        for (int k = 0; k < 4; k++) {
          const uint32_t c = cb[4 * g + k], a = c == 0xFFu ? 256u : c;
          const uint32_t nx = k == 0 ? dec_bin_mixed2_nx_abs<true>(da, a, st.load_at(a), st) : dec_bin_mixed2_nx_abs<false>(da, a, st.load_at(a), st);
          if (k == 0) put_out_byte<0>(oa, ob, nx);
          else if (k == 1) put_out_byte<1>(oa, ob, nx);
          else if (k == 2) put_out_byte<2>(oa, ob, nx);
          else put_out_byte<3>(oa, ob, nx);
          if (da.R < 256u) bad = 0xDEAD0000u | (i0 + 4 * g + k);
          if ((c == 0xFFu) != (st.cm[256] >= kDecBypassRaw && a == 256u)) bad = 0xDEAD3000u;  // the dummy stays a bypass byte
        }
        ow[g] = out_word(oa, ob);
      }
    } else if (cm == 3 && full && !odd) {
      if (path_counts) path_counts[1]++;
      const uint32_t in = 0x7Fu;
      for (int g = 0; g < 4; g++) {
        if (g == 2) src.ensure(47);
        if (src.avail_bits() < (int32_t)clz32(da.R)) bad = 0xDEAD1000u;
        dec_topup_abs(da, src);
        for (int k = 0; k < 4; k++) {
          const uint32_t c = cb[4 * g + k], a = c & in;
          const uint32_t mbk = (c & 0x80u) ? ~0u : 0u;
          const uint32_t nx = k == 0 ? dec_bin_mixed_nx_abs<true>(da, mbk, a, st.load_at(a), st)
                                     : dec_bin_mixed_nx_abs<false>(da, mbk, a, st.load_at(a), st);
          if (k == 0) put_out_byte<0>(oa, ob, nx);
          else if (k == 1) put_out_byte<1>(oa, ob, nx);
          else if (k == 2) put_out_byte<2>(oa, ob, nx);
          else put_out_byte<3>(oa, ob, nx);
          if (da.R < 256u) bad = 0xDEAD0000u | (i0 + 4 * g + k);
        }
        ow[g] = out_word(oa, ob);
      }
    } else {
      if (path_counts) path_counts[2]++;
      dec_from_abs(d, da);
      for (int g = 0; g < 4; g++) {
        if (g == 2) src.ensure(47);
        if (src.avail_bits() < 23 - d.sh) bad = 0xDEAD1000u;
        dec_topup_nocheck(d, src);
        for (int k = 0; k < 4; k++) {
          if (k == 3) dec_guard(d, src, 8);
          uint32_t w = 0;
          if (i0 + 4 * g + k < n) {
            const uint32_t c = cb[4 * g + k];
            if (is_ctx(c)) w = dec_bin_nx(d, c, st.load_at(c), st);
            else if (c == kBypass || c == kProbBase + 50u) w = dec_bin_ep(d);
            else if (is_prob(c) && prob_valid(c - kProbBase)) w = dec_bin_prob(d, c - kProbBase, kProbRow[prob_row_index(c - kProbBase)]);
            if (d.sh < 0) bad = 0xDEAD0000u | (i0 + 4 * g + k);
          }
          if (k == 0) put_out_byte<0>(oa, ob, w);
          else if (k == 1) put_out_byte<1>(oa, ob, w);
          else if (k == 2) put_out_byte<2>(oa, ob, w);
          else put_out_byte<3>(oa, ob, w);
        }
        ow[g] = out_word(oa, ob);
      }
      da = dec_to_abs(d);
    }
    for (uint32_t k = 0; k < 16 && i0 + k < n; k++) bins_out[i0 + k] = (uint8_t)((ow[k >> 2] >> (8 * (k & 3))) & 1u);
  }
  if (bad) return bad;
  const uint8_t *p = buf + start;
  dec_from_abs(d, da);
  const uint32_t status = dec_finish(d, src.bpos - 8u * skew, len, [p](uint32_t i) { return (uint32_t)p[i]; });
  if (ctx_final) st.final_packed(ctx_final, nctx);
  return status;
}
extern "C" {
uint32_t lane_decode_stream_out(const uint8_t *buf, uint64_t total, uint64_t start, uint32_t len, const uint8_t *ctx, uint32_t n,
                                const uint8_t *ctx_init, int nctx, int lat, uint8_t *bins_out, uint8_t *ctx_final, uint32_t *path_counts) {
  // bit 1 of lat: the previous source (WindowSrc); default: k_decode's PieceSrc
  if (lat & 2) return decode_stream_out_t<WindowSrc<HostMem>, 4u>(buf, total, start, len, ctx, n, ctx_init, nctx, lat & 1, bins_out, ctx_final, path_counts);
  return decode_stream_out_t<PieceSrc<HostMem>, 16u>(buf, total, start, len, ctx, n, ctx_init, nctx, lat & 1, bins_out, ctx_final, path_counts);
}

// one stream through the reference-formulation decoder (RefDec): what the engine uses for streams starting with 0xFF
uint32_t lane_decode_stream_ref(const uint8_t *payload, uint32_t len, const uint8_t *ctx, uint32_t n, const uint8_t *ctx_init,
                                int nctx, uint8_t *bins_out) {
  HostStates st;
  st.init(ctx_init, nctx);
  RefReader rd{payload, len, 0, 0};
  RefDec r;
  ref_dec_start(r, rd);
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t pp = nctx <= 128 && is_prob_byte(ctx[i]) ? ctx[i] - kProbBase : 0u;
    if (ctx[i] == kBypass || pp == 50u) bins_out[i] = (uint8_t)ref_dec_bin_ep(r, rd);
    else if (pp) bins_out[i] = prob_valid(pp) ? (uint8_t)ref_dec_bin_prob(r, pp, kProbRow[prob_row_index(pp)], rd) : 0;
    else bins_out[i] = (uint8_t)ref_dec_bin(r, ctx[i], st, rd);
  }
  return ref_dec_finish(r, rd, len);
}

// the 16-bit next-state table of the encode kernel against the one-entry-per-state table: mismatches over all
// (state 0..62, mps, mps/lps) combinations
int lane_next16_mismatches() {
  int bad = 0;
  for (uint32_t s = 0; s < 63; s++)
    for (uint32_t mps = 0; mps < 2; mps++)
      for (uint32_t lps = 0; lps < 2; lps++) {
        const uint32_t q = (s << 2) | (mps << 1), m = lps ? ~0u : 0u;
        const uint32_t want = next_state_q(q, m, kNext.e[s]) & 0xFFu;
        const uint32_t got = next_state_q16(m, next_state_entry16(q >> 1, kTransLps[s])) & 0xFFu;
        bad += want != got;
        bad += want != next_state_by_bin(q >> 1, lps ? 1u - mps : mps, kTransLps[s]);
      }
  return bad;
}

// the kernels' packed helpers: state-byte offsets of four contexts, and 16 decoded bins collected two bits apiece
void lane_ctx_quad(uint32_t cw, uint32_t *out4) {
  const CtxQuad z = lds_ctx_quad(cw);
  for (int k = 0; k < 4; k++) out4[k] = quad_field(z, k);
}
void lane_collect16(const uint32_t *bin_words, uint8_t *bytes16) {
  uint32_t acc = 0xDEADBEEFu;
  for (int j = 0; j < 16; j++) acc = collect_bin(acc, bin_words[j]);
  for (int w = 0; w < 4; w++) {
    const uint32_t b = collected_bytes(acc, w);
    std::memcpy(bytes16 + 4 * w, &b, 4);
  }
}

// symbol-level front end (cabac_symbols.h): params = {method, k, nq, selector, types, nlbp, col_rows}
long sym_encode_stream(const uint8_t *sym, uint32_t nsym, const uint32_t *params, const uint8_t *ctx_init, int nctx,
                       uint8_t *out, size_t cap, uint32_t *nbins_out) {
  const SymParams P{params[0], params[1], params[2], params[3], params[4], params[5], params[6]};
  HostStates st;
  st.init(ctx_init, nctx);
  HostSink sink{out, 0, cap};
  EncCore e;
  enc_start(e);
  SymWalk w;
  w.start();
  uint32_t nbins = 0;
  while (w.idx < nsym) {
    if (w.n == 1) w.cur = binarize(sym[w.idx], P);
    const uint32_t c = w.ctx(P);
    enc_bin(e, ((w.cur.bits >> (w.n - 1)) & 1u) << 1, c, st);
    enc_flush(e, sink);
    nbins++;
    if (w.n == w.cur.len) w.next_symbol(P);
    else w.n++;
  }
  enc_finish(e, sink);
  if (nbins_out) *nbins_out = nbins;
  return sink.len <= cap ? (long)sink.len : -1;
}

// The code tree of the symbol kernels' fast path (cabac_symbols.h: SymTrie) against the general walker: for the symbol
// sequence `sym`, the context of every bin and the symbol found at every leaf must be what SymWalk / binarize give.
// Returns the number of tree nodes, 0 if the parameters do not get a tree, -1 - (bin index) at the first disagreement.
long sym_trie_check(const uint8_t *sym, uint32_t nsym, const uint32_t *params) {
  const SymParams P{params[0], params[1], params[2], params[3], params[4], params[5], params[6]};
  static SymTrie T;
  if (!sym_trie_build(P, &T)) return 0;
  SymWalk w;
  w.start();
  uint32_t node = 0, up = kTrieNoUp, row = 0;
  long nb = 0;
  while (w.idx < nsym) {
    if (w.n == 1) {
      w.cur = binarize(sym[w.idx], P);
      node = 0;
    }
    if (T.ctx[node * kTrieUp + up] != w.ctx(P)) return -1 - nb;
    const uint32_t bin = (w.cur.bits >> (w.n - 1)) & 1u;
    const uint32_t nx = T.child[node * 2 + bin];
    nb++;
    if (w.n == w.cur.len) {
      if (nx != (0x80u | sym[w.idx])) return -1 - nb;
      up = sym[w.idx];
      row++;
      if (row == P.col_rows) {
        row = 0;
        up = P.selector == 0 ? up : kTrieNoUp;
      }
      w.next_symbol(P);
    } else {
      if (nx & 0x80u) return -1 - nb;
      node = nx;
      w.n++;
    }
  }
  return (long)T.nodes;
}

uint32_t sym_decode_stream(const uint8_t *payload, uint32_t len, uint32_t nsym, const uint32_t *params,
                           const uint8_t *ctx_init, int nctx, uint8_t *sym_out) {
  const SymParams P{params[0], params[1], params[2], params[3], params[4], params[5], params[6]};
  HostStates st;
  st.init(ctx_init, nctx);
  WindowSrc<HostMem> src;
  src.mem = HostMem{payload, len, nullptr, nullptr};
  src.len = len;
  src.start(0);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  SymWalk w;
  w.start();
  uint32_t it = 0;
  while (w.idx < nsym) {
    if ((it++ & 15u) == 0) src.boundary();
    if (d.sh < 8) dec_refill(d, src);
    const uint32_t c = w.ctx(P);
    const uint32_t bin = dec_bin(d, c, st);
    w.cur = append_bin(w.cur, w.n, bin);
    const bool finished = symbol_finished(w.cur, P);
    if (finished || w.n == 32) {
      sym_out[w.idx] = finished ? (uint8_t)debinarize(w.cur, P) : (uint8_t)0;
      w.next_symbol(P);
    } else {
      w.n++;
    }
  }
  return dec_finish(d, src.bpos, len, [payload](uint32_t i) { return (uint32_t)payload[i]; });
}

}  // extern "C"
