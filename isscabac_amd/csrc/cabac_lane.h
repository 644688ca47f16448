// cabac_lane.h -- the per-stream (= per-lane) CABAC arithmetic of the MI355X engine.
//
// One wavefront lane owns one bitstream, so everything in here is lane-private sequential integer
// work: no cross-lane traffic, no MFMA.  The functions are written against two small policies so
// that the very same code is compiled into the HIP kernels (cabac_kernels.hip: context states in LDS
// or a VGPR, tables in LDS or through ds_bpermute, bytes to an HBM slot) and into a host harness used by the
// CPU test-suite (tests/lane_harness.cpp: plain arrays) -- the harness is a test build of THIS code,
// not a fallback.
//
// What is restated (ref: = /root/reference):
//   enc_bin / enc_bin_ep  CABAC_ArithmeticEncoder::encodeBin / encodeBinEP   ref: CABAC/CABAC_ArithmeticEncoder.cpp:113-178, 250-270
//   enc_flush / enc_lead  testAndWriteOut / writeOut                         ref: :369-412
//   enc_finish            encodeBinTrm(1) + finish                           ref: :326-367, :70-105
//   dec_bin / dec_bin_ep  CABAC_ArithmeticDecoder::decodeBin / decodeBinEP   ref: CABAC/CABAC_ArithmeticDecoder.cpp:87-190, 288-331
//   dec_finish            decodeBinTrm + finish                              ref: :423-472, :73-85
//   next_mps / next_lps   ContextModel::updateMPS / updateLPS                ref: CABAC/ContextModel.cpp:107-123,136-158
//
// How it differs from the reference's formulation (results are bit-identical):
//   * encoder `low` is 64 bit wide and bytes are moved out every few bins instead of after every bin
//     (the byte sequence is the same big number either way); renorm count = clz32(lps)-23 instead of
//     sm_aucRenormTable (equal for every reachable LPS value, states 0..62);
//   * decoder keeps look-ahead bits in `v` (topped up bit-granularly to sh = 23) and moves the comparison
//     point (`sh`) instead of shifting the value and pulling one byte at a time; the MPS/LPS decision only
//     depends on bits the reference has also read, so every decoded bin is identical.
#ifndef ISSCABAC_LANE_H
#define ISSCABAC_LANE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CABAC_HD __host__ __device__ __forceinline__
#define CABAC_HD_RARE __host__ __device__ __forceinline__
#else
#define CABAC_HD inline
#define CABAC_HD_RARE inline
#endif
// "does any lane of the wave see x": wave-uniform on the device, the lane itself on the host harness
#if defined(__HIP_DEVICE_COMPILE__)
#define CABAC_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
// "does any lane have a < b" (signed) for the guards that sit on the per-word path: one v_cmp into an SGPR pair and one
// scalar test.  Around a nested per-lane test the ballot idiom above compiles to three VALU and five SALU instructions
// on this target (the lane mask is materialised as 0/1 and compared again).  CABAC_RARE_BODY() opens the guarded block
// so that the per-lane test inside is not hoisted in front of the branch and merged with it.
static __device__ __forceinline__ bool cabac_wave_any_lt(int32_t a, int32_t b) {
  uint64_t m;
  asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m != 0ull;
}
#define CABAC_WAVE_ANY_LT(a, b) cabac_wave_any_lt((int32_t)(a), (int32_t)(b))
#define CABAC_RARE_BODY() asm volatile("")
#else
#define CABAC_WAVE_ANY(x) (x)
#define CABAC_WAVE_ANY_LT(a, b) ((int32_t)(a) < (int32_t)(b))
#define CABAC_RARE_BODY() ((void)0)
#endif

namespace isscabac {

constexpr uint32_t kBypass = 0xFFu;

// rangeTabLps (ITU-T H.265 Table 9-46 / H.264 Table 9-44), one dword per state, byte q = column q.
// Same values as sm_aucLPSTable, ref: CABAC/CABAC_ArithmeticEncoder.cpp:414-480.
#define ISSCABAC_LPS_ROWS                                                                            \
  0xF0D0B080u, 0xE3C5A780u, 0xD8BB9E80u, 0xCDB2967Bu, 0xC3A98E74u, 0xB9A0876Fu, 0xAF988069u,         \
  0xA6907A64u, 0x9E89745Fu, 0x96826E5Au, 0x8E7B6855u, 0x87756351u, 0x806F5E4Du, 0x7A695949u,         \
  0x74645545u, 0x6E5F5042u, 0x685A4C3Eu, 0x6356483Bu, 0x5E514538u, 0x594D4135u, 0x55493E33u,         \
  0x50453B30u, 0x4C42382Eu, 0x483F352Bu, 0x453B3229u, 0x41383027u, 0x3E362D25u, 0x3B332B23u,         \
  0x38302921u, 0x352E2720u, 0x322B251Eu, 0x3029231Du, 0x2D27211Bu, 0x2B251F1Au, 0x29231E18u,         \
  0x27211C17u, 0x25201B16u, 0x231E1A15u, 0x211D1814u, 0x1F1B1713u, 0x1E1A1612u, 0x1C191511u,         \
  0x1B171410u, 0x1916130Fu, 0x1815120Eu, 0x1714110Eu, 0x1613100Du, 0x15120F0Cu, 0x14110E0Cu,         \
  0x13100E0Bu, 0x120F0D0Bu, 0x110F0C0Au, 0x100E0C0Au, 0x0F0D0B09u, 0x0E0C0B09u, 0x0E0C0A08u,         \
  0x0D0B0908u, 0x0C0B0907u, 0x0C0A0907u, 0x0B0A0807u, 0x0B090806u, 0x0A090706u, 0x09080706u,         \
  0x02020202u

// transIdxLps (H.265 Table 9-45), the state part of m_aucNextStateLPS, ref: CABAC/ContextModel.cpp:148-158
#define ISSCABAC_TRANS_LPS                                                                           \
  0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, \
  22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34,    \
  35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63

// sm_aucLPSTProbTable[49][4] (ref: CABAC/CABAC_ArithmeticEncoder.cpp:494-547, the decoder's copy CABAC_ArithmeticDecoder.cpp:554-):
// the rangeTabLps rows of the fixed-probability bins, one dword per row, byte q = column q; row i serves an MPS
// probability of 51 + i percent.
#define ISSCABAC_PROB_ROWS                                                                           \
  0xEACAAB8Bu, 0xE5C6A788u, 0xE0C2A486u, 0xDCBEA083u, 0xD7BA9D80u, 0xD2B6997Du, 0xCDB2967Au,         \
  0xC9AE9277u, 0xC4A98F74u, 0xBFA58B72u, 0xBAA1886Fu, 0xB59D846Cu, 0xB1998169u, 0xAC957E66u,         \
  0xA7917A63u, 0xA28C7761u, 0x9E88735Eu, 0x9984705Bu, 0x94806C58u, 0x8F7C6955u, 0x8A786552u,         \
  0x86746250u, 0x81705E4Du, 0x7C6B5B4Au, 0x77675747u, 0x73635444u, 0x6E5F5041u, 0x695B4D3Eu,         \
  0x6457493Cu, 0x5F534639u, 0x5B4E4236u, 0x564A3F33u, 0x51463B30u, 0x4C42382Du, 0x483E342Bu,         \
  0x433A3128u, 0x3E362D25u, 0x39322A22u, 0x352D261Fu, 0x3029231Cu, 0x2B251F1Au, 0x26211C17u,         \
  0x211D1814u, 0x1D191511u, 0x1815110Eu, 0x13110E0Bu, 0x0E0C0A09u, 0x0A080706u, 0x05040404u

// Fixed-probability bins (ref: encodeBinProb / decodeBinProb, compiled under RWTH_CABAC_FIXED_PROBABILITY): a ctx byte of
// kProbBase + p marks a bin whose probability of being 1 is p percent; no context model is involved.  p = 50 is a
// bypass bin (ref: :188-192).  p = 46..49 and 51..54 are refused: with those four rows the LPS exceeds half of the
// smallest range, the reference's MPS path then leaves m_uiRange below 256 (:228-236 shifts by exactly one bit) and its
// decoder indexes sm_aucLPSTable out of bounds on the next context bin (CABAC_ArithmeticDecoder.cpp:94).
constexpr uint32_t kProbBase = 155u;
CABAC_HD bool is_prob_byte(uint32_t c) { return c > kProbBase && c != kBypass; }
CABAC_HD bool prob_valid(uint32_t p) { return p >= 1u && p <= 99u && !(p >= 46u && p <= 54u && p != 50u); }
CABAC_HD uint32_t prob_row_index(uint32_t p) { return (p < 50u ? 50u - p : p - 50u) - 1u; }  // p != 50

CABAC_HD uint32_t clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }

// x << n for results that fit 16 bits -- the renormalised range (9 bits), table addresses (11 bits).  On gfx950
// v_lshlrev_b32 issues at half rate and v_lshlrev_b16 at full rate (tools/ubench/valu_ops2: 1.77 against 0.99 ns per
// instruction and SIMD at four waves; right shifts are full rate either way); the 16-bit form writes zeros to bits 16..31.
// x must be below 2^16 and n below 16.
// ... but not for the decode kernels' table address: there the 16-bit shift (inline assembly) measured SLOWER, 2.879 against
// 2.812 ms (medians of 8 on one box) -- the compiler places its own shift better than one it cannot see into
CABAC_HD uint32_t shl16(uint32_t x, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_lshlrev_b16_e32 %0, %1, %2" : "=v"(r) : "v"(n), "v"(x));
  return r;
#else
  return (x << n) & 0xFFFFu;
#endif
}
template <int K>
CABAC_HD uint32_t shl16c(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  static_assert(K == 2 || K == 3, "the shifts the table lookups use");
  if (K == 2) asm("v_lshlrev_b16_e32 %0, 2, %1" : "=v"(r) : "v"(x));
  else asm("v_lshlrev_b16_e32 %0, 3, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return (x << K) & 0xFFFFu;
#endif
}

// Context states are kept "doubled": q = m_ucState << 1 = (state << 2) | (mps << 1).  q is directly the byte
// address of entry `state` in a table of dwords (and the ds_bpermute address of lane `state`), and the
// mps sits at bit 1 where one xor with a pre-shifted bin decides LPS/MPS.
CABAC_HD uint32_t q_from_packed(uint32_t p) { return p << 1; }
CABAC_HD uint32_t packed_from_q(uint32_t q) { return q >> 1; }

// rangeTabLps[state][(range >> 6) & 3] out of the state's table row (range is 256..510)
CABAC_HD uint32_t lps_of(uint32_t row, uint32_t range) {
#if defined(__HIP_DEVICE_COMPILE__)
  // range is 256..510, so range >> 6 is 4..7 with zeros above: as a v_perm_b32 selector its byte 0 picks byte
  // (range >> 6) - 4 of the high operand (row) and its bytes 1..3 pick byte 0 of the low operand (0) -- the entry,
  // clean, in two instructions
  return __builtin_amdgcn_perm(row, 0u, range >> 6);
#else
  return (row >> ((range >> 3) & 0x18u)) & 0xFFu;
#endif
}

// Next-state table entry of one state s, both transitions in one word so that a single lookup serves
// ContextModel::updateLPS and updateMPS (ref: CABAC/ContextModel.cpp:107-123, tables :136-158), in the
// doubled representation and without the mps bit (xor the old mps bit back in):
//   bits 0..7   = min(s + 1, 62) << 2                       -- MPS successor
//   bits 10..17 = (transIdxLps[s] << 2) | (s == 0 ? 2 : 0)  -- LPS successor; the MPS flips only when an LPS hits state 0
// The LPS field sits at bit 10 (not 8) so that the shift amount `lps_mask & 10` doubles as "bit 1 = this bin was an
// LPS": q ^ (lps_mask & 10) then carries the decoded bin (mps ^ lps) at bit 1 (bin_word_b1).
constexpr uint32_t kLpsShift = 10;
constexpr uint32_t next_state_entry(uint32_t s, uint32_t trans) {
  return ((s < 62 ? s + 1 : s) << 2) | (((trans << 2) | (s == 0 ? 2u : 0u)) << kLpsShift);
}
// lps_mask: 0 for an MPS, ~0 for an LPS.  Only the low byte of the result is meaningful (it is stored as a byte).
CABAC_HD uint32_t next_state_q(uint32_t q, uint32_t lps_mask, uint32_t entry) {
  return (entry >> (lps_mask & kLpsShift)) ^ (q & 2u);
}
// The same transitions as a table of 128 16-bit entries indexed by the packed state p = q >> 1 = (state << 1) | mps,
// each holding the complete successors (mps included): bits 0..7 after an MPS, bits 8..15 after an LPS.  Entry p sits
// at byte address q, so an LDS lookup needs no address arithmetic, and the update is (entry >> (lps_mask & 8)).
constexpr uint32_t next_state_entry16(uint32_t p, uint32_t trans) {
  return (((p >> 1) < 62 ? (p >> 1) + 1 : (p >> 1)) << 2 | (p & 1u) << 1) |
         ((trans << 2 | ((p & 1u) ^ ((p >> 1) == 0 ? 1u : 0u)) << 1) << 8);
}
CABAC_HD uint32_t next_state_q16(uint32_t lps_mask, uint32_t entry16) { return entry16 >> (lps_mask & 8u); }
// ... and as 256 bytes indexed by (p << 1) | bin = q | bin: the successor of packed state p after coding `bin`
constexpr uint32_t next_state_by_bin(uint32_t p, uint32_t bin, uint32_t trans) {
  return (next_state_entry16(p, trans) >> (bin == (p & 1u) ? 0 : 8)) & 0xFFu;
}
// States policies may provide next_q(q, lps_mask, entry) for their own entry format; the default is next_state_q
template <class States>
CABAC_HD auto next_q_of(const States &st, uint32_t q, uint32_t m, uint32_t nx, int) -> decltype(st.next_q(q, m, nx)) {
  return st.next_q(q, m, nx);
}
template <class States>
CABAC_HD uint32_t next_q_of(const States &, uint32_t q, uint32_t m, uint32_t nx, long) {
  return next_state_q(q, m, nx);
}

// a word whose bit 1 is the value of the bin just coded with state q (bit 0 is 0, the rest is garbage)
CABAC_HD uint32_t bin_word_b1(uint32_t q, uint32_t lps_mask) { return q ^ (lps_mask & kLpsShift); }

// v_perm_b32: byte i of the result = byte number sel.byte[i] of the eight bytes {hi, lo} (0..3 = lo, 4..7 = hi)
CABAC_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t x = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) r |= (uint32_t)((x >> (8u * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
  return r;
#endif
}
// v_alignbit_b32: the low 32 bits of {hi, lo} >> s
CABAC_HD uint32_t alignbit_b32(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
#endif
}
// Decoded bins are collected two bits per bin: one v_alignbit moves bits 1:0 of a bin word into bits 31:30 of the
// accumulator and everything older two places down.  After 16 bins, bin j of the 16 sits at bit 2j + 1.
CABAC_HD uint32_t collect_bin(uint32_t acc, uint32_t bin_word) { return alignbit_b32(bin_word, acc, 2); }
// bins 4w .. 4w+3 of a collected 16-bin word as four bytes of 0/1 (bin 4w in byte 0): the bins sit at bits 1,3,5,7 of
// byte w and the even bits are 0 (bit 0 of every collected word is 0); (1 + 2^6 + 2^12 + 2^18) moves bit 2i to bit 8i
// and no carry of the product reaches those positions
CABAC_HD uint32_t collected_bytes(uint32_t acc, int w) {
  return (((acc >> (8 * w + 1)) & 0x7Fu) * 0x00041041u) & 0x01010101u;  // shift + mask = one v_bfe_u32
}

// Offsets of four contexts' state bytes in the kernels' LDS state image, from the four ctx bytes of one input word:
// context c lives at (c / 4) * 256 + lane * 4 + c % 4 (see LdsStates), i.e. offset X(c) = ((c >> 2) << 8) | (c & 3)
// before the lane term.  z01 = X(c0) | X(c1) << 16, z23 = X(c2) | X(c3) << 16: 3 + 2 instructions per four bins
// instead of 4 per bin, and the per-bin address is one add with a 16-bit operand select.
// 0x80 in every byte of a ctx word that has bit 7 set without being the bypass marker 0xFF (fixed-probability bins,
// invalid bytes): such chunks take the general path.  (b & 0x7F) + 1 reaches bit 7 exactly for b & 0x7F == 0x7F.
CABAC_HD uint32_t odd_high_bytes(uint32_t cw) { return cw & ~((cw & 0x7F7F7F7Fu) + 0x01010101u) & 0x80808080u; }
// ~0 if bit (8 k + bit) of w is set
CABAC_HD uint32_t byte_bit_mask(uint32_t w, int k, int bit) { return (uint32_t)((int32_t)(w << (31 - 8 * k - bit)) >> 31); }

struct CtxQuad {
  uint32_t z01, z23;
};
CABAC_HD CtxQuad lds_ctx_quad(uint32_t cw) {
  const uint32_t lo2 = cw & 0x03030303u, hi6 = (cw >> 2) & 0x3F3F3F3Fu;
  CtxQuad z;
  z.z01 = perm_b32(hi6, lo2, 0x05010400u);
  z.z23 = perm_b32(hi6, lo2, 0x07030602u);
  return z;
}
CABAC_HD uint32_t quad_field(const CtxQuad &z, int k) {
  return k == 0 ? (z.z01 & 0xFFFFu) : k == 1 ? (z.z01 >> 16) : k == 2 ? (z.z23 & 0xFFFFu) : (z.z23 >> 16);
}
// x, but the compiler may assume nothing about its value (stops it from narrowing a select of two zero-extended bytes to a
// byte select followed by a mask); no instruction
CABAC_HD uint32_t opaque(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}
// (m & a) | (~m & b) in one instruction (v_bfi_b32); written out for the host build
CABAC_HD uint32_t select_mask(uint32_t m, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
#else
  return (m & a) | (~m & b);
#endif
}
// ~0 if bit 1 of x is set, else 0
CABAC_HD uint32_t mask_from_bit1(uint32_t x) { return (uint32_t)((int32_t)(x << 30) >> 31); }

// ------------------------------------------------------------------------------------------------
// Encoder
// ------------------------------------------------------------------------------------------------
struct EncCore {
  uint64_t low;       // W valid bits, a possible carry at bit W
  uint32_t range;     // 256..510 between bins
  int32_t W;          // number of valid bits in low  (reference: 32 - m_bitsLeft)
  uint32_t buffered;  // m_bufferedByte
  uint32_t nbuf;      // m_numBufferedBytes
};

CABAC_HD void enc_start(EncCore &e) {  // ref: CABAC_ArithmeticEncoder.cpp:54-68
  e.low = 0;
  e.range = 510;
  e.W = 9;  // bitsLeft = 23
  e.buffered = 0xFF;
  e.nbuf = 0;
}

// States policy: load(ctx) -> doubled state q, store(ctx, q) (stores the low byte), lps_row(q), next_entry(q)
// (plus addr / load_at / store_at for the quad variants).  b1: any word whose bit 1 is the bin.
// Bin policy: xor_q(q) -> a word whose bit 1 is bin ^ mps (mps = bit 1 of q).  BinB1 wraps a word whose bit 1 is the bin.
struct BinB1 {
  uint32_t b1;
  CABAC_HD uint32_t xor_q(uint32_t q) const { return b1 ^ q; }
};
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_at(EncCore &e, Bin bin, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint32_t row = st.lps_row(q);
  const uint32_t nx = st.next_entry(q);  // unconditional: the lookup may be a cross-lane op
  const uint32_t lps = lps_of(row, e.range);  // column (range >> 6) & 3
  const uint32_t rmps = e.range - lps;
  const uint32_t m = mask_from_bit1(bin.xor_q(q));  // ~0: bin != mps
  // new range before renormalisation; its leading zeros give the shift for both paths:
  // LPS: clz32(lps) - 23 == sm_aucRenormTable[lps >> 3] (lps in 6..240); MPS: rmps in 128..510 -> 1 or 0
  const uint32_t r = rmps ^ ((rmps ^ lps) & m);
  const uint32_t n = clz32(r) - 23u;
  e.low = (e.low + (rmps & m)) << n;
  e.range = r << n;
  e.W += (int32_t)n;
  st.store_at(at, next_q_of(st, q, m, nx, 0));
}
// The same step for a States policy whose tables are byte tables in memory: lps_mem(q, range) -> rangeTabLps entry,
// next_mem(q | bin) -> successor state (Bin::or_q(q) = q | bin, the bin at bit 0).  Both lookups need one address
// instruction and no extraction, and the state update waits neither for the decision nor for the range arithmetic.
// ... and on the single-lookup entries (enc_bin_row_acc's: entry_mem(q) -> row | successors by bin | flags), `low` updated at once:
// k_encode's step for chunks it codes lane by lane (fixed-probability bins, ragged ends) -- with it the kernel's only table is the
// 1 KiB of entries.  Same results as enc_bin_at_mem.
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_at_row(EncCore &e, Bin bin, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(q);
  const uint32_t t = bin.next_of((uint32_t)(ent >> 32));
  st.store_at(at, t);
  const uint32_t m = bin.lps_mask(q, t);  // ~0: bin != mps
  const uint32_t lps = lps_of((uint32_t)ent, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);
  const uint32_t n = clz32(r) - 23u;
  e.low = (e.low + (rmps & m)) << n;
  e.range = r << n;
  e.W += (int32_t)n;
}
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_at_mem(EncCore &e, Bin bin, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  st.store_at(at, st.next_mem(bin.or_q(q)));  // table indexed by (state, mps, bin): no decision needed first
  const uint32_t m = mask_from_bit1(bin.xor_q(q));  // ~0: bin != mps
  const uint32_t lps = st.lps_mem(q, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);
  const uint32_t n = clz32(r) - 23u;
  e.low = (e.low + (rmps & m)) << n;
  e.range = r << n;
  e.W += (int32_t)n;
}

// The additions four bins make to `low`, accumulated in 32 bits, so that the 64-bit `low` is shifted and added to once
// per input word instead of once per bin (64-bit shifts and adds issue at half rate on CDNA4):
//   ((((low + a0) << n0) + a1) << n1 ...) = (low << S) + T,   T = (((a0 << n0) + a1) << n1 ...),   S = n0 + n1 + n2 + n3.
// a < 2^9, and a != 0 (LPS) implies n >= 1, so the non-zero terms of T sit at strictly decreasing shifts <= S and
// T < 2^(S + 10).  S <= 22 for any four context bins: a 6-bit shift needs an LPS of 6 or 7 (states 61, 62 in columns 0, 1),
// which leaves range = 384 or 448, columns 2 or 3, where every LPS is >= 8 -- so the worst run is 6,5,6,5 (exhaustive
// search over range x state x path: tests/lane_harness.cpp: lane_max_shift_sum).  T is therefore exact in 32 bits.
struct EncWordAcc {
  uint32_t T, S;
};
// high word of the encoder's 8-byte entry for packed state ps (= q >> 1; its mps is bit 0), see enc_bin_row_acc
constexpr uint32_t enc_entry_hi(uint32_t ps, uint32_t succ_bin0, uint32_t succ_bin1) {
  return succ_bin0 | succ_bin1 << 8 | ((ps & 1u) ? 0xFFu : 0u) << 16 | ((ps & 1u) ? 0u : 0xFFu) << 24;  // flag bytes 0x00 / 0xFF
}
// The encoder's step for one bin of an input word (first: the word's first bin) -- everything except `low` and W, which
// enc_word_close updates -- on ONE table lookup: entry_mem(q) -> 64 bits by packed state (q >> 1): the state's 4-byte rangeTabLps row
// in the low word, its two successors -- by bin value: byte 0 for a bin of 0, byte 1 for a bin of 1 -- in the high word,
// followed by two flags at bits 16 and 24: "a bin of 0 / of 1 is not this state's mps" (enc_entry_hi), so that the shift
// that picks the successor also brings the LPS flag to bit 16.
// Three LDS operations per bin instead of four (an 8-byte read costs the LDS pipe what a 1-byte read does, measured:
// tools/ubench/lds_tp.hip), two round trips behind each other instead of three, and none of them inside the
// range -> LPS -> range recurrence: the LPS column is picked from the row in a register (lps_of).
// Bin policy: next_of(hi) -> hi >> (8 * bin), lps_mask(q, t) -> ~0 if the bin is not q's mps (t = next_of(hi)).
// The two things the LPS decision picks: r = the range the bin leaves (before renormalisation), a = what `low` gains.
// Default: by the policy's mask; the encode kernel's policies overload it with a compare on the flag byte and two
// selects on VCC (cheaper to issue than sign-extract + bit-select + and).
template <class Bin>
CABAC_HD void enc_pick(const Bin &bin, uint32_t q, uint32_t t, uint32_t lps, uint32_t rmps, uint32_t &r, uint32_t &a) {
  const uint32_t m = bin.lps_mask(q, t);  // ~0: bin != mps
  r = select_mask(m, lps, rmps);
  a = rmps & m;
}
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_row_acc(EncCore &e, EncWordAcc &g, bool first, Bin bin, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(q);
  const uint32_t t = bin.next_of((uint32_t)(ent >> 32));  // low byte: the successor; bit 16: the bin is not q's mps
  st.store_at(at, t);
  const uint32_t lps = lps_of((uint32_t)ent, e.range);
  const uint32_t rmps = e.range - lps;
  uint32_t r, a;
  enc_pick(bin, q, t, lps, rmps, r, a);
  const uint32_t n = clz32(r) - 23u;
  g.T = ((first ? 0u : g.T) + a) << n;
  g.S = (first ? 0u : g.S) + n;
  e.range = shl16(r, n);
}
CABAC_HD void enc_word_close(EncCore &e, const EncWordAcc &g) {
  e.low = (e.low << g.S) + g.T;
  e.W += (int32_t)g.S;
}

// One select-based step for chunks that mix context-coded and bypass bins (BASELINE config 5: every other bin is a
// bypass bin), again accumulated per input word: with  T' = ((T + a) << n) + b  a context bin has (a, n, b) =
// (rmps & m, shift, 0) and a bypass bin (0, 1, bin ? range : 0) -- low = 2 low + bin * range, ref:
// CABAC_ArithmeticEncoder.cpp:250-270.  mb = ~0 for a bypass bin, 0 for a context bin; bmask = ~0 for a bypass bin of 1
// (the caller forms it for a whole input word at once).
// A bypass lane still reads some state (its address is made valid by the caller) and writes it back unchanged; its table
// lookup goes by kEncBypassQ -- state 63, which no context ever holds (refused at init, no successor of any state): the
// tables carry an all-zero entry there, so the lane finds LPS = 0 and "the bin is the mps" and its range passes through the
// step untouched (it is normalised: shift 0) with nothing added before the shift.  No lane branches.
// The bounds of EncWordAcc hold: a bypass bin adds one bit and a term below 2^9 like a context bin with shift 1.
constexpr uint32_t kEncBypassQ = 63u << 2;
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_mixed_acc(EncCore &e, EncWordAcc &g, bool first, Bin bin, uint32_t mb, uint32_t bmask, Addr at,
                                States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(select_mask(mb, kEncBypassQ, q));  // one lookup: row + successors by bin + LPS flags
  const uint32_t t = bin.next_of((uint32_t)(ent >> 32));
  st.store_at(at, select_mask(mb, q, t));
  const uint32_t m = bin.lps_mask(q, t);  // ~0: a context bin that is not its mps
  const uint32_t lps = lps_of((uint32_t)ent, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);  // a bypass bin: the range itself
  const uint32_t n = clz32(r) - 23u;             // ... 0
  const uint32_t nt = n - mb;                    // ... which moves low by one bit all the same
  g.T = (((first ? 0u : g.T) + (rmps & m)) << nt) + (e.range & bmask);
  g.S = (first ? 0u : g.S) + nt;
  e.range = shl16(r, n);
}

// The mixed step without its three selects (round 4).  A bypass bin is given a state of its own to read and write -- a dummy
// context behind the state image, whose byte only ever receives the all-zero entry's successor (0) --, its table index is made
// kEncBypassQ's neighbour 0xFE by OR-ing a per-word byte (mbfe) into whatever it read, and the three per-bin quantities that
// depend on the kind of the bin come out of per-word SWAR words by operand select: cwm (bytes of 0 / 1: the extra shift of a
// bypass bin), mbfe (0 / 0xFE), wbm (1 where a bypass bin is 1: times the range = what `low` gains).  Same results as
// enc_bin_mixed_acc; device only (the Bin policy's *_byte members are SDWA instructions).
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_mixed2_acc(EncCore &e, EncWordAcc &g, bool first, Bin bin, uint32_t cwm, uint32_t mbfe, uint32_t wbm, Addr at,
                                 States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(bin.or_byte(q, mbfe));  // a bypass bin: entry 0xFE >> 1 = 127, all zero
  const uint32_t t = bin.next_of((uint32_t)(ent >> 32));
  st.store_at(at, t);                                       // (a bypass bin: 0 into its dummy byte)
  const uint32_t m = bin.lps_mask_sext(t);                  // ~0: a context bin that is not its mps
  const uint32_t lps = lps_of((uint32_t)ent, e.range);      // a bypass bin: 0
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);             // ... the range itself
  const uint32_t n = clz32(r) - 23u;                        // ... 0
  const uint32_t nt = bin.add_byte(n, cwm);                 // ... which moves low by one bit all the same
  g.T = (((first ? 0u : g.T) + (rmps & m)) << nt) + bin.mul_byte(e.range, wbm);
  g.S = (first ? 0u : g.S) + nt;
  e.range = shl16(r, n);
}

// The mixed step with the dummy context HOLDING the bypass state (round 6; batches of <= 64 contexts): the dummy's byte is 0xFE
// (packed state 127 = state 63, no context's state) and the table's entry for it has an all-zero row and 0xFE as both successors,
// so a bypass bin needs nothing OR-ed into what it read -- it reads 0xFE, finds LPS 0 and "the bin is the mps", and writes 0xFE
// back -- and, as in the decoder (dec_bin_mixed2_nx_abs), its state address is the dummy row by itself under a 5-bit group mask.
// What is left of enc_bin_mixed2_acc's extras: the bypass bin's one-bit shift (cwm) and `low` += range for a bypass bin of 1 (wbm).
constexpr uint32_t kEncBypassHi = 0x0000FEFEu;  // high word of the encoder's bypass entry (packed state 127): successors 0xFE, flags 0
template <class States, class Addr, class Bin>
CABAC_HD void enc_bin_mixed3_acc(EncCore &e, EncWordAcc &g, bool first, Bin bin, uint32_t cwm, uint32_t wbm, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(q);
  const uint32_t t = bin.next_of((uint32_t)(ent >> 32));
  st.store_at(at, t);                                       // (a bypass bin: 0xFE into its dummy byte again)
  const uint32_t m = bin.lps_mask_sext(t);                  // ~0: a context bin that is not its mps
  const uint32_t lps = lps_of((uint32_t)ent, e.range);      // a bypass bin: 0
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);             // ... the range itself
  const uint32_t n = clz32(r) - 23u;                        // ... 0
  const uint32_t nt = bin.add_byte(n, cwm);                 // ... which moves low by one bit all the same
  g.T = (((first ? 0u : g.T) + (rmps & m)) << nt) + bin.mul_byte(e.range, wbm);
  g.S = (first ? 0u : g.S) + nt;
  e.range = shl16(r, n);
}

// Latency-oriented form of enc_bin_row_acc for the low-occupancy kernel instance (few waves per SIMD: nothing hides an
// LDS round trip, so none may sit on a bin-to-bin dependency chain).  The caller hands in the state q -- loaded one bin
// ahead, with the previous bin's successor forwarded in a register when both bins use the same context.  In two halves,
// because the kernel issues the first by hand (cabac_kernels.hip: EncLatPipe) and runs the arithmetic of bin k - 1 behind
// the lookup of bin k:
//   enc_lat_lookup  the lookup of a bin: returns the successor (one clean byte: to store and to forward) and the row
//   enc_lat_arith   range / low arithmetic of a bin from its state and row
// Same arithmetic, same results.
template <class States, class Bin>
CABAC_HD uint32_t enc_lat_lookup(Bin bin, uint32_t q, States &st, uint32_t &row) {
  const uint64_t ent = st.entry_mem(q);
  row = (uint32_t)ent;
  return bin.next_of((uint32_t)(ent >> 32)) & 0xFFu;
}
template <class Bin>
CABAC_HD void enc_lat_arith(EncCore &e, EncWordAcc &g, bool first, Bin bin, uint32_t q, uint32_t row) {
  const uint32_t m = mask_from_bit1(bin.xor_q(q));  // ~0: bin != mps
  const uint32_t lps = lps_of(row, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);
  const uint32_t n = clz32(r) - 23u;
  g.T = ((first ? 0u : g.T) + (rmps & m)) << n;
  g.S = (first ? 0u : g.S) + n;
  e.range = shl16(r, n);
}
template <class States, class Addr, class Bin>
CABAC_HD uint32_t enc_bin_lat_acc(EncCore &e, EncWordAcc &g, bool first, Bin bin, Addr at, uint32_t q, States &st) {
  uint32_t row;
  const uint32_t nx = enc_lat_lookup(bin, q, st, row);
  st.store_at(at, nx);
  enc_lat_arith(e, g, first, bin, q, row);
  return nx;
}

template <class States>
CABAC_HD void enc_bin(EncCore &e, uint32_t b1, uint32_t ctx, States &st) {
  enc_bin_at(e, BinB1{b1}, st.addr(ctx), st);
}

// context-coded or bypass bin in one branch-free step (ctx == kBypass selects encodeBinEP); lets a wave
// whose lanes mix both kinds stay convergent
template <class States>
CABAC_HD void enc_bin_any(EncCore &e, uint32_t b1, uint32_t ctx, States &st) {
  const bool ep = ctx == kBypass;
  const uint32_t c = ep ? 0u : ctx;
  const uint32_t q = st.load(c);
  const uint32_t row = st.lps_row(q);
  const uint32_t nx = st.next_entry(q);
  const uint32_t lps = lps_of(row, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t m = ep ? 0u : mask_from_bit1(b1 ^ q);
  const uint32_t r = ep ? e.range : (rmps ^ ((rmps ^ lps) & m));
  const uint32_t n = ep ? 1u : clz32(r) - 23u;
  e.low = ((e.low + (rmps & m)) << n) + ((ep && (b1 & 2u)) ? e.range : 0u);
  e.range = ep ? e.range : (r << n);
  e.W += (int32_t)n;
  const uint32_t qn = next_q_of(st, q, m, nx, 0);
  st.store(c, ep ? q : qn);
}

CABAC_HD void enc_bin_ep(EncCore &e, uint32_t bin) {  // ref: :250-270
  e.low = (e.low << 1) + (bin ? e.range : 0u);
  e.W += 1;
}

// encodeBinProb for p != 50 (ref: :180-243); row = ISSCABAC_PROB_ROWS[prob_row_index(p)].  For the rows prob_valid()
// admits, rmps >= 128 and lps is 4..215, so the shift is the same clz32 - 23 as for a context bin (<= 6 bits).
CABAC_HD void enc_bin_prob(EncCore &e, uint32_t bin, uint32_t p, uint32_t row) {
  const uint32_t mps = p > 50u ? 1u : 0u;
  const uint32_t lps = lps_of(row, e.range);
  const uint32_t rmps = e.range - lps;
  const bool is_lps = bin != mps;
  const uint32_t r = is_lps ? lps : rmps;
  const uint32_t n = clz32(r) - 23u;
  e.low = (e.low + (is_lps ? rmps : 0u)) << n;
  e.range = r << n;
  e.W += (int32_t)n;
}

// ------------------------------------------------------------------------------------------------
// The wave-pair encoder (cabac_pair.hip: k_encode_pair) splits a bin between two waves: wave A runs the context-state pass
// and leaves one RECORD per bin, wave B codes the records.  The split is exact because the states follow from bins and
// contexts alone (ref: CABAC_ArithmeticEncoder.cpp:113-178 reads the state, never the other way round).
//   record of a context-coded (or fixed-probability) bin: row = the state's rangeTabLps row, m = ~0 if the bin is not the mps
//   PLAIN / MIXED chunks: a bypass bin is row 0, m 0 (its range passes through untouched) with ext = 1 (its one bit of shift)
//                         and addm = ~0 if the bin is 1 (`low` gains the range, ref: :250-270)
//   GENERAL chunks:       the row word itself says what the bin is: kEncRecNone (nothing: past the stream's end or an
//                         invalid index), kEncRecBypass | bin << 1, or a row (real rows have no zero byte: they are >= 4)
// enc_rec_t: what A reads off the entry's high word (enc_entry_hi) for a bin: low byte = successor state, bits 16..23 = the
// flag byte "this bin is not the state's mps" (0x00 / 0xFF).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kEncRecNone = 0u, kEncRecBypass = 1u;
CABAC_HD uint32_t enc_rec_t(uint32_t entry_hi, uint32_t bin) { return entry_hi >> (8u * bin); }
CABAC_HD void enc_rec_plain(EncCore &e, EncWordAcc &g, bool first, uint32_t row, uint32_t m) {
  const uint32_t lps = lps_of(row, e.range);
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);
  const uint32_t n = clz32(r) - 23u;
  g.T = ((first ? 0u : g.T) + (rmps & m)) << n;
  g.S = (first ? 0u : g.S) + n;
  e.range = shl16(r, n);
}
CABAC_HD void enc_rec_mixed(EncCore &e, EncWordAcc &g, bool first, uint32_t row, uint32_t m, uint32_t ext, uint32_t addm) {
  const uint32_t lps = lps_of(row, e.range);  // a bypass bin: 0
  const uint32_t rmps = e.range - lps;
  const uint32_t r = select_mask(m, lps, rmps);  // ... the range itself
  const uint32_t n = clz32(r) - 23u;             // ... 0
  const uint32_t nt = n + ext;
  g.T = (((first ? 0u : g.T) + (rmps & m)) << nt) + (e.range & addm);
  g.S = (first ? 0u : g.S) + nt;
  e.range = shl16(r, n);
}
CABAC_HD void enc_rec_general(EncCore &e, uint32_t row, uint32_t m) {
  if (row >= 4u) {  // enc_bin_at_mem / enc_bin_prob: the same arithmetic
    const uint32_t lps = lps_of(row, e.range);
    const uint32_t rmps = e.range - lps;
    const uint32_t r = select_mask(m, lps, rmps);
    const uint32_t n = clz32(r) - 23u;
    e.low = (e.low + (rmps & m)) << n;
    e.range = r << n;
    e.W += (int32_t)n;
  } else if (row & 1u) {
    enc_bin_ep(e, row >> 1);
  }
}

// one 9-bit lead value through the outstanding-byte logic of writeOut, ref: :388-411
template <class Sink>
CABAC_HD void enc_lead(EncCore &e, uint32_t lead, Sink &out) {
  if (lead == 0xFFu) {
    e.nbuf++;
  } else if (e.nbuf > 0) {
    const uint32_t carry = lead >> 8;
    out.put((e.buffered + carry) & 0xFFu);
    const uint32_t fill = (0xFFu + carry) & 0xFFu;
    while (e.nbuf > 1) {
      out.put(fill);
      e.nbuf--;
    }
    e.buffered = lead & 0xFFu;
  } else {
    e.nbuf = 1;
    e.buffered = lead;
  }
}

// move complete bytes out of low.  The reference does this after every bin (bitsLeft < 12 <=> W >= 21);
// here it may run less often, W just has to stay below 63.
template <class Sink>
CABAC_HD void enc_flush(EncCore &e, Sink &out) {
  while (e.W >= 21) {
    e.W -= 8;
    const uint32_t lead = (uint32_t)(e.low >> e.W) & 0x1FFu;
    e.low &= (1ull << e.W) - 1ull;
    enc_lead(e, lead, out);
  }
}

template <class Sink>
CABAC_HD void enc_finish(EncCore &e, Sink &out) {
  // encodeBinTrm(1), ref: :326-341
  e.range -= 2;
  e.low = (e.low + e.range) << 7;
  e.range = 256;
  e.W += 7;
  enc_flush(e, out);  // now 13 <= W <= 20, as in the reference after its testAndWriteOut
  // finish(), ref: :76-104
  if ((e.low >> e.W) & 1ull) {
    out.put((e.buffered + 1u) & 0xFFu);
    while (e.nbuf > 1) {
      out.put(0x00u);
      e.nbuf--;
    }
    e.low -= 1ull << e.W;
  } else {
    if (e.nbuf > 0) out.put(e.buffered & 0xFFu);
    while (e.nbuf > 1) {
      out.put(0xFFu);
      e.nbuf--;
    }
  }
  // write(low >> 8, 24 - bitsLeft); write(1, 1); writeAlignZero()
  const uint32_t nb = (uint32_t)e.W - 8u;  // 5..12 payload bits
  uint32_t tail = ((((uint32_t)(e.low >> 8)) & ((1u << nb) - 1u)) << 1) | 1u;  // nb + 1 bits
  if (nb + 1u <= 8u) {
    out.put((tail << (8u - (nb + 1u))) & 0xFFu);
  } else {
    tail <<= 16u - (nb + 1u);
    out.put(tail >> 8);
    out.put(tail & 0xFFu);
  }
}

// ------------------------------------------------------------------------------------------------
// Encoder output in 16-bit digits (used by the batch bin kernel)
// ------------------------------------------------------------------------------------------------
// Same outstanding-digit logic as writeOut, on 16-bit digits: `buffered` holds the last digit that may still take a
// carry, `nbuf` - 1 the number of 0xFFFF digits behind it.  A lane then moves output every ~20 bins instead of every
// ~10, and a 4-bin check almost never needs a second pass.  The byte stream is the same big number; enc_finish16
// converts the pending digits to the byte-granular state and ends exactly like the reference's finish().
// Sink additionally provides put16(x): the two bytes of x, high byte first.
template <class Sink>
CABAC_HD void enc_lead16(EncCore &e, uint32_t lead, Sink &out) {
  if (lead == 0xFFFFu) {
    e.nbuf++;
  } else if (e.nbuf > 0) {
    const uint32_t carry = lead >> 16;
    out.put16((e.buffered + carry) & 0xFFFFu);
    const uint32_t fill = (0xFFFFu + carry) & 0xFFFFu;
    while (e.nbuf > 1) {
      out.put16(fill);
      e.nbuf--;
    }
    e.buffered = lead & 0xFFFFu;
  } else {
    e.nbuf = 1;
    e.buffered = lead;  // the very first digit cannot carry
  }
}

// keeps 13 <= W' after every digit like the reference's byte-wise writeOut (W' >= 9 is what bounds a later carry to one
// bit; 25 measured no faster than 29); W must stay below 63
#ifndef ISSCABAC_EMIT_THR
#define ISSCABAC_EMIT_THR 29
#endif
template <class Sink>
CABAC_HD void enc_flush16(EncCore &e, Sink &out) {
  while (e.W >= ISSCABAC_EMIT_THR) {
    e.W -= 16;
    const uint32_t lead = (uint32_t)(e.low >> e.W) & 0x1FFFFu;
    e.low &= (1ull << e.W) - 1ull;
    enc_lead16(e, lead, out);
  }
}

// The same for a whole wave at a point where some lane has to move output: the ordinary case -- a lane has at most one
// digit ready, it is not 0xFFFF, and exactly one digit is pending behind it -- runs as one straight, predicated piece of
// code; anything else (a 0xFFFF digit has probability 2^-16) takes the general loop above.
template <class Sink>
CABAC_HD void enc_flush16_wave(EncCore &e, Sink &out) {
  const int32_t wn = e.W - 16;
  // digit, with the carry at bit 16 and nothing above it (low < 2^(W+1)); meaningless unless W >= 16
  const uint32_t top = (uint32_t)(e.low >> (wn & 63));
  // plain: exactly one digit ready (29 <= W < 45, so what stays behind fits 32 bits), not 0xFFFF, one digit pending
  const bool plain = (uint32_t)(wn - (ISSCABAC_EMIT_THR - 16)) < 16u && (top & 0xFFFFu) != 0xFFFFu && e.nbuf == 1u;
  if (plain) {
    out.put16(e.buffered + (top >> 16));  // put16 keeps the low 16 bits
    e.buffered = top & 0xFFFFu;
    e.W = wn;
    e.low = (uint32_t)e.low & ((1u << wn) - 1u);
  }
  if (CABAC_WAVE_ANY(e.W >= ISSCABAC_EMIT_THR)) enc_flush16(e, out);  // whatever was not plain
}

template <class Sink>
CABAC_HD void enc_finish16(EncCore &e, Sink &out) {
  // encodeBinTrm(1), ref: :326-341
  e.range -= 2;
  e.low = (e.low + e.range) << 7;
  e.range = 256;
  e.W += 7;
  enc_flush16(e, out);  // W < ISSCABAC_EMIT_THR
  // pending digits [buffered][0xFFFF x (nbuf-1)] -> pending bytes [b][0xFF x (n-1)] of the byte-granular machine
  if (e.nbuf == 0) {
    e.buffered = 0xFFu;
  } else {
    const uint32_t hi = e.buffered >> 8, lo = e.buffered & 0xFFu;
    if (lo == 0xFFu) {
      e.buffered = hi;
      e.nbuf = 2 * e.nbuf;
    } else {
      out.put(hi);  // a carry into `lo` cannot reach it any more
      e.buffered = lo;
      e.nbuf = 2 * e.nbuf - 1;
    }
  }
  enc_flush(e, out);  // at most one more byte: W <= 20, as in the reference after its testAndWriteOut
  // finish(), ref: :76-104
  if ((e.low >> e.W) & 1ull) {
    out.put((e.buffered + 1u) & 0xFFu);
    while (e.nbuf > 1) {
      out.put(0x00u);
      e.nbuf--;
    }
    e.low -= 1ull << e.W;
  } else {
    if (e.nbuf > 0) out.put(e.buffered & 0xFFu);
    while (e.nbuf > 1) {
      out.put(0xFFu);
      e.nbuf--;
    }
  }
  const uint32_t nb = (uint32_t)e.W - 8u;
  uint32_t tail = ((((uint32_t)(e.low >> 8)) & ((1u << nb) - 1u)) << 1) | 1u;
  if (nb + 1u <= 8u) {
    out.put((tail << (8u - (nb + 1u))) & 0xFFu);
  } else {
    tail <<= 16u - (nb + 1u);
    out.put(tail >> 8);
    out.put(tail & 0xFFu);
  }
}

// ------------------------------------------------------------------------------------------------
// Decoder
// ------------------------------------------------------------------------------------------------
// v holds the code value with (sh) fraction bits below the 9-bit range: the reference's m_uiValue is
// v >> (sh - 7) with its not-yet-read low bits zero.  A context bin needs sh >= 7 and takes at most 6 bits, a bypass
// bin needs sh >= 8 and takes 1; sh <= 23 keeps v inside 32 bits.
struct DecCore {
  uint32_t v;
  uint32_t range;
  int32_t sh;
};

// Src policy: uint32_t next() -> next payload byte (0xFF past the end, like fstream::get at EOF);
// for the top-ups additionally feed(v, g) / feed_nocheck(v, g) -> (v << g) | next g payload bits, g in 0..23
template <class Src>
CABAC_HD void dec_refill(DecCore &d, Src &src) {
  while (d.sh <= 15) {
    d.v = (d.v << 8) | src.next();
    d.sh += 8;
  }
}

// bit-granular top-up used at fixed points of the bin loop: fills v up to sh = 23
template <class Src>
CABAC_HD void dec_topup(DecCore &d, Src &src) {
  d.v = src.feed(d.v, (uint32_t)(23 - d.sh));
  d.sh = 23;
}
// the same when the caller knows the window holds the bits (the kernels re-base the window every 8 bins: two top-ups
// of at most 24 bits each): no availability test, no branch
template <class Src>
CABAC_HD void dec_topup_nocheck(DecCore &d, Src &src) {
  d.v = src.feed_nocheck(d.v, (uint32_t)(23 - d.sh));
  d.sh = 23;
}
// After a top-up three context bins can never run short (23 - 2 * 6 >= 7); before a fourth one the wave checks and,
// practically never, tops up the lanes that need it.
template <class Src>
CABAC_HD void dec_guard(DecCore &d, Src &src, int32_t min_sh) {
  if (CABAC_WAVE_ANY_LT(d.sh, min_sh)) {
    CABAC_RARE_BODY();
    if (d.sh < min_sh) dec_topup(d, src);
  }
}

template <class Src>
CABAC_HD void dec_start(DecCore &d, Src &src) {  // ref: CABAC_ArithmeticDecoder.cpp:54-71
  d.range = 510;
  d.v = src.next() << 8;
  d.v |= src.next();
  d.sh = 7;
  dec_refill(d, src);
}

// returns a word whose bit 1 is the decoded bin (bit 0 is 0): what collect_bin() takes
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_at_b1(DecCore &d, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint32_t row = st.lps_row(q);
  const uint32_t nx = st.next_entry(q);
  const uint32_t lps = lps_of(row, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
  // one subtract decides and updates: its borrow (v < scaled) is the MPS case
  uint32_t diff;
  const bool mps_path = __builtin_sub_overflow(d.v, scaled, &diff);
  const uint32_t m = mps_path ? 0u : ~0u;  // ~0: LPS
  const uint32_t r = mps_path ? rmps : lps;
  const uint32_t n = clz32(r) - 23u;
  d.v = mps_path ? d.v : diff;
  d.range = r << n;
  d.sh -= (int32_t)n;
  st.store_at(at, next_q_of(st, q, m, nx, 0));
  return bin_word_b1(q, m);
}
// The same step for a States policy with byte tables in memory (see enc_bin_at_mem): lps_mem(q, range) and
// next_path_mem(path_index(q) + path) -> successor state, path = 1 for the MPS path (the borrow of the deciding subtraction, added
// in with one add-with-carry).  Returns q + path: bit 1 = mps, bit 0 = "was an MPS"; sixteen of these collected with
// collect_bin() turn into bin words with pairs_to_bins().
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_at_mem(DecCore &d, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint32_t lps = st.lps_mem(q, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
  uint32_t diff;
  const bool mps_path = __builtin_sub_overflow(d.v, scaled, &diff);
  // the policy may bias the index by a multiple of 4 (a table address): its low two bits stay (mps, path)
  const uint32_t idx = st.path_index(q) + (uint32_t)mps_path;
  st.store_at(at, st.next_path_mem(idx));
  const uint32_t r = mps_path ? rmps : lps;
  const uint32_t n = clz32(r) - 23u;
  d.v = mps_path ? d.v : diff;
  d.range = r << n;
  d.sh -= (int32_t)n;
  return idx;
}
// The same step on ONE table lookup (see enc_bin_row_acc): entry_mem(q) -> the state's rangeTabLps row in the low word and
// its two successors by path in the high word (byte 0: LPS path, byte 1: MPS path).  The lookup depends on q alone, so
// the range -> LPS -> range recurrence never touches memory.  Returns q + path like dec_bin_at_mem; nx receives the
// successor (a clean byte: the latency-oriented instance forwards it as the next state).
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_row_q(DecCore &d, Addr at, uint32_t q, States &st, uint32_t &nx) {
  const uint64_t ent = st.entry_mem(q);
  const uint32_t lps = lps_of((uint32_t)ent, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
#if defined(__HIP_DEVICE_COMPILE__)
  // the decision and everything it selects, on the borrow in VCC: the successor byte comes out of the entry with an
  // SDWA source select (no shift), the value, the new range and the (mps, path) pair with one instruction each
  uint32_t diff, r, idx;
  asm("v_sub_co_u32_e32 %0, vcc, %4, %5\n\t"
      "v_cndmask_b32_sdwa %1, %6, %6, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %7, %8, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %4, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, 0, %9, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r), "=&v"(idx)
      : "v"(d.v), "v"(scaled), "v"((uint32_t)(ent >> 32)), "v"(lps), "v"(rmps), "v"(q)
      : "vcc");
  st.store_at(at, nx);
  const uint32_t n = clz32(r) - 23u;
  d.v = diff;
  d.range = r << n;
  d.sh -= (int32_t)n;
  return idx;
#else
  uint32_t diff;
  const bool mps_path = __builtin_sub_overflow(d.v, scaled, &diff);
  nx = (uint32_t)(ent >> (mps_path ? 40 : 32)) & 0xFFu;
  st.store_at(at, nx);
  const uint32_t r = mps_path ? rmps : lps;
  const uint32_t n = clz32(r) - 23u;
  d.v = mps_path ? d.v : diff;
  d.range = r << n;
  d.sh -= (int32_t)n;
  return q + (uint32_t)mps_path;
#endif
}
// ------------------------------------------------------------------------------------------------
// The decoder's steps with the decoded bin riding in bit 0 of the successor byte.
// A doubled state q = (state << 2) | (mps << 1) is even, so bit 0 of the state byte is free: the decoder's entries hold
// `successor | bin decoded on that path` (dec_entry_hi_out), the byte picked by the decision is at once the value the state
// image takes and -- its bit 0 -- the output.  No (mps, path) pair, no collect, no expansion to bytes afterwards: the
// picked byte lands in its place of the output word (dec_bin_out), garbage above bit 0 is masked off when the tile leaves.
// State bytes in the image therefore carry a stale bin in bit 0 and every decoder lookup goes by the RAW byte:
// States::entry_raw(qb) -> the 8-byte entry of state qb >> 1 = row | dec_entry_hi_out << 32 (256 entries, odd ones
// duplicate their even neighbour); the mps is bit 1 as ever.
// ------------------------------------------------------------------------------------------------
// high word of the raw-byte entry of packed state ps (= qb >> 1, mps = ps & 1) from its two successors:
// byte 0 = LPS path: successor | (mps ^ 1), byte 1 = MPS path: successor | mps
constexpr uint32_t dec_entry_hi_out(uint32_t ps, uint32_t succ_lps, uint32_t succ_mps) {
  return (succ_lps | ((ps & 1u) ^ 1u)) | (succ_mps | (ps & 1u)) << 8;
}
// Bin K of a word (0..3) into the output pair: bins 0 and 1 start oa and ob (byte 0 = the low byte of w, the rest 0), bins 2
// and 3 add byte 2, everything else preserved (one SDWA move on the device)
template <int K>
CABAC_HD void put_out_byte(uint32_t &oa, uint32_t &ob, uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (K == 0) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(oa) : "v"(w));
  else if (K == 1) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(ob) : "v"(w));
  else if (K == 2) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(oa) : "v"(w));
  else asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(ob) : "v"(w));
#else
  uint32_t &o = (K & 1) ? ob : oa;
  if (K & 2) o = (o & ~0x00FF0000u) | ((w & 0xFFu) << 16);
  else o = w & 0xFFu;
#endif
}
// the four bytes of a word's bins in stream order: oa = (bin 0, -, bin 2, -), ob = (bin 1, -, bin 3, -)
CABAC_HD uint32_t out_word(uint32_t oa, uint32_t ob) { return perm_b32(ob, oa, 0x06020400u); }

// Bin K of a word (0..3) on the raw-byte entry, the picked successor byte written straight into the output pair:
// subtract-with-borrow decides, three selects on the borrow.  13 VALU + 3 LDS operations per bin with the state address.
// States: entry_raw(qb), store_at(at, x) (stores the low byte of x), store_at_hi(at, x) (stores byte 2 of x).
// Needs sh >= the bin's own shift; four context bins after a top-up to 23 never run short (they shift <= 22 bits in
// total: tests/lane_harness.cpp lane_max_shift_sum), so the kernels' plain words carry no guard.
template <int K, class States, class Addr>
CABAC_HD void dec_bin_out(DecCore &d, Addr at, uint32_t qb, States &st, uint32_t &oa, uint32_t &ob) {
  const uint64_t ent = st.entry_raw(qb);
  const uint32_t lps = lps_of((uint32_t)ent, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
  uint32_t r;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t &o = (K & 1) ? ob : oa;
  uint32_t diff;
  // bins 0 and 1 of a word start their output register (the other bytes read 0), bins 2 and 3 add byte 2 to it
#define ISSCABAC_DEC_OUT(SEL, UNUSED, OC)                                                                              \
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"                                                                           \
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"   \
      "v_cndmask_b32_e32 %2, %6, %7, vcc\n\t"                                                                          \
      "v_cndmask_b32_e32 %0, %0, %3, vcc"                                                                              \
      : "=&v"(diff), OC(o), "=&v"(r)                                                                                   \
      : "v"(d.v), "v"(scaled), "v"((uint32_t)(ent >> 32)), "v"(lps), "v"(rmps)                                         \
      : "vcc")
  if (K & 2) {
    ISSCABAC_DEC_OUT("BYTE_2", "UNUSED_PRESERVE", "+v");
    st.store_at_hi(at, o);
  } else {
    ISSCABAC_DEC_OUT("BYTE_0", "UNUSED_PAD", "=&v");
    st.store_at(at, o);
  }
#undef ISSCABAC_DEC_OUT
  d.v = diff;
#else
  const bool mps_path = d.v < scaled;
  const uint32_t nx = (uint32_t)(ent >> (mps_path ? 40 : 32)) & 0xFFu;
  put_out_byte<K>(oa, ob, nx);
  st.store_at(at, nx);
  r = mps_path ? rmps : lps;
  d.v = mps_path ? d.v : d.v - scaled;
#endif
  const uint32_t n = clz32(r) - 23u;
  d.range = r << n;
  d.sh -= (int32_t)n;
}
// The same step handing the picked byte back in a register of its own (clean: successor | bin): for the paths that
// forward it (latency-oriented instance) or run under a lane's own exec mask (general path).  One instruction more.
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_nx(DecCore &d, Addr at, uint32_t qb, States &st) {
  const uint64_t ent = st.entry_raw(qb);
  const uint32_t lps = lps_of((uint32_t)ent, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
  uint32_t r, nx;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t diff;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %6, %7, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %3, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r)
      : "v"(d.v), "v"(scaled), "v"((uint32_t)(ent >> 32)), "v"(lps), "v"(rmps)
      : "vcc");
  d.v = diff;
#else
  const bool mps_path = d.v < scaled;
  nx = (uint32_t)(ent >> (mps_path ? 40 : 32)) & 0xFFu;
  r = mps_path ? rmps : lps;
  d.v = mps_path ? d.v : d.v - scaled;
#endif
  st.store_at(at, nx);
  const uint32_t n = clz32(r) - 23u;
  d.range = r << n;
  d.sh -= (int32_t)n;
  return nx;
}
// ... and for words that mix context-coded and bypass bins (see dec_bin_mixed below; mb = ~0 for a bypass bin): a bypass
// lane decides against the whole range one bit further down, picks its bin out of the constant 0x0001 (byte 0: not below
// = 1, byte 1: below = 0) instead of the entry and writes its state byte back unchanged.  Returns the picked byte.
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_mixed_nx(DecCore &d, uint32_t mb, Addr at, uint32_t qb, States &st) {
  const uint64_t ent = st.entry_raw(qb);
  const uint32_t lps = lps_of((uint32_t)ent, d.range);
  const uint32_t rmps = d.range - lps;
  const int32_t sh = d.sh + (int32_t)mb;  // mb = -1: the bypass bin's bit
  const uint32_t scaled = select_mask(mb, d.range, rmps) << sh;
  const uint32_t hi = select_mask(mb, 0x0001u, (uint32_t)(ent >> 32));
  const uint32_t rl = select_mask(mb, d.range, lps), rm = select_mask(mb, d.range, rmps);
  uint32_t r, nx;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t diff;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %6, %7, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %3, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r)
      : "v"(d.v), "v"(scaled), "v"(hi), "v"(rl), "v"(rm)
      : "vcc");
  d.v = diff;
#else
  const bool below = d.v < scaled;
  nx = (hi >> (below ? 8 : 0)) & 0xFFu;
  r = below ? rm : rl;
  d.v = below ? d.v : d.v - scaled;
#endif
  st.store_at(at, select_mask(mb, qb, nx));
  const uint32_t n = clz32(r) - 23u;  // 0 for a bypass bin: its range is normalised
  d.range = r << n;
  d.sh = sh - (int32_t)n;
  return nx;
}

// ------------------------------------------------------------------------------------------------
// The decode kernel's core in absolute form: R = range << sh in place of (range, sh).  v is compared with (range - LPS) << sh
// anyway, so the kernel keeps the range at v's scale and never normalises it: a bin's renormalisation is no work at all
// (R simply becomes the smaller number), the comparison point is read off R when it is needed (sh = 23 - clz(R): the
// 9-bit range has its bit 8 set) and LPS << sh is the only shift left on the recurrence.  Against the step on (range, sh)
// (dec_bin_out): ffbh, two subtractions, a right shift and one left shift instead of ffbh, two subtractions, a right
// shift and two left shifts -- and the first bin after a top-up (R's bit 31 set: FIRST) knows both positions as
// constants.  Same decisions, same values: R == d.range << d.sh at every bin (dec_to_abs / dec_from_abs convert).
// ------------------------------------------------------------------------------------------------
struct DecAbs {
  uint32_t v;
  uint32_t R;
};
CABAC_HD DecAbs dec_to_abs(const DecCore &d) {
  DecAbs a;
  a.v = d.v;
  a.R = d.range << d.sh;
  return a;
}
CABAC_HD void dec_from_abs(DecCore &d, const DecAbs &a) {
  const uint32_t sh = 23u - clz32(a.R);
  d.v = a.v;
  d.sh = (int32_t)sh;
  d.range = a.R >> sh;
}
// bit-granular top-up: R's leading one to bit 31 (sh = 23), v along with it (dec_topup_nocheck)
template <class Src>
CABAC_HD void dec_topup_abs(DecAbs &a, Src &src) {
  const uint32_t c = clz32(a.R);
  a.R <<= c;
  a.v = src.feed_nocheck(a.v, c);
}
// rangeTabLps[state][column] out of the state's row, column + 4 given (4..7, zeros above: see lps_of)
CABAC_HD uint32_t lps_col(uint32_t row, uint32_t col4) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(row, 0u, col4);
#else
  return (row >> (8u * (col4 & 3u))) & 0xFFu;
#endif
}
// LPS << sh for the range R stands for
template <bool FIRST>
CABAC_HD uint32_t abs_lps(uint32_t row, uint32_t R) {
  if (FIRST) return lps_col(row, R >> 29) << 23;
  const uint32_t t = 29u - clz32(R);  // sh + 6
  return lps_col(row, R >> t) << (t - 6u);
}
// dec_bin_out in absolute form
template <int K, bool FIRST, class States, class Addr>
CABAC_HD void dec_bin_out_abs(DecAbs &a, Addr at, uint32_t qb, States &st, uint32_t &oa, uint32_t &ob) {
  const uint64_t ent = st.entry_raw(qb);
  const uint32_t lpss = abs_lps<FIRST>((uint32_t)ent, a.R);
  const uint32_t rmps = a.R - lpss;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t &o = (K & 1) ? ob : oa;
  uint32_t diff, r;
#define ISSCABAC_DEC_OUT(SEL, UNUSED, OC)                                                                              \
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"                                                                           \
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"   \
      "v_cndmask_b32_e32 %2, %6, %4, vcc\n\t"                                                                          \
      "v_cndmask_b32_e32 %0, %0, %3, vcc"                                                                              \
      : "=&v"(diff), OC(o), "=&v"(r)                                                                                   \
      : "v"(a.v), "v"(rmps), "v"((uint32_t)(ent >> 32)), "v"(lpss)                                                     \
      : "vcc")
  if (K & 2) {
    ISSCABAC_DEC_OUT("BYTE_2", "UNUSED_PRESERVE", "+v");
    st.store_at_hi(at, o);
  } else {
    ISSCABAC_DEC_OUT("BYTE_0", "UNUSED_PAD", "=&v");
    st.store_at(at, o);
  }
#undef ISSCABAC_DEC_OUT
  a.v = diff;
  a.R = r;
#else
  const bool mps_path = a.v < rmps;
  const uint32_t nx = (uint32_t)(ent >> (mps_path ? 40 : 32)) & 0xFFu;
  put_out_byte<K>(oa, ob, nx);
  st.store_at(at, nx);
  a.R = mps_path ? rmps : lpss;
  a.v = mps_path ? a.v : a.v - rmps;
#endif
}
// dec_bin_nx in absolute form, in two halves: the arithmetic on an entry already read (the latency-oriented instance issues
// the read of the next bin's entry right behind this bin's decision and runs everything else in its shadow) ...
// (sel = column + 4 of the LPS, sh = the comparison point: both follow from R alone -- abs_positions -- and are worked out
// while the entry is on its way)
CABAC_HD void abs_positions(uint32_t R, uint32_t &sel, uint32_t &sh) {
  const uint32_t t = 29u - clz32(R);  // sh + 6
  sel = R >> t;
  sh = t - 6u;
}
CABAC_HD uint32_t dec_bin_nx_abs_at(DecAbs &a, uint64_t ent, uint32_t sel, uint32_t sh) {
  const uint32_t lpss = lps_col((uint32_t)ent, sel) << sh;
  const uint32_t rmps = a.R - lpss;
  uint32_t nx;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t diff, r;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %6, %4, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %3, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r)
      : "v"(a.v), "v"(rmps), "v"((uint32_t)(ent >> 32)), "v"(lpss)
      : "vcc");
  a.v = diff;
  a.R = r;
#else
  const bool mps_path = a.v < rmps;
  nx = (uint32_t)(ent >> (mps_path ? 40 : 32)) & 0xFFu;
  a.R = mps_path ? rmps : lpss;
  a.v = mps_path ? a.v : a.v - rmps;
#endif
  return nx;
}
template <bool FIRST>
CABAC_HD uint32_t dec_bin_nx_abs_ent(DecAbs &a, uint64_t ent) {
  uint32_t sel = a.R >> 29, sh = 23u;
  if (!FIRST) abs_positions(a.R, sel, sh);
  return dec_bin_nx_abs_at(a, ent, sel, sh);
}
// ... and whole: lookup by the raw state byte, step, store
template <bool FIRST, class States, class Addr>
CABAC_HD uint32_t dec_bin_nx_abs(DecAbs &a, Addr at, uint32_t qb, States &st) {
  const uint32_t nx = dec_bin_nx_abs_ent<FIRST>(a, st.entry_raw(qb));
  st.store_at(at, nx);
  return nx;
}
// dec_bin_mixed_nx in absolute form: a bypass lane (mb = ~0) decides against R >> 1, which is its new R either way
template <bool FIRST, class States, class Addr>
CABAC_HD uint32_t dec_bin_mixed_nx_abs(DecAbs &a, uint32_t mb, Addr at, uint32_t qb, States &st) {
  const uint64_t ent = st.entry_raw(qb);
  // (a bypass lane: "LPS << sh" = R / 2, so that both paths leave R / 2 -- R is even: sh >= 1 -- and R - R / 2 is the value compared)
  const uint32_t rl = select_mask(mb, a.R >> 1, abs_lps<FIRST>((uint32_t)ent, a.R));
  const uint32_t cmpv = a.R - rl;
  const uint32_t hi = select_mask(mb, 0x0001u, (uint32_t)(ent >> 32));
  uint32_t nx;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t diff, r;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %6, %4, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %3, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r)
      : "v"(a.v), "v"(cmpv), "v"(hi), "v"(rl)
      : "vcc");
  a.v = diff;
  a.R = r;
#else
  const bool below = a.v < cmpv;
  nx = (hi >> (below ? 8 : 0)) & 0xFFu;
  a.R = below ? cmpv : rl;
  a.v = below ? a.v : a.v - cmpv;
#endif
  st.store_at(at, select_mask(mb, qb, nx));
  return nx;
}

// The mixed step on a dummy context (the decoder's twin of enc_bin_mixed2_acc; batches of <= 64 contexts): a bypass bin's state
// address is a byte of its own -- a dummy row of the wave's LDS -- that holds the raw state byte 254 / 255 (state 63: no
// context's state, refused at initialisation and unreachable by any transition), and the table's entries for those two bytes are
// the BYPASS entry: an all-zero rangeTabLps row (LPS term 0), successors 0xFE | bin decoded -- the dummy stays a bypass byte and
// bit 0 of what is stored is the output, like everywhere --, and all ones in the upper half, whose sign IS the lane mask "this is
// a bypass bin".  So the step has no select on the entry and none on the state store (dec_bin_mixed_nx_abs: three selects and a
// two-instruction mask per bin); what remains is the bypass lane's "LPS term" R / 2, which no table can hold (R = range << sh
// with an odd range): one shift and one and-or on the mask.  ref: CABAC_ArithmeticDecoder.cpp:288-331 (bypass), :87-190.
constexpr uint32_t kDecBypassRaw = 0xFEu;
constexpr uint32_t kDecBypassHi = 0xFFFF0000u | 0xFEu << 8 | 0xFFu;  // byte 0: value >= R / 2 -> bin 1; byte 1: below -> bin 0
// bytes of a ctx word that are neither a context below 64 nor the bypass marker, as bit 7 of their byte (the chunk then takes the
// general path: with the 5-bit group mask of the mixed2 step such a byte would address the wave's tiles)
CABAC_HD uint32_t not_ctx64_nor_bypass(uint32_t cw) {
  const uint32_t low7 = cw & 0x7F7F7F7Fu, u = low7 + 0x01010101u, s = low7 + 0x40404040u;
  return ((cw & ~u) | (~cw & s)) & 0x80808080u;  // bit 7 set: not 0xFF; bit 7 clear: >= 64
}
template <bool FIRST, class States, class Addr>
CABAC_HD uint32_t dec_bin_mixed2_nx_abs(DecAbs &a, Addr at, uint32_t qb, States &st) {
  const uint64_t ent = st.entry_raw(qb);
  const uint32_t hi = (uint32_t)(ent >> 32);
  const uint32_t mb = (uint32_t)((int32_t)hi >> 31);  // ~0 on a bypass lane
  const uint32_t rl = ((a.R >> 1) & mb) | abs_lps<FIRST>((uint32_t)ent, a.R);  // (a bypass entry's row is 0: its LPS term is 0)
  const uint32_t cmpv = a.R - rl;
  uint32_t nx;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t diff, r;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %1, %5, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "v_cndmask_b32_e32 %2, %6, %4, vcc\n\t"
      "v_cndmask_b32_e32 %0, %0, %3, vcc"
      : "=&v"(diff), "=&v"(nx), "=&v"(r)
      : "v"(a.v), "v"(cmpv), "v"(hi), "v"(rl)
      : "vcc");
  a.v = diff;
  a.R = r;
#else
  const bool below = a.v < cmpv;
  nx = (hi >> (below ? 8 : 0)) & 0xFFu;
  a.R = below ? cmpv : rl;
  a.v = below ? a.v : a.v - cmpv;
#endif
  st.store_at(at, nx);
  return nx;
}

// The decoder's select-based step for chunks that mix context-coded and bypass bins (see enc_bin_mixed_acc): a bypass
// bin compares the value with the whole range one bit further down (ref: CABAC_ArithmeticDecoder.cpp:288-331) and leaves
// range and state alone.  Returns the (mps, path) pair of dec_bin_at_mem; a bypass bin reports mps 0, which makes
// pairs_to_bins() turn its path bit into the bin.  Needs sh >= 8 (a bypass bin) / 7 (a context bin).
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_mixed(DecCore &d, uint32_t mb, Addr at, States &st) {
  const uint32_t q = st.load_at(at);
  const uint64_t ent = st.entry_mem(q);  // one lookup: row + successors by path, see dec_bin_row_q
  const uint32_t lps = lps_of((uint32_t)ent, d.range);
  const uint32_t rmps = d.range - lps;
  const int32_t sh = d.sh + (int32_t)mb;  // mb = -1: the bypass bin's bit
  const uint32_t scaled = select_mask(mb, d.range, rmps) << sh;
  uint32_t diff;
  const bool below = __builtin_sub_overflow(d.v, scaled, &diff);  // the MPS path / a bypass bin of 0
  const uint32_t idx = q + (uint32_t)below;
  const uint32_t hi = (uint32_t)(ent >> 32);
  st.store_at(at, select_mask(mb, q, below ? (hi >> 8) : hi));  // (the store takes the low byte)
  const uint32_t r = select_mask(mb, d.range, below ? rmps : lps);
  const uint32_t n = clz32(r) - 23u;
  d.v = below ? d.v : diff;
  d.range = r << n;
  d.sh = sh - (int32_t)n;
  return idx & ~(mb & 2u);
}

// Latency-oriented form (see enc_bin_lat_acc): q comes from the caller (loaded one bin ahead, the previous successor
// forwarded for a repeated context); the single lookup by q brings the row and both successors, so the decision needs no
// lookup behind it and the next bin's state is one select away.  nx receives the successor.
template <class States, class Addr>
CABAC_HD uint32_t dec_bin_lat(DecCore &d, Addr at, uint32_t q, States &st, uint32_t &nx) {
  return dec_bin_row_q(d, at, q, st, nx);
}
// collected (mps, path) pairs -> the collect_bin() format: bin = mps ^ !path at the odd bits, even bits 0
CABAC_HD uint32_t pairs_to_bins(uint32_t acc) { return ~(acc ^ (acc << 1)) & 0xAAAAAAAAu; }
// a bypass bin in pair form
CABAC_HD uint32_t pair_of_bin(uint32_t bin) { return (bin << 1) | 1u; }

template <class States>
CABAC_HD uint32_t dec_bin(DecCore &d, uint32_t ctx, States &st) {
  return (dec_bin_at_b1(d, st.addr(ctx), st) >> 1) & 1u;
}

template <class States>
CABAC_HD uint32_t dec_bin_any_b1(DecCore &d, uint32_t ctx, States &st) {  // bin at bit 1, bit 0 = 0
  const bool ep = ctx == kBypass;
  const uint32_t c = ep ? 0u : ctx;
  const uint32_t q = st.load(c);
  const uint32_t row = st.lps_row(q);
  const uint32_t nx = st.next_entry(q);
  const uint32_t lps = lps_of(row, d.range);
  const uint32_t rmps = d.range - lps;
  const int32_t sh = d.sh - (ep ? 1 : 0);
  const uint32_t scaled = (ep ? d.range : rmps) << sh;
  const uint32_t m = d.v >= scaled ? ~0u : 0u;  // LPS for a context bin, 1 for a bypass bin
  const uint32_t r = rmps ^ ((rmps ^ lps) & m);
  const uint32_t n = ep ? 0u : clz32(r) - 23u;
  d.v -= scaled & m;
  d.range = ep ? d.range : (r << n);
  d.sh = sh - (int32_t)n;
  const uint32_t qn = next_q_of(st, q, m, nx, 0);
  st.store(c, ep ? q : qn);
  return ep ? (m & 2u) : bin_word_b1(q, m);
}
template <class States>
CABAC_HD uint32_t dec_bin_any(DecCore &d, uint32_t ctx, States &st) {
  return (dec_bin_any_b1(d, ctx, st) >> 1) & 1u;
}

CABAC_HD uint32_t dec_bin_ep(DecCore &d) {  // ref: :288-331
  d.sh -= 1;
  const uint32_t scaled = d.range << d.sh;
  const bool one = d.v >= scaled;
  d.v -= one ? scaled : 0u;
  return one ? 1u : 0u;
}

// decodeBinProb for p != 50 (ref: CABAC_ArithmeticDecoder.cpp:192-286) in the look-ahead formulation; needs sh >= 7 and
// takes at most 6 bits, like a context bin
CABAC_HD uint32_t dec_bin_prob(DecCore &d, uint32_t p, uint32_t row) {
  const uint32_t mps = p > 50u ? 1u : 0u;
  const uint32_t lps = lps_of(row, d.range);
  const uint32_t rmps = d.range - lps;
  const uint32_t scaled = rmps << d.sh;
  uint32_t diff;
  const bool mps_path = __builtin_sub_overflow(d.v, scaled, &diff);
  const uint32_t r = mps_path ? rmps : lps;
  const uint32_t n = clz32(r) - 23u;
  d.v = mps_path ? d.v : diff;
  d.range = r << n;
  d.sh -= (int32_t)n;
  return mps_path ? mps : 1u - mps;
}

// decodeBinTrm + the two asserts of finish() as flags; len = payload length of this stream.
// last_byte_at(i) must return payload byte i (i < len).
// nbits = payload bits merged into v so far (8 x the bytes taken from a byte source).
template <class ByteAt>
CABAC_HD uint32_t dec_finish(DecCore &d, uint32_t nbits, uint32_t len, ByteAt byte_at) {
  uint32_t status = 0;
  d.range -= 2;
  // bits the reference has shifted in total; it has read 2 + floor(shifted / 8) bytes
  uint32_t shifted = nbits - 16u - (uint32_t)(d.sh - 7);
  if (!(d.v >= (d.range << d.sh))) {
    status |= 1u;                      // terminate bin != 1 ...
    if (d.range < 256u) shifted += 1;  // ... and the reference then renormalises by one bit (:452-463)
  }
  const uint32_t last_idx = 1u + (shifted >> 3);
  const uint32_t r = shifted & 7u;  // 8 + m_bitsNeeded
  const uint32_t last = last_idx < len ? byte_at(last_idx) : 0xFFu;
  if (((last << r) & 0xFFu) != 0x80u) status |= 2u;
  if (last_idx >= len || len < 2u) status |= 4u;
  return status;
}

// ------------------------------------------------------------------------------------------------
// The decoder in the reference's own formulation (32-bit m_uiValue, m_bitsNeeded, one byte at a time).
// DecCore above is bit-identical to it as long as value < range << 7, which the arithmetic itself maintains from
// start() on -- unless the stream's first byte is 0xFF (no encoder emits that): then value >= range << 7 from the
// first bin, every bin decodes as an LPS, the excess doubles with every renormalisation shift and the reference's
// 32-bit register wraps; DecCore, which keeps up to 16 look-ahead bits in the same 32 bits, would wrap earlier.  Streams
// that start with 0xFF (only corrupt input) are therefore decoded with this restatement, so that even their garbage and
// their status flags are the reference's.   ref: CABAC/CABAC_ArithmeticDecoder.cpp:54-71, 87-190, 288-331, 423-472, 73-85
// Rd policy: uint32_t operator()() -> next payload byte, 0xFF past the end.
// ------------------------------------------------------------------------------------------------
struct RefDec {
  uint32_t value, range;
  int32_t need;  // m_bitsNeeded
};
CABAC_HD bool needs_ref_decoder(uint32_t first_byte) { return first_byte == 0xFFu; }

// a stream's payload [start, end) clipped to the payload buffer: offsets handed over in device memory are not validated
// on the host, and no kernel may read outside the buffer whatever they say.  Returns the length, start is clipped too.
CABAC_HD uint32_t clip_stream(uint64_t &start, uint64_t end, uint64_t payload_bytes) {
  if (start > payload_bytes) start = payload_bytes;
  if (end > payload_bytes) end = payload_bytes;
  const uint64_t len = end > start ? end - start : 0;
  return len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len;
}

template <class Rd>
CABAC_HD void ref_dec_start(RefDec &r, Rd &rd) {  // ref: :54-71
  r.range = 510;
  r.need = -8;
  r.value = rd() << 8;
  r.value |= rd();
}
template <class States, class Rd>
CABAC_HD uint32_t ref_dec_bin(RefDec &r, uint32_t ctx, States &st, Rd &rd) {  // ref: :87-190
  const uint32_t q = st.load(ctx);
  const uint32_t lps = (st.lps_row(q) >> (8u * ((r.range >> 6) & 3u))) & 0xFFu;
  const uint32_t nx = st.next_entry(q);
  r.range -= lps;
  const uint32_t scaled = r.range << 7;
  if (r.value < scaled) {  // MPS path
    st.store(ctx, next_q_of(st, q, 0u, nx, 0));
    if (scaled < (256u << 7)) {
      r.range = scaled >> 6;
      r.value += r.value;
      if (++r.need == 0) {
        r.need = -8;
        r.value += rd();
      }
    }
    return (q >> 1) & 1u;
  }
  const uint32_t n = clz32(lps) - 23u;  // == sm_aucRenormTable[lps >> 3]
  r.value = (r.value - scaled) << n;
  r.range = lps << n;
  st.store(ctx, next_q_of(st, q, ~0u, nx, 0));
  r.need += (int32_t)n;
  if (r.need >= 0) {
    r.value += rd() << r.need;
    r.need -= 8;
  }
  return ((q >> 1) & 1u) ^ 1u;
}
template <class Rd>
CABAC_HD uint32_t ref_dec_bin_ep(RefDec &r, Rd &rd) {  // ref: :288-331
  r.value += r.value;
  if (++r.need >= 0) {
    r.need = -8;
    r.value += rd();
  }
  const uint32_t scaled = r.range << 7;
  if (r.value >= scaled) {
    r.value -= scaled;
    return 1u;
  }
  return 0u;
}
template <class Rd>
CABAC_HD uint32_t ref_dec_bin_prob(RefDec &r, uint32_t p, uint32_t row, Rd &rd) {  // ref: :192-286, p != 50
  const uint32_t mps = p > 50u ? 1u : 0u;
  const uint32_t lps = (row >> (8u * ((r.range >> 6) & 3u))) & 0xFFu;
  r.range -= lps;
  const uint32_t scaled = r.range << 7;
  if (r.value < scaled) {
    if (scaled < (256u << 7)) {
      r.range = scaled >> 6;
      r.value += r.value;
      if (++r.need == 0) {
        r.need = -8;
        r.value += rd();
      }
    }
    return mps;
  }
  const uint32_t n = clz32(lps) - 23u;
  r.value = (r.value - scaled) << n;
  r.range = lps << n;
  r.need += (int32_t)n;
  if (r.need >= 0) {
    r.value += rd() << r.need;
    r.need -= 8;
  }
  return 1u - mps;
}
// decodeBinTrm + the two asserts of finish() as flags.  nread = payload bytes handed out by rd so far (it keeps counting
// past the end), last = the byte the most recent rd() returned, len = payload length of the stream
template <class Rd>
CABAC_HD uint32_t ref_dec_finish(RefDec &r, Rd &rd, uint32_t len) {  // ref: :423-472, :73-85
  uint32_t status = 0;
  r.range -= 2;
  const uint32_t scaled = r.range << 7;
  if (!(r.value >= scaled)) {
    status |= 1u;
    if (scaled < (256u << 7)) {
      r.range = scaled >> 6;
      r.value += r.value;
      if (++r.need == 0) {
        r.need = -8;
        r.value += rd();
      }
    }
  }
  if (((rd.last << (8 + r.need)) & 0xFFu) != 0x80u) status |= 2u;
  if (rd.nread - 1u >= len || len < 2u) status |= 4u;
  return status;
}
// byte reader over a stream's payload for the functions above
struct RefReader {
  const uint8_t *p;
  uint32_t len, nread, last;
  CABAC_HD uint32_t operator()() {
    last = nread < len ? p[nread] : 0xFFu;
    nread++;
    return last;
  }
};

// ------------------------------------------------------------------------------------------------
// Bit source for the decoder: a 64-bit MSB-first window `cur` over the stream's payload that is re-based at
// wave-uniform points (boundary() / ensure()) from a 16-byte piece fetched one boundary earlier, so the HBM/L2 latency
// of the fetch is never on the bin loop's critical path.  Positions are in bits from the Mem policy's base.
// Mem policy: void load16(uint32_t off, uint64_t &lo, uint64_t &hi) -> payload bytes [a, a+16) of this
// stream, a = off rounded down to 4 (little-endian in lo, hi); returns nothing past the payload buffer.
// ------------------------------------------------------------------------------------------------
#ifndef ISSCABAC_PREFETCH_BITS
#define ISSCABAC_PREFETCH_BITS 32u  // bits into the piece in use at which the next one is requested
#endif
template <class Mem>
struct WindowSrc {
  Mem mem;
  uint32_t len;   // payload bytes of this stream; bytes at or past len read as 0xFF (fstream EOF)
  uint32_t bpos;  // bit position of the next unread bit
  uint64_t cur;   // the next bits, MSB first
  uint32_t cend;  // bit position just past the last bit cur holds
  // piece in use: payload bits [pbit, pbit + 128) as one big-endian number, EOF fill applied, complete (no load pending
  // on these registers): ph = its upper 64 bits, pl1 = its lower 64 bits >> 1 (pre-shifted for the funnel in rebase())
  uint64_t ph, pl1;
  uint32_t pbit;
  // piece in flight: bytes [npos, npos + 16) as loaded (little endian), written by the most recent prefetch()
  uint64_t nlo, nhi;
  uint32_t npos;

  CABAC_HD uint32_t byte_pos() const { return bpos >> 3; }
  CABAC_HD int32_t avail_bits() const { return (int32_t)(cend - bpos); }
  CABAC_HD void prefetch() {
    npos = byte_pos() & ~3u;
    mem.load16(npos, nlo, nhi);
  }
  // ~0 >> (8 * k) for k real bytes in front of the fill, k in 0..8
  static CABAC_HD uint64_t ff_after(uint32_t k) { return k == 0 ? ~0ull : (k >= 8 ? 0ull : (~0ull >> (8u * k))); }
  // slow path: cur = the payload from bpos on, fetched now (blocking)
  static CABAC_HD_RARE uint64_t fetch8(Mem m, uint32_t at) {
    uint64_t a, b;
    const uint32_t base = at & ~3u, d = at - base;
    m.load16(base, a, b);
    return d == 0 ? a : (a >> (8u * d)) | (b << (64u - 8u * d));
  }
  CABAC_HD void reload_blocking() {
    const uint32_t at = byte_pos();
    cur = __builtin_bswap64(fetch8(mem, at));
    if (at + 8u > len) cur |= ff_after(len > at ? len - at : 0u);
    cur <<= bpos & 7u;
    cend = (bpos & ~7u) + 64u;
  }
  // make the piece in flight the piece in use (waits for its load)
  CABAC_HD void adopt() { adopt_from(__builtin_bswap64(nlo), __builtin_bswap64(nhi)); }
  // ... from the piece's two halves as big-endian numbers (the decode kernel converts them as it takes them out of the
  // registers the load landed in)
  CABAC_HD void adopt_from(uint64_t h, uint64_t pl) {
    ph = h;
    if (CABAC_WAVE_ANY(npos + 16u > len)) {  // only near the stream's end
      const uint32_t k = len > npos ? len - npos : 0u;  // real bytes in the piece
      if (k < 16u) {
        ph |= ff_after(k);
        pl |= ff_after(k > 8u ? k - 8u : 0u);
      }
    }
    pl1 = pl >> 1;
    pbit = 8u * npos;
  }
  CABAC_HD void start(uint32_t first_byte) {
    bpos = 8u * first_byte;
    cend = bpos;
    cur = 0;
    ph = pl1 = 0;
    prefetch();
    pbit = 8u * npos;
  }
  // re-base cur on the piece in use: 64 bits from bpos on.  Touches no register with a load in flight.
  CABAC_HD void rebase() {
    const uint32_t b = bpos - pbit;
    if (b < 64u) {
      cur = (ph << b) | (pl1 >> (63u - b));
      cend = bpos + 64u;
    } else {
      reload_blocking();  // more than 8 bytes into the piece: rare
    }
  }
  // wave-uniform: re-base when some lane's window holds fewer than n bits
  CABAC_HD void ensure(int32_t n) {
    if (CABAC_WAVE_ANY_LT(avail_bits(), n)) {
      CABAC_RARE_BODY();
      if (avail_bits() < n) rebase();
    }
  }
  // every 16 bins: adopt the piece fetched one boundary ago (the only point that waits for a payload load), re-base,
  // and fetch a new piece once 4 bytes of the current one are used up -- on average every payload byte is requested
  // ~2x instead of ~10x (the re-reads were missing in L2)
  CABAC_HD void boundary() {
    adopt();
    rebase();
    if (bpos - pbit >= ISSCABAC_PREFETCH_BITS) prefetch();
  }
  // (v << g) | the next g bits, g in 0..23 (avail_bits() >= g is the caller's business): one 64-bit shift of {v, cur.hi}
  CABAC_HD uint32_t feed_nocheck(uint32_t v, uint32_t g) {
    const uint64_t pair = ((uint64_t)v << 32) | (cur >> 32);
    cur <<= g;
    bpos += g;
    return (uint32_t)((pair << g) >> 32);
  }
  CABAC_HD uint32_t feed(uint32_t v, uint32_t g) {
    if ((int32_t)g > avail_bits()) rebase();
    return feed_nocheck(v, g);
  }
  CABAC_HD uint32_t next() { return feed(0u, 8u); }
};

// ------------------------------------------------------------------------------------------------
// The same bit source on ALIGNED, NON-OVERLAPPING pieces (k_decode; round 6).  WindowSrc asks for a new piece at the window's own
// position as soon as four bytes of the piece in use are consumed: consecutive pieces overlap by up to 12 of their 16 bytes, a lane
// sends ~64 requests for its ~410 payload bytes -- 7 us apart, so nearly every one fills its 128-byte line anew: 8.6 GB of L2 fills
// for 0.43 GB of payload (profiles/r06/e_ab_memory_side.txt: without them k_decode is 5.4 % faster).  Here piece k is bytes
// [16 k, 16 k + 16) of the stream's buffer (Mem's base) and every byte is requested ONCE (~27 requests per lane).  The window
// steps over the payload in 8-byte words: w0, w1 = the two words the next 64 bits lie in (big-endian, EOF fill applied), the window
// always starts inside w0; when it has left w0, w1 moves down and the next word comes out of piece N -- its low half, then its high
// half, then N is requested anew (16 bytes further on), at least 8 bytes of consumption before its first word is needed.
// Same registers as WindowSrc (4 + 4), same interface.  pbit = bit position of w0's first bit (a multiple of 64): bit 6 of it says
// which half of N comes next.
// ------------------------------------------------------------------------------------------------
template <class Mem>
struct PieceSrc {
  Mem mem;
  uint32_t len;   // payload bytes of this stream; bytes at or past len read as 0xFF (fstream EOF)
  uint32_t bpos;  // bit position of the next unread bit
  uint64_t cur;   // the next bits, MSB first
  uint32_t cend;  // bit position just past the last bit cur holds
  uint64_t w0, w1;   // payload bits [pbit, pbit + 128)
  uint32_t pbit;
  uint64_t nlo, nhi; // piece N as loaded (little endian): the 16 bytes that hold the word behind w1; its load may be in flight

  CABAC_HD uint32_t byte_pos() const { return bpos >> 3; }
  CABAC_HD int32_t avail_bits() const { return (int32_t)(cend - bpos); }
  static CABAC_HD uint64_t ff_after(uint32_t k) { return k == 0 ? ~0ull : (k >= 8 ? 0ull : (~0ull >> (8u * k))); }
  // the 8 bytes at `at` as a big-endian number, bytes at or past len replaced by 0xFF
  CABAC_HD uint64_t filled(uint64_t be, uint32_t at) const {
    if (CABAC_WAVE_ANY(at + 8u > len)) be |= ff_after(len > at ? len - at : 0u);  // only near the stream's end
    return be;
  }
  // (re)start at byte `at`: w0, w1 = the aligned 16 bytes `at` lies in (fetched now: blocking), N = the 16 behind them
  CABAC_HD_RARE void restart(uint32_t at) {
    const uint32_t p0 = at & ~15u;
    uint64_t a, b;
    mem.load16(p0, a, b);
    w0 = filled(__builtin_bswap64(a), p0);
    w1 = filled(__builtin_bswap64(b), p0 + 8u);
    pbit = 8u * p0;
    mem.load16(p0 + 16u, nlo, nhi);
  }
  CABAC_HD void start(uint32_t first_byte) {
    bpos = 8u * first_byte;
    cend = bpos;
    cur = 0;
    restart(first_byte);
  }
  // the window has left w0: w1 moves down, the next word comes out of N (waits for N's load when its low half is taken)
  CABAC_HD void step_word() {
    const bool low = (pbit & 64u) == 0;      // w0, w1 are an aligned pair: the next word is N's low half
    const uint32_t at = (pbit >> 3) + 16u;    // its first byte
    w0 = w1;
    w1 = filled(__builtin_bswap64(low ? nlo : nhi), at);
    pbit += 64u;
    if (!low) mem.load16(at + 8u, nlo, nhi);  // N is used up: the 16 bytes behind it
  }
  // cur = 64 bits from bpos on.  The window started inside w0 when it was last re-based and holds 64 bits, so it starts inside w0 or
  // w1 now (bpos - pbit < 128): at most one step
  CABAC_HD void rebase() {
    uint32_t b = bpos - pbit;
    if (CABAC_WAVE_ANY(b >= 64u)) {
      if (b >= 64u) {
        step_word();
        b -= 64u;
      }
    }
    cur = (w0 << b) | ((w1 >> 1) >> (63u - b));
    cend = bpos + 64u;
  }
  CABAC_HD void ensure(int32_t n) {  // wave-uniform: re-base when some lane's window holds fewer than n bits
    if (CABAC_WAVE_ANY_LT(avail_bits(), n)) {
      CABAC_RARE_BODY();
      if (avail_bits() < n) rebase();
    }
  }
  CABAC_HD void boundary() { rebase(); }  // every 16 bins
  CABAC_HD uint32_t feed_nocheck(uint32_t v, uint32_t g) {
    const uint64_t pair = ((uint64_t)v << 32) | (cur >> 32);
    cur <<= g;
    bpos += g;
    return (uint32_t)((pair << g) >> 32);
  }
  CABAC_HD uint32_t feed(uint32_t v, uint32_t g) {
    if ((int32_t)g > avail_bits()) rebase();
    return feed_nocheck(v, g);
  }
  CABAC_HD uint32_t next() { return feed(0u, 8u); }
};

}  // namespace isscabac
#endif
