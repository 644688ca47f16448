// cabac_pair.hip -- the encode kernel for SMALL launches (at most two waves of streams per SIMD): a producer / consumer
// PAIR of wavefronts per 64 streams, written for gfx950 (MI355X, wave64).
//
// Why a pair.  With one wave per 64 streams (cabac_kernels.hip) a launch of <= 128 Ki streams leaves two of a SIMD's four
// wave slots and half of the LDS empty, and a wave by itself issues no faster than one instruction per ~2.2 ns whatever
// the other wave does: the ~29 instructions of a bin then ARE the time.  In the encoder the context states evolve from the
// bins and contexts alone -- no range, no low (ref: CABAC/CABAC_ArithmeticEncoder.cpp:113-178 reads the state, never the
// other way round; ContextModel.cpp:107-123) -- so the per-bin work splits into two recurrences that never feed back:
//   wave A (producer): input tiles, context-state pass (state -> table entry -> successor), per bin ONE record for B:
//                      the state's 4-byte rangeTabLps row and one bit "this bin is not the state's mps";
//   wave B (consumer): the range / low arithmetic on those records (pure VALU: no table, no state image, no input
//                      arrays), the 16-bit digit flush and the byte sink.
// Records travel through an LDS ring of two slots of 16 bins (one chunk); the two waves of a workgroup meet at one
// s_barrier per chunk: A fills slot c & 1 while B codes slot (c - 1) & 1.  Each wave's instruction stream is about half of
// the single wave's, both SIMD slots and the idle LDS are used, and neither chain contains the other's latency.
//
// A chunk is PLAIN (every lane has 16 context-coded bins: straight-line code in both waves), MIXED (context-coded and
// bypass bins only, config 5's chunks: the same state pass with a bypass bin on a dummy context and the all-zero table
// entry -- cabac_lane.h: enc_bin_mixed2_acc --, two more bits per record, one select-free step in B) or GENERAL
// (fixed-probability bins, streams that end inside it: A classifies every bin under the lane's own exec mask and B runs
// a step whose record can say "nothing", "bypass bin b", or "row + flag").  Bytes are identical to
// k_encode's for every input (the GPU parity tests run every case through both).
//
// LDS of a workgroup (19 712 bytes, eight workgroups per CU = 128 Ki streams resident):
//   [0, 1024)       the 8-byte entries by packed state (cabac_lane.h: enc_entry_hi); the last 16 bytes -- state 63, which
//                   no context ever holds -- carry the two slots' kind words
//   [1024, 5376)    context states, one byte per (lane, context), cabac_wave.h: LdsStates, + the bypass bins' dummy context row
//   [5376, 9472)    the two input half tiles (bins, ctx)
//   [9472, 10496)   B's output staging, 16 bytes per lane
//   [10496, 19712)  the ring: per slot 4 x 1 KiB of rows (word g of lane l at g * 1024 + 16 l: both the ds_write_b128 of A
//                   and the ds_read_b128 of B are conflict-free) + 512 bytes of flag bits (8 bytes per lane)
//   [19712, 19744)  the words through which the FOUR waves a workgroup starts with tell each other their SIMD; two of them stay,
//                   on the SIMDs a per-CU ticket names (cabac_wave.h: pinned_wave_index), the other two leave at once
#include "cabac_wave.h"

namespace isscabac {


namespace pairk {
constexpr uint32_t kKind = 1008;
constexpr uint32_t kStates = 1024;
constexpr uint32_t kDummyRow = kStates + 4096;  // the bypass bins' "context" 64 in mixed chunks (cabac_lane.h: enc_bin_mixed2_acc)
constexpr uint32_t kTileB = kDummyRow + 256, kTileC = kTileB + 2048, kStage = kTileC + 2048;
constexpr uint32_t kRing = kStage + 1024;
constexpr uint32_t kSlotRows = 4096, kSlotBytes = kSlotRows + 512;
constexpr uint32_t kLdsBytes = kRing + 2 * kSlotBytes;
static_assert(kLdsBytes == 19712 && 8 * kLdsBytes <= 160 * 1024, "eight workgroups per CU");
constexpr uint32_t kPlain = 0, kGeneral = 1, kMixed = 2;
constexpr uint32_t kRecNone = kEncRecNone, kRecBypass = kEncRecBypass;  // (cabac_lane.h: the records and B's steps, shared with the host harness)

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
typedef __attribute__((address_space(3))) u32x4 lds_v4;

// the 8-byte entry of the doubled state in the LOW BYTE of q (the bytes above may hold anything): [row | enc_entry_hi]
__device__ __forceinline__ uint64_t entry_of(uint32_t q, uint32_t two) {
  uint32_t a;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(two), "v"(q));
  return *reinterpret_cast<const lds_u64 *>((uintptr_t)a);
}
// bit (16 + (i & 7)) of the flag word of bin i: the entry's flag byte (0x00 / 0xFF at bits 16..23 of t) masked to one bit
__device__ __forceinline__ uint32_t flag_bit(int i) { return 1u << (16 + (i & 7)); }
__device__ __forceinline__ void put_rows(uint32_t at, const uint4 &v) {
  const u32x4 w = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<lds_v4 *>((uintptr_t)at) = w;
}
__device__ __forceinline__ uint4 get_rows(uint32_t at) {
  const u32x4 w = *reinterpret_cast<const lds_v4 *>((uintptr_t)at);
  return make_uint4(w.x, w.y, w.z, w.w);
}
// Which two of a workgroup's four waves are the pair: cabac_wave.h, pinned_wave_index (KEEP 2).  The wave's role (0: A, 1: B) or -1: leave.
// `ids`: 32 bytes of LDS behind the kernel's own (inside its last allocation granule).
__device__ uint32_t g_pair_ticket[kPinTickets];
__device__ __forceinline__ int pair_pin_role(uint32_t *ids, int tid) { return pinned_wave_index(ids, g_pair_ticket, tid, 2, false); }
}  // namespace pairk

template <int K>
__device__ __forceinline__ uint32_t pair_next_of(uint32_t w8, uint32_t hi) { return BinOfWord<K>{0u, 0u, w8}.next_of(hi); }
__device__ __forceinline__ uint32_t pair_next_of_k(int k, uint32_t w8, uint32_t hi) {
  switch (k & 3) {
    case 0: return pair_next_of<0>(w8, hi);
    case 1: return pair_next_of<1>(w8, hi);
    case 2: return pair_next_of<2>(w8, hi);
    default: return pair_next_of<3>(w8, hi);
  }
}


// The producer's state pass over one chunk of 16 bins for every lane (PLAIN, or MIXED: context-coded and bypass bins).  The
// lookup of bin k goes out as soon as the successor of bin k - 1 is back (it is the state of bin k when both name the same
// context: forwarded in a register); the state of bin k + 1 is read one bin ahead, behind the store of the successor of
// bin k - 1.  Leaves the rows in the slot (word g of the lane at rows_at + 1024 g) and returns the flag words: bit 16 + i
// of f0 (bins 0..7) / f1 (bins 8..15) = "bin i is not its state's mps"; MIXED adds bit i = "bypass bin" and bit 8 + i =
// "bypass bin of value 1".  A bypass bin reads and writes the dummy context behind the state image and looks up the
// all-zero entry 127 (its state byte or-ed with 0xFE): row 0, successor 0, flag 0.
template <bool MIXED>
__device__ __forceinline__ void pair_state_pass(const LdsStates &stx, const uint4 &wb, const uint4 &wc, uint32_t rows_at, uint32_t two,
                                                uint32_t &f0, uint32_t &f1) {
  using namespace pairk;
  constexpr uint32_t kIn = 0x3F3F3F3Fu, kDummy = 64u * 0x01010101u, kGather = 0x01020408u;
  uint32_t mbfe = 0;
  auto quad_of = [&](int g) {
    const uint32_t cw = word_of(wc, g);
    if (!MIXED) return stx.prep4(cw);
    const uint32_t cwm = (cw >> 7) & 0x01010101u, wbm = word_of(wb, g) & cwm;
    mbfe = cwm * 0xFEu;
    // bytes of 0 / 1 -> four bits, bin 4 g + k at bit k
    const uint32_t kb = (cwm * kGather) >> 24, vb = (wbm * kGather) >> 24;
    uint32_t &f = g < 2 ? f0 : f1;
    f |= (kb | vb << 8) << (4 * (g & 1));
    return stx.prep4(select_mask(mbfe | cwm, kDummy, cw & kIn));
  };
  auto or_mbfe = [&](int k, uint32_t q) {
    if (!MIXED) return q;
    switch (k & 3) {
      case 0: return BinOfWord<0>{0u, 0u, 0u}.or_byte(q, mbfe);
      case 1: return BinOfWord<1>{0u, 0u, 0u}.or_byte(q, mbfe);
      case 2: return BinOfWord<2>{0u, 0u, 0u}.or_byte(q, mbfe);
      default: return BinOfWord<3>{0u, 0u, 0u}.or_byte(q, mbfe);
    }
  };
  uint4 rw;
  CtxQuad z = quad_of(0);
  uint32_t at = stx.addr32_of(z, 0), at_prev = 0;
  uint32_t qld = stx.load_at(at);
  uint64_t same = 0, ent_prev = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    uint32_t q = qld, tk = 0;
    if (k > 0) {
      tk = pair_next_of_k(k - 1, word_of(wb, (k - 1) >> 2) << 3, (uint32_t)(ent_prev >> 32));
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(q) : "v"(qld), "v"(tk), "s"(same));
    }
    const uint64_t ent = entry_of(or_mbfe(k, q), two);
    if (k > 0) {
      stx.store_at(at_prev, tk);
      if (k - 1 < 8) f0 |= tk & flag_bit(k - 1);
      else f1 |= tk & flag_bit(k - 1);
      set_word(rw, (k - 1) & 3, (uint32_t)ent_prev);
      if (((k - 1) & 3) == 3) put_rows(rows_at + ((k - 1) >> 2) * 1024u, rw);
    }
    uint32_t at_next = 0, qld_next = 0;
    uint64_t same_next = 0;
    if (k < 15) {
      if (((k + 1) & 3) == 0) z = quad_of((k + 1) >> 2);
      at_next = stx.addr32_of(z, (k + 1) & 3);
      qld_next = stx.load_at(at_next);
      asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(same_next) : "v"(at_next), "v"(at));
    }
    ent_prev = ent;
    at_prev = at;
    at = at_next;
    qld = qld_next;
    same = same_next;
  }
  const uint32_t tk = pair_next_of<3>(wb.w << 3, (uint32_t)(ent_prev >> 32));
  stx.store_at(at_prev, tk);
  f1 |= tk & flag_bit(15);
  rw.w = (uint32_t)ent_prev;
  put_rows(rows_at + 3 * 1024u, rw);
}

__global__ void __launch_bounds__(256, 4) k_encode_pair(EncArgs a) {
  using namespace pairk;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 128) {
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    const int ps = tid;  // 128 packed states
    const uint32_t sb = c_next.eb[ps >> 1] >> (16 * (ps & 1));
    tab[2 * ps] = ps >= 126 ? 0u : c_lps_rows[ps >> 1];
    tab[2 * ps + 1] = ps >= 126 ? 0u : enc_entry_hi(ps, sb & 0xFFu, (sb >> 8) & 0xFFu);
  }
  const int role = pair_pin_role(reinterpret_cast<uint32_t *>(smem + kLdsBytes), tid);  // (its barrier is the tables' too)
  if (role < 0) return;
  const int wib = __builtin_amdgcn_readfirstlane(role);  // 0: producer (A), 1: consumer (B)
  const uint64_t s0 = (uint64_t)blockIdx.x * 64;
  if (s0 >= a.S) return;

  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  // wave-uniform, and the same in both waves: the number of chunks is the number of barriers
  const uint32_t nmax = __builtin_amdgcn_readfirstlane(wave_max_u32(n)), nmin = __builtin_amdgcn_readfirstlane(wave_min_u32(n));
  const uint32_t nchunks = (nmax + 15u) >> 4;
  const uint32_t lane_rows = kRing + (uint32_t)lane * 16u, lane_flags = kRing + kSlotRows + (uint32_t)lane * 8u;

  if (wib == 0) {
    // ------------------------------------------------------------------------------------------ producer
    uint8_t *wbase = smem + kStates;
    {
      uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
      for (uint32_t g = 0; g < 16; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;  // (see k_encode: the whole image)
      st32[16 * 64] = 0;  // the dummy context: only ever receives the all-zero entry's successor
    }
    LdsStates stx{wbase + lane * 4};
    __builtin_amdgcn_s_setprio(1);
    uint8_t *tile_b = smem + kTileB, *tile_c = smem + kTileC;
    using BTile = TilePairT<4>;
    using CTile = BTile;
    auto tile_sync = [] { wave_lds_sync(); };
    const uint32_t nrounds = (nmax + BTile::T - 1) / BTile::T;
    const uint32_t nrows = (uint32_t)min((uint64_t)64, a.S - s0);
    const uint8_t *bins_w = a.bins + s0 * a.pitch, *ctx_w = a.ctx + s0 * a.pitch;
    uint4 pb[BTile::NLD], pc[CTile::NLD];
    if (nrounds) {
      BTile::load(pb, bins_w, nrows, a.pitch, 0, lane);
      CTile::load(pc, ctx_w, nrows, a.pitch, 0, lane);
    }
    tile_sync();
    const uint32_t two = opaque(2u);
    uint32_t c = 0;
    for (uint32_t t = 0; t < nrounds; t++) {
#pragma unroll
      for (int h = 0; h < BTile::NP; h++) {
        const uint32_t pos = t * BTile::T + h * 32;
        if (pos >= nmax) break;
        BTile::half_to_lds(tile_b, pb, h, lane);
        CTile::half_to_lds(tile_c, pc, h, lane);
        tile_sync();
        if (h == BTile::NP - 1 && pos + 32 < nmax) {  // the registers are free again: the next round
          BTile::load(pb, bins_w, nrows, a.pitch, pos + 32, lane);
          CTile::load(pc, ctx_w, nrows, a.pitch, pos + 32, lane);
        }
        for (int j = 0; j < 2; j++) {
          const uint32_t i0 = pos + j * 16;
          if (i0 >= nmax) break;
          const uint4 wb = BTile::row_chunk(tile_b, lane, j);
          const uint4 wc = CTile::row_chunk(tile_c, lane, j);
          const bool full = i0 + 16 <= nmin;
          const bool hi = chunk_not_plain<1>(wc, 0u);
          const uint32_t slot = (c & 1u) * kSlotBytes;
          const uint32_t rows_at = lane_rows + slot, flags_at = lane_flags + slot;
          uint32_t f0 = 0, f1 = 0;
          [[maybe_unused]] uint4 rw;
          if (full && !hi) {
            pair_state_pass<false>(stx, wb, wc, rows_at, two, f0, f1);
            *reinterpret_cast<lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))) = kPlain;
          } else if (full && !wave_any((odd_high_bytes(wc.x) | odd_high_bytes(wc.y) | odd_high_bytes(wc.z) | odd_high_bytes(wc.w)) != 0)) {
            pair_state_pass<true>(stx, wb, wc, rows_at, two, f0, f1);
            *reinterpret_cast<lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))) = kMixed;
          } else {
            // bypass bins, fixed-probability bins, ragged ends: every lane classifies its own bin (cabac_kernels.hip: the
            // general path of k_encode) and leaves a record B's uniform step understands
#pragma unroll
            for (int k = 0; k < 16; k++) {
              uint32_t row = kRecNone, fl = 0;
              if (full || i0 + k < n) {
                const uint32_t cb = byte_of(wc, k), bin = (word_of(wb, k >> 2) >> (8 * (k & 3))) & 1u;
                if (is_ctx_byte<1>(cb, 64u)) {
                  const uint32_t ak = stx.base32() + (((cb & ~3u) << 6) | (cb & 3u));
                  const uint64_t ent = entry_of(stx.load_at(ak), two);
                  const uint32_t tk = (uint32_t)(ent >> 32) >> (8u * bin);
                  stx.store_at(ak, tk);
                  row = (uint32_t)ent;
                  fl = tk;
                } else if (cb == kBypass || cb == kProbBase + 50u) {
                  row = kRecBypass | bin << 1;
                } else if (is_prob_ctx_byte<1>(cb) && prob_valid(cb - kProbBase)) {
                  row = c_prob_rows[prob_row_index(cb - kProbBase)];
                  fl = bin != (cb - kProbBase > 50u ? 1u : 0u) ? 0x00FF0000u : 0u;
                }
              }
              if (k < 8) f0 |= fl & flag_bit(k);
              else f1 |= fl & flag_bit(k);
              set_word(rw, k & 3, row);
              if ((k & 3) == 3) put_rows(rows_at + (k >> 2) * 1024u, rw);
            }
            *reinterpret_cast<lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))) = kGeneral;
          }
          *reinterpret_cast<lds_u64 *>((uintptr_t)flags_at) = (uint64_t)f0 | (uint64_t)f1 << 32;
          __syncthreads();  // chunk c is in its slot; B has left slot (c + 1) & 1
          c++;
        }
        tile_sync();
      }
    }
    __syncthreads();  // (B's last chunk)
    return;
  }

  // -------------------------------------------------------------------------------------------- consumer
  EncCore e;
  enc_start(e);
  SlotSink sink;
  sink.stage = smem + kStage + lane * 16;
  sink.slots = a.slots + s0 * a.slot_stride;
  sink.mine = (uint32_t)(s - s0) * (uint32_t)a.slot_stride;
  sink.pos = 0;
  sink.on = valid;
  __syncthreads();  // chunk 0 is there
  for (uint32_t c = 0; c < nchunks; c++) {
    const uint32_t slot = (c & 1u) * kSlotBytes;
    const uint32_t rows_at = lane_rows + slot, flags_at = lane_flags + slot;
    const uint32_t kind = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))));
    const uint64_t fw = *reinterpret_cast<const lds_u64 *>((uintptr_t)flags_at);
    const uint32_t f0 = (uint32_t)fw, f1 = (uint32_t)(fw >> 32);
    if (kind == kPlain) {
      EncWordAcc acc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint4 rw = get_rows(rows_at + g * 1024u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = 4 * g + k;
          const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(i < 8 ? f0 : f1, 16 + (i & 7), 1);  // ~0: the bin is not its state's mps
          enc_rec_plain(e, acc, k == 0, word_of(rw, k), m);
        }
        enc_word_close(e, acc);
        if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
      }
    } else if (kind == kMixed) {
      // context-coded and bypass bins: one step for both (cabac_lane.h: enc_bin_mixed_acc) -- a bypass bin's record has row 0
      // and flag 0, so its range passes through untouched (shift 0), bit i of the flag word adds its one bit of shift and
      // bit 8 + i says whether `low` gains the range (ref: CABAC/CABAC_ArithmeticEncoder.cpp:250-270)
      EncWordAcc acc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint4 rw = get_rows(rows_at + g * 1024u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = 4 * g + k;
          const uint32_t f = i < 8 ? f0 : f1;
          enc_rec_mixed(e, acc, k == 0, word_of(rw, k), (uint32_t)__builtin_amdgcn_sbfe(f, 16 + (i & 7), 1), __builtin_amdgcn_ubfe(f, i & 7, 1),
                        (uint32_t)__builtin_amdgcn_sbfe(f, 8 + (i & 7), 1));
        }
        enc_word_close(e, acc);
        if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint4 rw = get_rows(rows_at + g * 1024u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = 4 * g + k;
          enc_rec_general(e, word_of(rw, k), (uint32_t)__builtin_amdgcn_sbfe(i < 8 ? f0 : f1, 16 + (i & 7), 1));
        }
        if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
      }
    }
    __syncthreads();  // slot c & 1 is free again; chunk c + 1 is there
  }
  enc_flush16(e, sink);
  enc_finish16(e, sink);
  sink.close();
  if (valid) a.lens[s] = sink.bytes();
}

bool encode_pair_enabled() { return true; }

hipError_t run_encode_pair(const EncArgs &a, hipStream_t stream) {
  static int static_lds = -1;  // absolute LDS addresses: the dynamic LDS must start at 0 (cabac_kernels.hip: run_encode)
  if (static_lds < 0) {
    hipFuncAttributes fa;
    const hipError_t err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_encode_pair));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  const uint64_t nblk = (a.S + 63) / 64;
  // four waves per workgroup, two of which stay (pair_pin_role)
  hipLaunchKernelGGL(k_encode_pair, dim3((unsigned)nblk), dim3(256), pairk::kLdsBytes + 32, stream, a);
  return hipGetLastError();
}


// ================================================================================================
// The decode kernel for small launches: a decoder wave and a HELPER wave per 64 streams.
//
// The decoder's per-bin chain cannot be split -- the state of bin k + 1 follows from the decision of bin k (ref:
// CABAC/CABAC_ArithmeticDecoder.cpp:87-190) -- but a third of the single wave's instructions are not on it: the ctx tile
// (global loads, the transposition through LDS), the state addresses of the chunk's 16 bins, the chunk classification, and
// the output path (unpacking 128 bins of 64 rows and storing them as full 128-byte row segments).  At one or two waves per
// SIMD a wave issues one instruction per ~2.2 ns whatever its neighbour does, so every instruction the decoding wave does
// not have to issue is time.  Wave A (helper) prepares, per chunk, the 16 state addresses of every lane (16-bit LDS
// addresses, two to a dword), the ctx bytes and the chunk's kind in a ring slot, and later stores the bins wave B has
// left in the (double-buffered) output staging; wave B (decoder) runs k_decode's latency-oriented step on them.  One
// s_barrier per chunk, as in k_encode_pair.  Bins and status words are identical to k_decode's for every input.
//
// LDS of a workgroup (18 448 bytes): [0, 2048) the raw-byte entries | 4 KiB context states | 2 KiB ctx half tile (A) |
// 2 x 2 KiB output staging (OutBuf) | 2 slots x 3 KiB: addresses of bins 0..7, of bins 8..15, the 16 ctx bytes | kinds
// ================================================================================================

namespace dpair {
constexpr uint32_t kStates = 2048;
constexpr uint32_t kTile = kStates + 4096;
constexpr uint32_t kObuf = kTile + 2048;
constexpr uint32_t kRing = kObuf + 2 * OutBuf::BYTES;
constexpr uint32_t kSlotBytes = 3072;
constexpr uint32_t kKind = kRing + 2 * kSlotBytes;
constexpr uint32_t kLdsBytes = kKind + 16;
static_assert(8 * kLdsBytes <= 160 * 1024, "eight workgroups per CU");
constexpr uint32_t kPlain = 0, kMixed = 1, kGeneral = 2;
}  // namespace dpair

__global__ void __launch_bounds__(256, 4) k_decode_pair(DecArgs a) {
  using namespace dpair;
  using pairk::lds_u32;
  using pairk::get_rows;
  using pairk::put_rows;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  {
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    for (int qb = tid; qb < 256; qb += 256) {  // raw state byte qb: packed state qb >> 1, bit 0 is a stale bin (k_decode)
      const uint32_t ps = (uint32_t)qb >> 1;
      const uint32_t sp = c_next.ep[ps >> 1] >> (16 * (ps & 1));
      tab[2 * qb] = c_lps_rows[ps >> 1];
      tab[2 * qb + 1] = dec_entry_hi_out(ps, sp & 0xFFu, (sp >> 8) & 0xFFu);
    }
  }
  // four waves have started, two stay: on the SIMDs the CU's ticket names (pair_pin_role; its barrier is the tables' too)
  const int role = pairk::pair_pin_role(reinterpret_cast<uint32_t *>(smem + kLdsBytes), tid);
  if (role < 0) return;
  const int wib = __builtin_amdgcn_readfirstlane(role);  // 0: helper (A), 1: decoder (B)
  const uint64_t s0 = (uint64_t)blockIdx.x * 64;
  if (s0 >= a.S) return;
  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  const uint32_t nmin = __builtin_amdgcn_readfirstlane(wave_min_u32(n)), nmax = __builtin_amdgcn_readfirstlane(wave_max_u32(n));
  // every chunk of the rows is written -- zeros past the wave's longest stream --, like k_decode; one barrier per chunk
  const uint32_t nchunks = (a.nbins + 15u) >> 4;
  const uint32_t nrows = (uint32_t)min((uint64_t)64, a.S - s0);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  LdsStates stx{smem + kStates + lane * 4};

  if (wib == 0) {
    // -------------------------------------------------------------------------------------------- helper
    using DTile = TileT<4>;
    uint8_t *tile_c = smem + kTile;
    const uint8_t *ctx_w = a.ctx + s0 * a.pitch;
    uint8_t *out_w = a.bins_out + s0 * a.pitch;
    const uint32_t base2 = stx.base32() * 0x00010001u;  // the lane's state base in both halves (LDS addresses stay below 2^16)
    // 128 bins of every stream out of staging buffer blk & 1 as full 128-byte row segments (k_decode: store_out); chunks from
    // `nvalid` on were never written by B -- the rows end inside the block -- and read as zeros
    auto store_block = [&](uint32_t blk, uint32_t nvalid) {
      uint32_t *obuf = reinterpret_cast<uint32_t *>(smem + kObuf + (blk & 1u) * OutBuf::BYTES);
      for (uint32_t cz = nvalid; cz < 8; cz++) obuf[OutBuf::idx(cz, lane)] = 0;
      wave_lds_sync();
      const uint32_t off = blk * 128u;
      if (nrows == 64 && off + 128 <= a.pitch) {
        const uint32_t ln = opaque((uint32_t)lane);
        const uint32_t st_voff = (ln >> 3) * a.pitch + (ln & 7) * 16;
        const uint32_t st_lds = (uint32_t)(uintptr_t)(LdsStates::lds_u8 *)reinterpret_cast<uint8_t *>(obuf) + 4 * OutBuf::idx(ln & 7, ln >> 3);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint4 v = OutBuf::unpack(*reinterpret_cast<const lds_u32 *>((uintptr_t)(st_lds ^ (32u * k))));
          const u32x4 w = {v.x, v.y, v.z, v.w};
          uint8_t *base = out_w + ((uint64_t)(8 * k) * a.pitch + off);
          // (s_nop 1: a store of more than 64 bits must not be followed within two wait states by a VALU write of its data)
          if (kDecNTst) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(st_voff), "v"(w), "s"(base) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(st_voff), "v"(w), "s"(base) : "memory");
        }
        return;
      }
#pragma unroll 1
      for (int k = 0; k < 8; k++) {
        const uint32_t row = k * 8 + (lane >> 3), cc = lane & 7;
        const uint4 v = OutBuf::unpack(obuf[OutBuf::idx(cc, row)]);
        const uint32_t boff = off + cc * 16;
        if (row < nrows && boff < a.pitch) {
          uint8_t *p = out_w + (row * a.pitch + boff);
          if (kDecNT) nt_store16(p, v);
          else *reinterpret_cast<uint4 *>(p) = v;
        }
      }
    };
    const uint32_t nrounds = (a.nbins + DTile::T - 1) / DTile::T;
    uint4 pc[DTile::NLD];
    if (nrounds && nmax) DTile::template load<kDecNT>(pc, ctx_w, nrows, a.pitch, 0, lane);
    wave_lds_sync();
    uint32_t c = 0;
    for (uint32_t t = 0; t < nrounds; t++) {
#pragma unroll
      for (int h = 0; h < DTile::NP; h++) {
        const uint32_t pos = t * DTile::T + h * 32;
        if (pos >= a.nbins) break;
        const bool live_part = pos < nmax;
        if (live_part) {
          DTile::half_to_lds(tile_c, pc, h, lane);
          wave_lds_sync();
          if (h == DTile::NP - 1 && (t + 1) * DTile::T < nmax) DTile::template load<kDecNT>(pc, ctx_w, nrows, a.pitch, (t + 1) * DTile::T, lane);
        }
        for (int j = 0; j < 2; j++) {
          const uint32_t i0 = pos + j * 16;
          if (i0 >= a.nbins) break;
          if (i0 < nmax) {
            const uint4 wc = DTile::row_chunk(tile_c, lane, j);
            const bool full = i0 + 16 <= nmin;
            const bool hi = chunk_not_plain<1>(wc, 0u);
            const uint32_t slot = kRing + (c & 1u) * kSlotBytes + lane16;
            uint32_t kind = kGeneral;
            if (full && !hi) {
              kind = kPlain;
              uint4 p0, p1;  // 16-bit state addresses, bin 2 i in the low half of dword i
              CtxQuad z = stx.prep4(wc.x);
              p0.x = z.z01 + base2;
              p0.y = z.z23 + base2;
              z = stx.prep4(wc.y);
              p0.z = z.z01 + base2;
              p0.w = z.z23 + base2;
              z = stx.prep4(wc.z);
              p1.x = z.z01 + base2;
              p1.y = z.z23 + base2;
              z = stx.prep4(wc.w);
              p1.z = z.z01 + base2;
              p1.w = z.z23 + base2;
              put_rows(slot, p0);
              put_rows(slot + 1024u, p1);
            } else {
              if (full && !wave_any((odd_high_bytes(wc.x) | odd_high_bytes(wc.y) | odd_high_bytes(wc.z) | odd_high_bytes(wc.w)) != 0)) kind = kMixed;
              put_rows(slot + 2048u, wc);
            }
            *reinterpret_cast<lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))) = kind;
          }
          __syncthreads();  // chunk c is in its slot; B has finished chunk c - 1
          if (c >= 1 && ((c - 1) & 7u) == 7u) store_block((c - 1) >> 3, 8);
          c++;
        }
        if (live_part) wave_lds_sync();
      }
    }
    __syncthreads();  // B has finished the last chunk
    if (nchunks) store_block((nchunks - 1) >> 3, ((nchunks - 1) & 7u) + 1u);
    return;
  }

  // ---------------------------------------------------------------------------------------------- decoder
  {
    uint32_t *st32 = reinterpret_cast<uint32_t *>(smem + kStates) + lane;
    for (uint32_t g = 0; g < 16; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;  // (see k_decode: the whole image)
  }
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  const uint64_t abase = start & ~3ull;
  const uint32_t skew = (uint32_t)(start - abase);
  WindowSrc<DevMem> src;
  src.mem.payload = a.payload;
  src.mem.total = a.payload_bytes;
  src.mem.abase = abase;
  src.mem.lim = a.payload_bytes >= abase + 16 ? (int32_t)min(a.payload_bytes - 16 - abase, (uint64_t)0x7FFFFFFF) : -1;
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  if (valid && needs_ref_decoder(d.v >> 24) && a.exact_flag) *a.exact_flag = 1u;  // (k_decode_exact takes those streams)
  DecAbs da = dec_to_abs(d);
  __syncthreads();  // chunk 0 is there
  for (uint32_t c = 0; c < nchunks; c++) {
    const uint32_t i0 = c * 16u, cidx = c & 7u;
    uint32_t *obuf = reinterpret_cast<uint32_t *>(smem + kObuf + ((c >> 3) & 1u) * OutBuf::BYTES);
    uint32_t od = 0;  // the chunk's 16 bins packed as they come (OutBuf: bit g of byte b = bin 4 g + b)
    if (i0 < nmax) {
      src.boundary();
      const uint32_t slot = kRing + (c & 1u) * kSlotBytes + lane16;
      const uint32_t kind = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const lds_u32 *>((uintptr_t)(kKind + 4u * (c & 1u))));
      uint32_t oa = 0, ob = 0;
      if (ISSCABAC_DEC_PRIO) __builtin_amdgcn_s_setprio(ISSCABAC_DEC_PRIO);
      if (kind == kPlain) {
        // k_decode's latency-oriented step: a bin's entry is read right behind the decision of the bin before it (its state
        // may be that bin's successor: forwarded in a register for a repeated context, else loaded two bins ahead)
        const uint4 p0 = get_rows(slot), p1 = get_rows(slot + 1024u);
        auto addr_of = [&](int i) {
          const uint32_t w = word_of(i < 8 ? p0 : p1, (i >> 1) & 3);
          return (i & 1) ? w >> 16 : w & 0xFFFFu;
        };
        uint32_t at_cur = addr_of(0), at_nxt = addr_of(1);
        const uint32_t q0 = stx.load_at(at_cur);
        uint32_t qld_nxt = stx.load_at(at_nxt);
        uint64_t same_nxt;
        asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(same_nxt) : "v"(at_nxt), "v"(at_cur));
        uint64_t ent = stx.entry_raw(q0);
        uint32_t psel = 0, psh = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (g == 2) src.ensure(47);
          dec_topup_abs(da, src);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int i = 4 * g + k;
            if (k == 0) {  // right behind the top-up: R's leading one is bit 31
              psel = da.R >> 29;
              psh = 23u;  // the comparison point
            }
            const uint32_t nx = dec_bin_nx_abs_at(da, ent, psel, psh);
            if (i < 15) {
              uint32_t qn;
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(qn) : "v"(qld_nxt), "v"(nx), "s"(same_nxt));
              ent = stx.entry_raw(qn);
            }
            __builtin_amdgcn_sched_barrier(0);
            stx.store_at(at_cur, nx);
            uint32_t at_n2 = 0, qld_n2 = 0;
            if (i < 14) {
              at_n2 = addr_of(i + 2);
              qld_n2 = stx.load_at(at_n2);
            }
            if (k == 0) put_out_byte<0>(oa, ob, nx);
            else if (k == 1) put_out_byte<1>(oa, ob, nx);
            else if (k == 2) put_out_byte<2>(oa, ob, nx);
            else put_out_byte<3>(oa, ob, nx);
            if (i < 14) asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(same_nxt) : "v"(at_n2), "v"(at_nxt));
            if (k < 3) {
              abs_positions(da.R, psel, psh);
            }
            at_cur = at_nxt;
            at_nxt = at_n2;
            qld_nxt = qld_n2;
            __builtin_amdgcn_sched_barrier(0);
          }
          od |= (out_word(oa, ob) & 0x01010101u) << g;
        }
      } else {
        const uint4 wc = get_rows(slot + 2048u);
        if (kind == kMixed) {
          // context-coded and bypass bins only: one select-based step for every lane (dec_bin_mixed_nx_abs), see k_decode
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (g == 2) src.ensure(47);
            const uint32_t cw = word_of(wc, g);
            dec_topup_abs(da, src);
            const CtxQuad zs = stx.prep4(cw & 0x3F3F3F3Fu);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const uint32_t ak = stx.addr32_of(zs, k);
              const uint32_t mbk = byte_bit_mask(cw, k, 7);
              const uint32_t nx = k == 0 ? dec_bin_mixed_nx_abs<true>(da, mbk, ak, stx.load_at(ak), stx)
                                         : dec_bin_mixed_nx_abs<false>(da, mbk, ak, stx.load_at(ak), stx);
              if (k == 0) put_out_byte<0>(oa, ob, nx);
              else if (k == 1) put_out_byte<1>(oa, ob, nx);
              else if (k == 2) put_out_byte<2>(oa, ob, nx);
              else put_out_byte<3>(oa, ob, nx);
            }
            od |= (out_word(oa, ob) & 0x01010101u) << g;
          }
        } else {
          // fixed-probability bins, or streams that end inside this chunk: every lane takes its own branch (k_decode)
          const bool full = i0 + 16 <= nmin;
          dec_from_abs(d, da);
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (g == 2) src.ensure(47);
            const uint32_t cw = word_of(wc, g);
            dec_topup_nocheck(d, src);
            const CtxQuad z = stx.prep4(cw);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (k == 3) dec_guard(d, src, 8);
              uint32_t w = 0;
              if (full || i0 + 4 * g + k < n) {
                const uint32_t cb = (cw >> (8 * k)) & 0xFFu;
                if (is_ctx_byte<1>(cb, 64u)) {
                  const uint32_t ak = stx.addr32_of(z, k);
                  w = dec_bin_nx(d, ak, stx.load_at(ak), stx);
                } else if (cb == kBypass || cb == kProbBase + 50u) {
                  w = dec_bin_ep(d);
                } else if (is_prob_ctx_byte<1>(cb) && prob_valid(cb - kProbBase)) {
                  w = dec_bin_prob(d, cb - kProbBase, c_prob_rows[prob_row_index(cb - kProbBase)]);
                }
              }
              if (k == 0) put_out_byte<0>(oa, ob, w);
              else if (k == 1) put_out_byte<1>(oa, ob, w);
              else if (k == 2) put_out_byte<2>(oa, ob, w);
              else put_out_byte<3>(oa, ob, w);
            }
            od |= (out_word(oa, ob) & 0x01010101u) << g;
          }
          da = dec_to_abs(d);
        }
      }
      if (ISSCABAC_DEC_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    obuf[OutBuf::idx(cidx, lane)] = od;
    __syncthreads();  // the chunk's bins are in the staging buffer; chunk c + 1 is in its slot
  }
  if (valid && a.status) {
    uint64_t first = a.offsets[s];
    (void)clip_stream(first, first, a.payload_bytes);
    const uint32_t sk = (uint32_t)first & 3u;
    const uint8_t *p = a.payload + first;
    dec_from_abs(d, da);
    a.status[s] = dec_finish(d, src.bpos - 8u * sk, src.len - sk, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

bool decode_pair_enabled() { return true; }

hipError_t run_decode_pair(const DecArgs &a, hipStream_t stream) {
  static int static_lds = -1;  // absolute LDS addresses, as in run_encode_pair
  if (static_lds < 0) {
    hipFuncAttributes fa;
    const hipError_t err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_decode_pair));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  const uint64_t nblk = (a.S + 63) / 64;
  hipLaunchKernelGGL(k_decode_pair, dim3((unsigned)nblk), dim3(256), dpair::kLdsBytes + 32, stream, a);  // (four waves start, two stay)
  return hipGetLastError();
}

}  // namespace isscabac
