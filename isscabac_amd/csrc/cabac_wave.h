// cabac_wave.h -- what the batch kernels of the engine share (device only, gfx950): the LDS tables and table layout, the
// input tiles (64 streams x 64 or 128 bytes per load round through a 2 KiB LDS half tile), the context-state image
// policy, the byte sink and the kernel argument blocks.  Included by cabac_kernels.hip (the one-wave-per-64-streams
// kernels) and cabac_pair.hip (the producer / consumer wave pair for small launches).  Not installed.
#ifndef ISSCABAC_WAVE_H
#define ISSCABAC_WAVE_H
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "cabac_internal.h"
#include "cabac_lane.h"


namespace isscabac {

static __device__ const uint32_t c_lps_rows[64] = {ISSCABAC_LPS_ROWS};
static __device__ const uint32_t c_prob_rows[49] = {ISSCABAC_PROB_ROWS};  // fixed-probability bins: rare, read from memory
struct NextTab {
  uint32_t e[64];
  uint32_t e16[64];  // two 16-bit entries per dword: packed states 2i, 2i + 1
  uint32_t eb[64];   // four bytes per dword: (packed state, bin) = (2i, 0), (2i, 1), (2i + 1, 0), (2i + 1, 1)
  uint32_t ep[64];   // four bytes per dword: (packed state, path) with path 0 = LPS, 1 = MPS: e16 with its bytes swapped
  constexpr NextTab() : e{}, e16{}, eb{}, ep{} {
    constexpr uint8_t trans[64] = {ISSCABAC_TRANS_LPS};
    for (uint32_t s = 0; s < 64; s++) {
      e[s] = next_state_entry(s, trans[s]);
      e16[s] = next_state_entry16(2 * s, trans[s]) | next_state_entry16(2 * s + 1, trans[s]) << 16;
      ep[s] = ((e16[s] & 0x00FF00FFu) << 8) | ((e16[s] >> 8) & 0x00FF00FFu);
      eb[s] = next_state_by_bin(2 * s, 0, trans[s]) | next_state_by_bin(2 * s, 1, trans[s]) << 8 |
              next_state_by_bin(2 * s + 1, 0, trans[s]) << 16 | next_state_by_bin(2 * s + 1, 1, trans[s]) << 24;
    }
  }
};
static __device__ const NextTab c_next{};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // what the nontemporal builtins accept
__device__ __forceinline__ uint4 nt_load16(const uint8_t *p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ld16(const uint8_t *p) {  // (through the plain vector type: HIP's uint4 keeps the tile arrays in scratch here)
  const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store16(uint8_t *p, uint4 v) {
  u32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
}

// ctx bytes that the plain 16-bin path cannot take: bypass (0xFF), contexts >= 128 and, with the 4 KiB state image of
// <= 64 contexts, anything >= 64 -- an out-of-range context index must not reach another stream's LDS.  Such chunks take
// the general path, which looks at every byte and ignores an invalid one (that stream's result is unspecified, its
// neighbours' are not affected).
template <int CM>
constexpr uint32_t not_plain_mask() { return CM == 1 ? 0xC0C0C0C0u : 0x80808080u; }
// CM 2 (> 128 contexts): the state image is as large as the batch's context count asks for (a multiple of four contexts),
// `lim` = min(contexts of the image, 255): bytes from lim on are bypass bins (0xFF) or invalid (skipped, like everywhere)
template <int CM>
__device__ __forceinline__ bool is_ctx_byte(uint32_t c, uint32_t lim) { return CM == 1 ? c < 64u : CM == 3 ? c < 128u : c < lim; }
template <int CM>
__device__ __forceinline__ bool is_prob_ctx_byte(uint32_t c) { return CM != 2 && is_prob_byte(c); }

// waves per workgroup of the <= 64-context instances: 4.  With more than 64 contexts (8 .. 16 KiB of states per wave) the launcher
// picks 1 .. 4 by what fills a CU's LDS best (cabac_kernels.hip: waves_per_block_for)
#ifndef ISSCABAC_ENC_WPB
#define ISSCABAC_ENC_WPB 4
#endif
#ifndef ISSCABAC_DEC_WPB
#define ISSCABAC_DEC_WPB 4
#endif
// Issue priority of a wave while it codes a chunk's bins (s_setprio; 0 elsewhere: tile moves, loads, stores, window
// boundaries).  A SIMD's four waves are in different phases; preferring the ones inside the serial bin arithmetic over the
// ones that are about to wait for memory anyway measured, medians of six alternating runs on one box: k_encode 2.925 ->
// 2.894 ms, k_decode 2.827 -> 2.798 ms, round trip 713 -> 722 Gbin/s (levels 1, 2 and 3 alike).
#ifndef ISSCABAC_DEC_PRIO
#define ISSCABAC_DEC_PRIO 1
#endif
#ifndef ISSCABAC_ENC_PRIO
#define ISSCABAC_ENC_PRIO 1
#endif
// ------------------------------------------------------------------------------------------------
// Waves where the SIMDs are.  The hardware puts the four waves of a 256-thread workgroup on the CU's four SIMDs, one each, and spreads
// the waves of smaller workgroups unevenly (tools/ubench/simd_place.hip: eight two-wave workgroups on a CU -- the wave pairs at 128 Ki
// streams -- leave 5.3 waves on a SIMD on average and up to 8, where 4 would be even; twelve single waves 4.1 and up to 6), and the
// busiest SIMD is the launch's time.  A kernel whose workgroup is KEEP < 4 waves therefore starts FOUR: they tell each other their SIMD
// through 32 bytes of LDS, KEEP of them stay and the others leave at once (an ended wave no longer takes part in the workgroup's
// barriers).  Which stay comes from a per-CU ticket (how many such workgroups have started on this CU; never reset, only its two
// low bits count), so that the workgroups that share a CU fill its SIMDs evenly:
//   KEEP 2: the two on SIMDs (2 h, 2 h + 1), h = ticket & 1; their order swaps with ticket & 2 -- the wave pairs' two roles then sit
//           on every SIMD alike (all producers on two SIMDs and all consumers on the other two: slower than no pinning at all,
//           profiles/r06/u_sweep_by_pin.txt);
//   KEEP 3: all but the one on SIMD ticket & 3;      KEEP 1: the one on SIMD ticket & 3.
// Returns the wave's index among those that stay (0 .. KEEP - 1) or -1: leave.  Contains a workgroup barrier (every thread of
// the workgroup must call it; writes of the caller in front of it are visible behind it) and a second one if `reuse`: the caller
// is about to overwrite `ids`.  If the four waves are NOT on four different SIMDs, the waves' own indices take the SIMDs' place.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pinned_wave_index(uint32_t *ids, uint32_t *tickets, int tid, int keep, bool reuse) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint32_t simd = (hw >> 4) & 3u;
  if ((tid & 63) == 0) ids[tid >> 6] = simd;
  if (tid == 0) ids[4] = atomicAdd(&tickets[((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu)], 1u);  // (HW_ID 15:8: se | sh | cu)
  __syncthreads();
  const uint32_t i0 = ids[0], i1 = ids[1], i2 = ids[2], i3 = ids[3], j = ids[4];
  if (reuse) __syncthreads();
  const bool perm = ((1u << i0) | (1u << i1) | (1u << i2) | (1u << i3)) == 0xFu;
  const uint32_t me = perm ? simd : (uint32_t)tid >> 6;
  if (keep == 2) return (me >> 1) != (j & 1u) ? -1 : (int)((me & 1u) ^ ((j >> 1) & 1u));
  if (keep == 3) return me == (j & 3u) ? -1 : (int)((me - (j & 3u) - 1u) & 3u);
  return me == (j & 3u) ? 0 : -1;
}
constexpr int kPinTickets = 2048;  // (a __device__ uint32_t array of this size per translation unit that pins)

constexpr int enc_waves_per_block(int cm) { return cm == 1 ? ISSCABAC_ENC_WPB : 1; }
constexpr int dec_waves_per_block(int cm) { return cm == 1 ? ISSCABAC_DEC_WPB : 1; }
constexpr int kTabBytes = 1024;  // k_encode's tables: the 8-byte entries of the single-lookup steps (cabac_lane.h: enc_bin_row_acc,
                                 // enc_bin_at_row) by packed state -- the state's row, its two successors by bin, two flag bytes.
                                 // (Until round 6 two byte tables stood in front of them, 768 bytes, for the lane-by-lane step;
                                 // without them a CU holds a tenth wave of a 190-context batch.)
constexpr int kTabRowOff = 0;

// ------------------------------------------------------------------------------------------------
// small wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  // all lanes of a wave run in lockstep; this only stops the compiler from moving LDS accesses of
  // different lanes past each other and drains lgkmcnt
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
// some byte of the chunk's 16 ctx bytes names no context of the image (wave-uniform answer).  CM 2: byte >= lim, lim in
// 129..255, as a carry into bit 7 of the low seven bits (kadd = (256 - lim) in every byte, no carry between bytes)
template <int CM>
__device__ __forceinline__ bool chunk_not_plain(const uint4 &wc, uint32_t kadd) {
  if (CM != 2) return wave_any(((wc.x | wc.y | wc.z | wc.w) & not_plain_mask<CM>()) != 0);
  auto f = [&](uint32_t x) { return ((x & 0x7F7F7F7Fu) + kadd) & x; };
  return wave_any(((f(wc.x) | f(wc.y) | f(wc.z) | f(wc.w)) & 0x80808080u) != 0);
}
// fixed-probability bins exist in batches of at most 128 contexts (with more, bytes 129..254 are context indices)

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------
// 64 streams x 64 bytes per load round, 32 bytes per stream resident in LDS at a time
// ------------------------------------------------------------------------------------------------
// A wave fetches 64 contiguous bytes of each of its 64 streams per round (4 lanes x 16 B cover one stream:
// every HBM request is a full 64-byte segment -- 32-byte segments were measured to double FETCH_SIZE), keeps
// them in registers, and moves one 32-byte half at a time through LDS, where lane r reads its own stream's
// row back with ds_read_b128.  Halving the LDS residency is what lets 4 workgroups share a CU.
// LPR = lanes per row and load instruction: 4 -> 64-byte row segments, a load round is 64 bytes per stream in 4
// instructions per array (16 registers); 8 -> 128-byte segments, 128 bytes per stream in 8 instructions (32 registers).
// Whole 128-byte lines per request are what the memory system likes (tools/ubench/store_seg.hip: two arrays read this
// way, 1.30 ms against 2.54 ms with 64-byte segments at the coder kernels' occupancy), registers permitting.
// HOLE (k_encode's ctx tile with <= 64 contexts): rows 56..63 sit 256 bytes further up -- the 256 bytes they would occupy are the
// dummy row of the mixed step (group 31 of the wave's LDS region: where a bypass byte's state address lands under the step's 5-bit
// group mask; cabac_kernels.hip: WaveLds).  A shift by a whole bank row: the reads and writes stay conflict-free.
// Q (k_encode with more than 64 contexts): 16 bytes per stream resident instead of 32 -- a tile of 1 KiB instead of 2, which with
// 8 .. 16 KiB of state image per wave is a wave more on the CU or two; a part is then ONE chunk of 16 bins, written by a quarter of
// the lanes (four stores per part as before: twice the LDS stores per bin).
template <int LPR, bool HOLE = false, bool Q = false>
struct TileT {
  static constexpr int T = 16 * LPR;  // bytes per stream per load round
  static constexpr int H = Q ? 16 : 32;  // bytes per stream resident in LDS
  static constexpr int NLD = LPR;     // wave-wide load instructions per round (64 / LPR streams each)
  static constexpr int NP = T / H;    // H-byte parts per round
  static_assert(!Q || LPR == 4, "quarter residency: four lanes per row");
  // LDS image: row r = 32 bytes = two 16-byte chunks; chunk c sits in slot c ^ ((r >> 3) & 1), which makes
  // the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-byte slots of the 256-byte bank row
  __device__ static __forceinline__ uint32_t lds_off(uint32_t row, uint32_t c) {
    if (Q) return row * 16;  // (consecutive rows: every store and read covers whole bank rows)
    return row * H + (((c ^ (row >> 3)) & 1u) << 4) + ((HOLE && row >= 56u) ? 256u : 0u);
  }
  __device__ static __forceinline__ uint32_t row_of(int k, int lane) { return k * (64 / LPR) + lane / LPR; }
  __device__ static __forceinline__ uint32_t ch_of(int lane) { return lane % LPR; }

  // rows = this wave's first row in the [S][pitch] array (wave-uniform pointer), nrows = how many of the wave's 64 rows
  // exist (wave-uniform); the per-lane part of the address is a 32-bit offset (pitch <= 2^25, checked by the C ABI), so
  // the row addresses cost one VGPR each instead of hoisted 64-bit pointers
  // NT: non-temporal (streaming) loads -- the decode kernel's ctx stream then stops evicting the payload lines its
  // lanes come back to (FETCH_SIZE 11.2 -> 8.9 GB per launch, 2% faster); in the encode kernel they only cost time
  // A piece that would start past the row's end (off + 16 * piece >= pitch: the last round of a row whose pitch is no
  // multiple of T) is clamped to the row's last 16 bytes instead of predicated away: no lane reads what it gets -- those
  // bins lie past every stream's end -- and the loads stay unconditional (a predicate costs a zero fill of the registers,
  // exec-mask code around every instruction and, in front of the fill, a wait for everything in flight)
  template <bool NT = false>
  __device__ static __forceinline__ void load(uint4 (&r)[NLD], const uint8_t *rows, uint32_t nrows, uint32_t pitch,
                                              uint32_t off, int lane) {
    const uint32_t boff = min(off + ch_of(lane) * 16, pitch - 16u);
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const uint32_t row = min(row_of(k, lane), nrows - 1u);
      const uint32_t voff = row * pitch + boff;
      r[k] = NT ? nt_load16(rows + voff) : ld16(rows + voff);
    }
  }
  // The same for a wave whose 64 streams are NOT neighbours in the array (length-bucketed launches, cabac_order.hip):
  // rid[k] = stream id of the row this lane fetches a piece of with load k; one 64-bit multiply-add per load
  template <bool NT = false>
  __device__ static __forceinline__ void load_rows(uint4 (&r)[NLD], const uint8_t *base, const uint32_t (&rid)[NLD], uint32_t pitch,
                                                   uint32_t off, int lane) {
    const uint32_t boff = min(off + ch_of(lane) * 16, pitch - 16u);
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const uint8_t *p = base + ((uint64_t)rid[k] * pitch + boff);
      r[k] = NT ? nt_load16(p) : ld16(p);
    }
  }
  // ... with the ids read from LDS (ids[row]) when the loads are issued instead of kept in NLD registers
  template <bool NT = false>
  __device__ static __forceinline__ void load_rows_ids(uint4 (&r)[NLD], const uint8_t *base, const uint32_t *ids, uint32_t pitch,
                                                       uint32_t off, int lane) {
    const uint32_t boff = min(off + ch_of(lane) * 16, pitch - 16u);
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const uint8_t *p = base + ((uint64_t)ids[row_of(k, lane)] * pitch + boff);
      r[k] = NT ? nt_load16(p) : ld16(p);
    }
  }
  // part h (bytes H h .. H h + H - 1 of the round) of the registers -> LDS
  __device__ static __forceinline__ void half_to_lds(uint8_t *tile, const uint4 (&r)[NLD], int h, int lane) {
    if (Q ? (int)ch_of(lane) == h : (int)(ch_of(lane) >> 1) == h) {
#pragma unroll
      for (int k = 0; k < NLD; k++) *reinterpret_cast<uint4 *>(tile + lds_off(row_of(k, lane), ch_of(lane) & 1)) = r[k];
    }
  }
  __device__ static __forceinline__ uint4 row_chunk(const uint8_t *tile, int lane, int j) {
    return *reinterpret_cast<const uint4 *>(tile + lds_off(lane, j));
  }
  __device__ static __forceinline__ void put_row_chunk(uint8_t *tile, int lane, int j, uint4 v) {
    *reinterpret_cast<uint4 *>(tile + lds_off(lane, j)) = v;
  }
};
// The same round with the lanes paired differently: lanes 2 r and 2 r + 1 fetch 32 contiguous bytes of row r (+ 32 for odd k),
// instruction k covers 32 rows x 32 bytes, and the instructions k = 2 h and 2 h + 1 together hold exactly part h of all 64 rows
// -- so a part goes to LDS in TWO ds_write_b128 with every lane active, where TileT<4> needs four with half of the lanes and
// TileT<8> eight with a quarter of them (a wave-wide LDS store costs its 13 transfer cycles whatever the number of lanes).
// The instructions k and k + 2 ask for the two (T = 64) or more pieces of the same lines back to back.
template <int LPR>
struct TilePairT {
  static constexpr int T = 16 * LPR, H = 32, NLD = LPR, NP = T / H;
  __device__ static __forceinline__ uint32_t lds_off(uint32_t row, uint32_t c) { return row * H + (((c ^ (row >> 3)) & 1u) << 4); }
  __device__ static __forceinline__ uint32_t row_of(int k, int lane) { return (uint32_t)(lane >> 1) + 32u * (uint32_t)(k & 1); }
  __device__ static __forceinline__ uint32_t ch_of(int k, int lane) { return 2u * (uint32_t)(k >> 1) + (uint32_t)(lane & 1); }
  template <bool NT = false>
  __device__ static __forceinline__ void load(uint4 (&r)[NLD], const uint8_t *rows, uint32_t nrows, uint32_t pitch, uint32_t off,
                                              int lane) {
    const uint32_t ro0 = min(row_of(0, lane), nrows - 1u) * pitch, ro1 = min(row_of(1, lane), nrows - 1u) * pitch;
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const uint32_t voff = ((k & 1) ? ro1 : ro0) + min(off + ch_of(k, lane) * 16, pitch - 16u);  // (clamped like TileT::load)
      r[k] = NT ? nt_load16(rows + voff) : ld16(rows + voff);
    }
  }
  __device__ static __forceinline__ void half_to_lds(uint8_t *tile, const uint4 (&r)[NLD], int h, int lane) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) *reinterpret_cast<uint4 *>(tile + lds_off(row_of(kk, lane), lane & 1)) = r[2 * h + kk];
  }
  __device__ static __forceinline__ uint4 row_chunk(const uint8_t *tile, int lane, int j) {
    return *reinterpret_cast<const uint4 *>(tile + lds_off(lane, j));
  }
  // (the length-bucketed instances keep TileT -- their rows are not neighbours --; these two only let `if (PERM)` compile)
  template <bool NT = false>
  __device__ static __forceinline__ void load_rows(uint4 (&)[NLD], const uint8_t *, const uint32_t (&)[NLD], uint32_t, uint32_t, int) {}
  template <bool NT = false>
  __device__ static __forceinline__ void load_rows_ids(uint4 (&)[NLD], const uint8_t *, const uint32_t *, uint32_t, uint32_t, int) {}
};
// Whole 128-byte lines per stream and load round, asked for as TWO 64-byte segments back to back: load k (k < 4) fetches bytes
// 0..63 of rows 16 k .. 16 k + 15 like TileT<4>, load k + 4 bytes 64..127 of the same rows, issued right behind it -- the L2 fills
// each line once (the second request finds the fill pending) where TileT<4>'s rounds, 64 bins apart, fill it twice.  The parts go
// to LDS exactly like TileT<4>'s (four half-wave stores per part; TileT<8> needs eight quarter-wave ones).  32 registers per array.
template <bool HOLE = false, bool Q = false>  // (as in TileT)
struct TileLine4T {
  static constexpr int T = 128, H = Q ? 16 : 32, NLD = 8, NP = T / H;
  __device__ static __forceinline__ uint32_t lds_off(uint32_t row, uint32_t c) {
    if (Q) return row * 16;
    return row * H + (((c ^ (row >> 3)) & 1u) << 4) + ((HOLE && row >= 56u) ? 256u : 0u);
  }
  __device__ static __forceinline__ uint32_t row_of(int k, int lane) { return (k & 3) * 16 + lane / 4; }
  __device__ static __forceinline__ uint32_t ch_of(int lane) { return lane % 4; }
  template <bool NT = false>
  __device__ static __forceinline__ void load(uint4 (&r)[NLD], const uint8_t *rows, uint32_t nrows, uint32_t pitch, uint32_t off, int lane) {
    const uint32_t b0 = min(off + ch_of(lane) * 16, pitch - 16u), b1 = min(off + 64u + ch_of(lane) * 16, pitch - 16u);  // (clamped like TileT::load)
    const uint32_t ro0 = min(row_of(0, lane), nrows - 1u) * pitch, ro1 = min(row_of(1, lane), nrows - 1u) * pitch;
    const uint32_t ro2 = min(row_of(2, lane), nrows - 1u) * pitch, ro3 = min(row_of(3, lane), nrows - 1u) * pitch;
    r[0] = NT ? nt_load16(rows + (ro0 + b0)) : ld16(rows + (ro0 + b0));
    r[4] = NT ? nt_load16(rows + (ro0 + b1)) : ld16(rows + (ro0 + b1));
    r[1] = NT ? nt_load16(rows + (ro1 + b0)) : ld16(rows + (ro1 + b0));
    r[5] = NT ? nt_load16(rows + (ro1 + b1)) : ld16(rows + (ro1 + b1));
    r[2] = NT ? nt_load16(rows + (ro2 + b0)) : ld16(rows + (ro2 + b0));
    r[6] = NT ? nt_load16(rows + (ro2 + b1)) : ld16(rows + (ro2 + b1));
    r[3] = NT ? nt_load16(rows + (ro3 + b0)) : ld16(rows + (ro3 + b0));
    r[7] = NT ? nt_load16(rows + (ro3 + b1)) : ld16(rows + (ro3 + b1));
  }
  // h may be a run-time value (the kernels keep the loop over the parts rolled): the parts always leave registers 0..3, and the
  // line's second half moves down into them before part 2 (16 moves per 128 bytes of every row; a register index computed from h
  // would put the array into scratch)
  __device__ static __forceinline__ void half_to_lds(uint8_t *tile, uint4 (&r)[NLD], int h, int lane) {
    if (h == NP / 2) {
      r[0] = r[4];
      r[1] = r[5];
      r[2] = r[6];
      r[3] = r[7];
    }
    if (Q ? (int)ch_of(lane) == (h & 3) : (int)(ch_of(lane) >> 1) == (h & 1)) {
#pragma unroll
      for (int k = 0; k < 4; k++) *reinterpret_cast<uint4 *>(tile + lds_off(row_of(k, lane), ch_of(lane) & 1)) = r[k];
    }
  }
  __device__ static __forceinline__ uint4 row_chunk(const uint8_t *tile, int lane, int j) {
    return *reinterpret_cast<const uint4 *>(tile + lds_off(lane, j));
  }
  // (the length-bucketed instances keep TileT<4> -- these only let `if (PERM)` compile)
  template <bool NT = false>
  __device__ static __forceinline__ void load_rows(uint4 (&)[NLD], const uint8_t *, const uint32_t (&)[NLD], uint32_t, uint32_t, int) {}
};
typedef TileLine4T<false> TileLine4;
typedef TileT<4> Tile;  // the decode kernel's (16 registers for its one input array)
// the encode kernel's two input arrays, bins and ctx: 64-byte segments (128-byte ones measured slower here: 3.26 ms with
// both, 3.09 with either, 3.06 with neither -- the encode kernel writes little, its reads are not what holds it up)
constexpr bool kEncNT = false;  // encode kernel: plain loads (streaming ones measured 4.13 ms against 3.02 with 64-byte segments, 2.86
                                // against 2.79 with whole lines: profiles/r06/ab_line.txt)
// Round 6, same box, medians of 5 (profiles/r06/ab_line.txt): both arrays by whole lines k_encode 2.885 -> 2.790 ms and FETCH_SIZE
// 8.56 -> 4.30 GB per launch (L2 <-> fabric traffic 17.9 -> 9.5 GB = 1.05 x the algorithmic bytes); either array alone 2.876 / 2.857;
// with streaming (nt) loads on top 2.862.  125 VGPRs, no spill (TileT<8> for both: 128 + 2 spilled, 8 quarter-wave LDS stores per part).
// (more than 128 contexts -- CM 2, at most 2.5 waves per SIMD -- keep the 64-byte segments: 190 contexts 3.98 -> 4.19 ms, 253 contexts
// 4.77 -> 4.95 ms with whole lines, 127 contexts 3.72 -> 3.63: profiles/r06/c_other_shapes.txt)
template <int CM, bool PERM, bool HOLE = false>
using EncTile = typename std::conditional<!PERM && CM != 2, TileLine4T<HOLE, CM != 1>, TileT<4, HOLE, CM != 1>>::type;  // (the bucketed instances keep 64-byte segments: 16 + 16 registers)

// The decode kernel's output staging.  A chunk's 16 output bytes (bit 0 of each = the bin, see cabac_lane.h: dec_bin_out)
// are packed four to a byte -- bit i of byte b of the dword = bin 4 i + b -- so that 128 bins of 64 streams fit 2 KiB and
// leave as full 128-byte row segments (64-byte segments, one per load round, were measured at 3.1 TB/s against 4.7 for
// load + store: tools/ubench/tile_bw.hip, store_seg.hip -- the kernel was store-bound with them).  Dword (c, row) of the
// eight chunks c sits at c * 64 + (row ^ 8 c): a lane writes its own row (all banks distinct), and the store pass
// reads (8 rows) x (8 chunks) per instruction, again all banks distinct.
// GAP (k_decode with <= 64 contexts): chunk 7 sits 256 bytes further up -- the 256 bytes it would occupy are the dummy row of the
// mixed step (cabac_lane.h: dec_bin_mixed2_nx_abs), which the 5-bit group mask of a bypass byte's state address lands on
template <bool GAP>
struct OutBufT {
  static constexpr uint32_t BYTES = 8 * 64 * 4 + (GAP ? 256 : 0);
  __device__ static __forceinline__ uint32_t idx(uint32_t c, uint32_t row) { return c * 64 + (row ^ (c * 8)) + ((GAP && c == 7) ? 64u : 0u); }
  __device__ static __forceinline__ uint32_t pack(const uint4 &v) {
    const uint32_t M = 0x01010101u;
    uint32_t d = v.x & M;
    d |= (v.y & M) << 1;
    d |= (v.z & M) << 2;
    d |= (v.w & M) << 3;
    return d;
  }
  __device__ static __forceinline__ uint4 unpack(uint32_t d) {
    const uint32_t M = 0x01010101u;
    return make_uint4(d & M, (d >> 1) & M, (d >> 2) & M, (d >> 3) & M);
  }
};
typedef OutBufT<false> OutBuf;

__device__ __forceinline__ uint32_t word_of(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void set_word(uint4 &v, int i, uint32_t w) {
  if (i == 0) v.x = w;
  else if (i == 1) v.y = w;
  else if (i == 2) v.z = w;
  else v.w = w;
}
__device__ __forceinline__ uint32_t byte_of(const uint4 &v, int k) { return (word_of(v, k >> 2) >> (8 * (k & 3))) & 0xFFu; }
// a word whose bit 1 is bin k of the chunk (bins are bytes 0/1): what enc_bin's b1 wants, one shift
__device__ __forceinline__ uint32_t bin_at_bit1(const uint4 &v, int k) {
  const uint32_t w = word_of(v, k >> 2);
  return (k & 3) == 0 ? (w << 1) : (w >> (8 * (k & 3) - 1));
}

// Bin policy of the encode kernel (cabac_lane.h: enc_bin_at): the four bins of one input word, pre-shifted so that bin k
// sits at bit 1 of byte k; xor with the state picks the byte in the same instruction (SDWA operand select)
template <int K>
struct BinOfWord {
  uint32_t w2;  // input word << 1
  uint32_t w;   // input word
  uint32_t w8;  // input word << 3: byte k = 8 * bin k, what next_of shifts by (only the single-lookup step reads it)
  // the successor for bin K out of the entry's high word (byte 0: bin 0, byte 1: bin 1); the store takes the low byte
  __device__ __forceinline__ uint32_t next_of(uint32_t hi) const {
    uint32_t r;
    if (K == 0) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w8), "v"(hi));
    else if (K == 1) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w8), "v"(hi));
    else if (K == 2) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w8), "v"(hi));
    else asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w8), "v"(hi));
    return r;
  }
  __device__ __forceinline__ uint32_t or_q(uint32_t q) const {
    uint32_t r;
    if (K == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(q), "v"(w));
    else if (K == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(q), "v"(w));
    else if (K == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(q), "v"(w));
    else asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(q), "v"(w));
    return r;
  }
  __device__ __forceinline__ uint32_t lps_mask(uint32_t, uint32_t t) const { return (uint32_t)__builtin_amdgcn_sbfe(t, 16, 1); }
  // x | byte K of y, x + byte K of y, x * byte K of y (x below 2^24): one SDWA instruction each -- per-bin fields of a per-word
  // SWAR word without an extraction of their own
  __device__ __forceinline__ uint32_t or_byte(uint32_t x, uint32_t y) const {
    uint32_t r;
    if (K == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "v"(y));
    else asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "v"(y));
    return r;
  }
  __device__ __forceinline__ uint32_t add_byte(uint32_t x, uint32_t y) const {
    uint32_t r;
    if (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "v"(y));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "v"(y));
    return r;
  }
  __device__ __forceinline__ uint32_t mul_byte(uint32_t x, uint32_t y) const {
    uint32_t r;
    if (K == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "v"(y));
    else if (K == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "v"(y));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "v"(y));
    return r;
  }
  __device__ __forceinline__ uint32_t lps_mask_sext(uint32_t t) const {  // the shifted entry's flag byte (0x00 / 0xFF) as a 32-bit mask
    uint32_t m;
    asm("v_mov_b32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(m) : "v"(t));
    return m;
  }
  __device__ __forceinline__ uint32_t xor_q(uint32_t q) const {
    uint32_t r;
    if (K == 0) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(q), "v"(w2));
    else if (K == 1) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(q), "v"(w2));
    else if (K == 2) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(q), "v"(w2));
    else asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(q), "v"(w2));
    return r;
  }
};

// enc_pick (cabac_lane.h) for the kernel's bins: t = the entry's high word shifted by the bin, its byte 2 the flag "the bin
// is not the state's mps" (enc_entry_hi) -- one SDWA compare into VCC, two selects
// the flag byte is 0x00 / 0xFF: sign-extended by the operand select it IS the mask -- one SDWA move, a bit select and a
// full-rate and (4 + 4 + 2 issue cycles against 4 + 4 + 4 for compare + two selects; the and runs at twice the rate of
// a select on gfx950: tools/ubench/valu_ops)
template <int K>
__device__ __forceinline__ void enc_pick(const BinOfWord<K> &, uint32_t, uint32_t t, uint32_t lps, uint32_t rmps, uint32_t &r, uint32_t &a) {
  uint32_t m;
  asm("v_mov_b32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(m) : "v"(t));
  r = select_mask(m, lps, rmps);
  a = rmps & m;
}

// f(BinOfWord<k & 3>{w2, w1}) for a loop counter k that is a constant once the loop is unrolled
template <class F>
__device__ __forceinline__ auto with_bin_of_word(int k, uint32_t w2, uint32_t w1, F &&f) {
  switch (k & 3) {
    case 0: return f(BinOfWord<0>{w2, w1, w1 << 3});
    case 1: return f(BinOfWord<1>{w2, w1, w1 << 3});
    case 2: return f(BinOfWord<2>{w2, w1, w1 << 3});
    default: return f(BinOfWord<3>{w2, w1, w1 << 3});
  }
}

// ------------------------------------------------------------------------------------------------
// context-state policies (see cabac_lane.h: States)
// ------------------------------------------------------------------------------------------------
// One byte per (stream, ctx) in LDS at (ctx / 4) * 256 + lane * 4 + ctx % 4: whatever contexts the 64 lanes
// pick, lane l only ever touches bank l % 32.  The tables are byte tables in LDS, read by absolute LDS address.
struct LdsStates {
  uint8_t *base;  // wave's state region + lane * 4
  __device__ __forceinline__ uint8_t *addr(uint32_t c) const { return base + (((c & ~3u) << 6) | (c & 3u)); }
  // four contexts of one input word at a time (cabac_lane.h: lds_ctx_quad); all four must be < 128
  __device__ __forceinline__ CtxQuad prep4(uint32_t cw) const { return lds_ctx_quad(cw); }
  // address = lane base + 16-bit field of the quad, in ONE instruction (SDWA word select) and as a 32-bit LDS address
  typedef __attribute__((address_space(3))) uint8_t lds_u8;
  __device__ __forceinline__ lds_u8 *addr_of(const CtxQuad &z, int k) const {
    const uint32_t b32 = (uint32_t)(uintptr_t)(lds_u8 *)base, zz = k < 2 ? z.z01 : z.z23;
    uint32_t a;
    if ((k & 1) == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "v"(b32), "v"(zz));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a) : "v"(b32), "v"(zz));
    return (lds_u8 *)(uintptr_t)a;
  }
  __device__ __forceinline__ uint32_t load_at(const lds_u8 *a) const { return *a; }
  __device__ __forceinline__ void store_at(lds_u8 *a, uint32_t p) const { *a = (uint8_t)p; }
  __device__ __forceinline__ uint32_t load_at(const uint8_t *a) const { return *a; }
  __device__ __forceinline__ void store_at(uint8_t *a, uint32_t p) const { *a = (uint8_t)p; }
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return *addr(c); }
  __device__ __forceinline__ void store(uint32_t c, uint32_t p) const { *addr(c) = (uint8_t)p; }
  // (LDS addresses as integers: the kernels' dynamic LDS starts at 0 -- they have no static __shared__, which the engine
  // checks at start-up -- so a table's offset folds into the instruction and q itself is the address)
  // the single-lookup steps' 8-byte entry by packed state: (q >> 1) * 8 = q << 2
  // (the address as a 16-bit shift: full rate, cabac_lane.h: shl16; a 24-bit multiply is half rate like the 32-bit shift)
  __device__ __forceinline__ uint64_t entry_mem(uint32_t q) const {
    return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t *>((uintptr_t)(kTabRowOff + shl16c<2>(q)));
  }
  // the decoder's entries by RAW state byte (cabac_lane.h: dec_bin_out): 256 x 8 bytes at LDS 0 (kDecTabBytes)
  __device__ __forceinline__ uint64_t entry_raw(uint32_t qb) const {

    return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t *>((uintptr_t)(qb << 3));  // (the 16-bit shift of the encoder's address measured slower here: 2.879 vs 2.812 ms)
  }
  // byte 2 of p -> the state byte at LDS address a (ds_write_b8_d16_hi: no shift)
  __device__ __forceinline__ void store_at_hi(uint32_t a, uint32_t p) const {
    asm volatile("ds_write_b8_d16_hi %0, %1" : : "v"(a), "v"(p) : "memory");
  }
  // state addresses as plain 32-bit LDS addresses (the latency-oriented paths compare them)
  __device__ __forceinline__ uint32_t base32() const { return (uint32_t)(uintptr_t)(lds_u8 *)base; }
  __device__ __forceinline__ uint32_t addr32_of(const CtxQuad &z, int k) const { return (uint32_t)(uintptr_t)addr_of(z, k); }
  __device__ __forceinline__ uint32_t load_at(uint32_t a) const { return *(const lds_u8 *)(uintptr_t)a; }
  __device__ __forceinline__ void store_at(uint32_t a, uint32_t p) const { *(lds_u8 *)(uintptr_t)a = (uint8_t)p; }
};
// ------------------------------------------------------------------------------------------------
// byte sink: 16 bytes staged per lane in LDS, then one 16-byte store into this stream's HBM slot
// (4-byte stores cost a 32-byte HBM write each: WRITE_SIZE was 8x the payload)
// ------------------------------------------------------------------------------------------------
struct SlotSink {
  uint8_t *stage;   // LDS, this lane's 16 bytes
  uint8_t *slots;   // HBM, the wave's first slot (wave-uniform, 16-byte aligned)
  uint32_t mine;    // byte offset of this stream's slot from there (64 slots stay far below 4 GiB)
  uint32_t pos;
  bool on;
  __device__ __forceinline__ void put(uint32_t b) {
    stage[pos & 15u] = (uint8_t)b;
    pos++;
    if ((pos & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(stage);
      if (on) *reinterpret_cast<uint4 *>(slots + (mine + pos - 16u)) = v;
    }
  }
  // two bytes, high byte first; pos is even whenever this is called (16-bit digits come before any single byte)
  __device__ __forceinline__ void put16(uint32_t x) {
    *reinterpret_cast<uint16_t *>(stage + (pos & 15u)) = (uint16_t)__builtin_amdgcn_perm(x, x, 0x0C0C0001u);  // low 16 bits, byte-swapped
    pos += 2;
    if ((pos & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(stage);
      if (on) *reinterpret_cast<uint4 *>(slots + (mine + pos - 16u)) = v;
    }
  }
  __device__ __forceinline__ uint32_t bytes() const { return pos; }
  __device__ __forceinline__ void close() {
    if (on && (pos & 15u)) *reinterpret_cast<uint4 *>(slots + (mine + (pos & ~15u))) = *reinterpret_cast<const uint4 *>(stage);
  }
};

struct EncArgs {
  const uint8_t *bins, *ctx;
  const uint32_t *nbins_per_stream;
  const uint32_t *order;  // length-bucketed launch: lane l of wave w codes stream order[64 w + l] (else null)
  uint8_t *slots;
  uint32_t *lens;
  uint64_t S, slot_stride;
  uint32_t nbins, pitch, nctx;
  uint32_t img_bytes;  // CM 2: bytes of a wave's state image, 256 per four contexts
  uint32_t wpb;        // waves per workgroup of this launch (set by the launcher)
  uint32_t ctx_init[64];
};

struct DevMem {
  const uint8_t *payload;  // whole payload buffer (wave-uniform)
  uint64_t total;          // its size in bytes (wave-uniform)
  uint64_t abase;          // 4-byte aligned offset at or below the stream's first byte
  int32_t lim;             // largest off for which [abase + off, +16) lies inside the buffer (-1: none), capped at 2^31-1
  struct alignas(4) Q4 { uint32_t w[4]; };
  struct Pair { uint64_t lo, hi; };
  // the last few bytes of the whole payload buffer: never read past its end
  static __device__ __forceinline__ Pair load16_tail(const uint8_t *payload, uint64_t total, uint64_t at) {
    Pair r{0, 0};
#pragma unroll 1
    for (uint32_t i = 0; i < 16; i++) {
      const uint64_t b = at + i < total ? payload[at + i] : 0;
      if (i < 8) r.lo |= b << (8 * i);
      else r.hi |= b << (8 * (i - 8));
    }
    return r;
  }
  __device__ __forceinline__ void load16(uint32_t off, uint64_t &lo, uint64_t &hi) const {
    const uint64_t at = abase + off;
    if ((int32_t)off <= lim) {  // off is far below 2^31 (a stream's payload is bounded by 6/8 of its <= 2^25 bins)
      const Q4 q = *reinterpret_cast<const Q4 *>(payload + at);
      lo = (uint64_t)q.w[0] | ((uint64_t)q.w[1] << 32);
      hi = (uint64_t)q.w[2] | ((uint64_t)q.w[3] << 32);
    } else {
      const Pair r = load16_tail(payload, total, at);
      lo = r.lo;
      hi = r.hi;
    }
  }
};

struct DecArgs {
  const uint8_t *payload, *ctx;
  const uint64_t *offsets;
  const uint32_t *nbins_per_stream;
  const uint32_t *order;  // as in EncArgs
  uint8_t *bins_out;
  uint32_t *status;
  uint32_t *exact_flag;  // set when a stream of this launch needs k_decode_exact
  uint64_t S, payload_bytes;
  uint32_t nbins, pitch, nctx;
  uint32_t img_bytes;  // as in EncArgs
  uint32_t wpb;        // as in EncArgs
  uint32_t ctx_init[64];
};

constexpr bool kDecNT = true;  // decode kernel: ctx loads are streaming
constexpr bool kDecNTst = true;  // ... and the bin stores

// LDS of the decode kernel: the raw-byte entries (256 x 8 bytes, cabac_lane.h: dec_bin_out) at 0, then per wave the
// context states, the round tile and (bucketed launches) the wave's stream ids
constexpr int kDecTabBytes = 2048;

// the digit flush of the encode kernels: digits leave `low` when some lane could overflow it within the next four bins
#ifndef ISSCABAC_FLUSH_AT
#define ISSCABAC_FLUSH_AT 39  // four more bins add at most 22 (context) bits: W stays <= 62
#endif

// cabac_pair.hip: the encode kernel of small launches (<= 64 contexts, no length bucketing)
bool encode_pair_enabled();
hipError_t run_encode_pair(const EncArgs &a, hipStream_t stream);
bool decode_pair_enabled();
hipError_t run_decode_pair(const DecArgs &a, hipStream_t stream);

}  // namespace isscabac
#endif
