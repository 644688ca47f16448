// cabac_kernels.hip -- HIP kernels of the batched CABAC engine, written for gfx950 (MI355X, wave64).
//
// Mapping (DESIGN.md has the long version):
//   * one wavefront = 64 consecutive streams, one lane = one bitstream.  The serial range / low /
//     renormalisation dependency of CABAC stays inside a lane; lanes never talk during coding.
//   * bins / ctx arrive stream-major (each stream's bins contiguous).  A wave reads a 64 x T byte tile
//     with coalesced 16-byte loads (T/16 lanes cover one stream's T contiguous bytes), parks it in LDS
//     with an XOR swizzle and every lane then pulls its own stream's bytes back 16 at a time with
//     conflict-free ds_read_b128.  The next tile's global loads are issued before the current tile is
//     coded, so HBM latency hides behind ~T bins of arithmetic.
//   * per-stream context states: one byte per (stream, ctx) in LDS at  (ctx/4)*256 + lane*4 + ctx%4 ,
//     so every lane always hits its own bank; with <= 4 contexts they live in one VGPR instead.
//   * rangeTabLps row and transIdxLps are 64-entry tables: lane i keeps entry i in a VGPR and a lookup
//     is one ds_bpermute_b32 (LDS crossbar, no bank conflicts); divergent (ragged) sections use an LDS
//     copy instead because bpermute cannot read inactive lanes.
//   * emitted bytes go to a per-stream HBM slot of provable worst-case size; a scan over the lengths and
//     a gather kernel then produce the tight payload + offsets (the reference's one-file-per-stream
//     content, concatenated).
//   * the decoder fetches its payload bytes 16 at a time, one chunk ahead of use (WindowSrc).
//
// No MFMA anywhere: this is integer / table work bounded by VALU issue and LDS, then HBM.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "cabac_internal.h"
#include "cabac_lane.h"

namespace isscabac {

__device__ const uint32_t c_lps_rows[64] = {ISSCABAC_LPS_ROWS};
struct NextTab {
  uint32_t e[64];
  constexpr NextTab() : e{} {
    constexpr uint8_t trans[64] = {ISSCABAC_TRANS_LPS};
    for (uint32_t s = 0; s < 64; s++) e[s] = next_state_entry(s, trans[s]);
  }
};
__device__ const NextTab c_next{};

constexpr int kWavesPerBlock = 4;
constexpr int kTabBytes = 512;

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
// 64 x T byte tile: coalesced global <-> swizzled LDS <-> one row per lane
// ------------------------------------------------------------------------------------------------
template <int T>
struct Tile {
  static_assert(T == 32 || T == 64 || T == 128 || T == 256, "tile bytes per stream");
  static constexpr int NCH = T / 16;    // 16-byte chunks per row = lanes that share a row in one load
  static constexpr int RPI = 64 / NCH;  // rows covered by one wave-wide load instruction
  static constexpr int RPB = 16 / NCH;  // rows per 256-byte LDS bank row
  // chunk c of row r lives in slot c ^ f(r): the 16 lanes of every ds_read_b128 lane group then hit
  // 16 distinct 16-byte slots of the bank row
  __device__ static __forceinline__ uint32_t swz(uint32_t row) { return (row / RPB) % NCH; }
  __device__ static __forceinline__ uint32_t lds_off(uint32_t row, uint32_t ch) {
    return row * T + ((ch ^ swz(row)) << 4);
  }

  // lane -> (row, chunk) of load instruction k
  __device__ static __forceinline__ uint32_t row_of(int k, int lane) { return k * RPI + lane / NCH; }
  __device__ static __forceinline__ uint32_t ch_of(int lane) { return lane % NCH; }

  __device__ static __forceinline__ void load(uint4 (&r)[NCH], const uint8_t *base, uint64_t s0, uint64_t S,
                                              uint32_t pitch, uint32_t off, int lane) {
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      uint64_t s = s0 + row_of(k, lane);
      s = s < S ? s : S - 1;
      const uint32_t boff = off + ch_of(lane) * 16;
      r[k] = boff < pitch ? *reinterpret_cast<const uint4 *>(base + s * pitch + boff) : make_uint4(0, 0, 0, 0);
    }
  }
  __device__ static __forceinline__ void to_lds(uint8_t *tile, const uint4 (&r)[NCH], int lane) {
#pragma unroll
    for (int k = 0; k < NCH; k++)
      *reinterpret_cast<uint4 *>(tile + lds_off(row_of(k, lane), ch_of(lane))) = r[k];
  }
  __device__ static __forceinline__ uint4 row_chunk(const uint8_t *tile, int lane, int j) {
    return *reinterpret_cast<const uint4 *>(tile + lds_off(lane, j));
  }
  __device__ static __forceinline__ void put_row_chunk(uint8_t *tile, int lane, int j, uint4 v) {
    *reinterpret_cast<uint4 *>(tile + lds_off(lane, j)) = v;
  }
  __device__ static __forceinline__ void store(uint8_t *base, const uint8_t *tile, uint64_t s0, uint64_t S, uint32_t pitch,
                                               uint32_t off, int lane) {
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const uint32_t row = row_of(k, lane);
      const uint64_t s = s0 + row;
      const uint32_t boff = off + ch_of(lane) * 16;
      if (s < S && boff < pitch)
        *reinterpret_cast<uint4 *>(base + s * pitch + boff) = *reinterpret_cast<const uint4 *>(tile + lds_off(row, ch_of(lane)));
    }
  }
};

__device__ __forceinline__ uint32_t word_of(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ uint32_t byte_of(const uint4 &v, int k) { return (word_of(v, k >> 2) >> (8 * (k & 3))) & 0xFFu; }

// ------------------------------------------------------------------------------------------------
// context-state policies (see cabac_lane.h: States)
// ------------------------------------------------------------------------------------------------
// states in LDS, tables through the crossbar: only legal while all 64 lanes are active
struct LdsStatesXbar {
  uint8_t *base;  // wave's state region + lane * 4
  uint32_t row_tab, trans_tab;
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return base[((c & ~3u) << 6) | (c & 3u)]; }
  __device__ __forceinline__ void store(uint32_t c, uint32_t p) { base[((c & ~3u) << 6) | (c & 3u)] = (uint8_t)p; }
  __device__ __forceinline__ uint32_t lps_row(uint32_t s) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s << 2), (int)row_tab); }
  __device__ __forceinline__ uint32_t next_entry(uint32_t s) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s << 2), (int)trans_tab); }
};
// states in LDS, tables in LDS memory: any exec mask
struct LdsStatesMem {
  uint8_t *base;
  const uint32_t *tab;  // [0..63] rangeTabLps rows, [64..127] next-state entries
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return base[((c & ~3u) << 6) | (c & 3u)]; }
  __device__ __forceinline__ void store(uint32_t c, uint32_t p) { base[((c & ~3u) << 6) | (c & 3u)] = (uint8_t)p; }
  __device__ __forceinline__ uint32_t lps_row(uint32_t s) const { return tab[s]; }
  __device__ __forceinline__ uint32_t next_entry(uint32_t s) const { return tab[64 + s]; }
};
// <= 4 contexts: the four state bytes in one VGPR
struct RegStatesXbar {
  uint32_t *packed;
  uint32_t row_tab, trans_tab;
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return (*packed >> (8 * c)) & 0xFFu; }
  __device__ __forceinline__ void store(uint32_t c, uint32_t p) { *packed = (*packed & ~(0xFFu << (8 * c))) | (p << (8 * c)); }
  __device__ __forceinline__ uint32_t lps_row(uint32_t s) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s << 2), (int)row_tab); }
  __device__ __forceinline__ uint32_t next_entry(uint32_t s) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s << 2), (int)trans_tab); }
};
struct RegStatesMem {
  uint32_t *packed;
  const uint32_t *tab;
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return (*packed >> (8 * c)) & 0xFFu; }
  __device__ __forceinline__ void store(uint32_t c, uint32_t p) { *packed = (*packed & ~(0xFFu << (8 * c))) | (p << (8 * c)); }
  __device__ __forceinline__ uint32_t lps_row(uint32_t s) const { return tab[s]; }
  __device__ __forceinline__ uint32_t next_entry(uint32_t s) const { return tab[64 + s]; }
};

// ------------------------------------------------------------------------------------------------
// byte sink: dword accumulator -> this stream's HBM slot
// ------------------------------------------------------------------------------------------------
struct SlotSink {
  uint32_t *slot;
  uint32_t acc, nacc, wpos;
  bool on;
  __device__ __forceinline__ void put(uint32_t b) {
    acc |= b << (8 * nacc);
    nacc++;
    if (nacc == 4) {
      if (on) slot[wpos] = acc;
      wpos++;
      acc = 0;
      nacc = 0;
    }
  }
  __device__ __forceinline__ uint32_t bytes() const { return wpos * 4 + nacc; }
  __device__ __forceinline__ void close() {
    if (on && nacc) slot[wpos] = acc;
  }
};

struct EncArgs {
  const uint8_t *bins, *ctx;
  const uint32_t *nbins_per_stream;
  uint8_t *slots;
  uint32_t *lens;
  uint64_t S, slot_stride;
  uint32_t nbins, pitch, nctx;
  uint32_t ctx_init[64];
};

template <int CM>
__device__ __forceinline__ uint32_t state_bytes_per_wave() { return CM == 0 ? 0u : (CM == 1 ? 4096u : 16384u); }

// ------------------------------------------------------------------------------------------------
// encode kernel
// ------------------------------------------------------------------------------------------------
template <int T, int CM>
__global__ void __launch_bounds__(256) k_encode(EncArgs a) {
  using TL = Tile<T>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
  if (tid < 64) {
    tab[tid] = c_lps_rows[tid];
    tab[64 + tid] = c_next.e[tid];
  }
  __syncthreads();
  const uint64_t s0 = ((uint64_t)blockIdx.x * kWavesPerBlock + wib) * 64;
  if (s0 >= a.S) return;

  constexpr uint32_t SB = CM == 0 ? 0u : (CM == 1 ? 4096u : 16384u);
  uint8_t *wbase = smem + kTabBytes + wib * (SB + 2 * T * 64);
  uint8_t *tile_b = wbase + SB, *tile_c = tile_b + T * 64;

  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  const uint32_t nmax = wave_max_u32(n), nmin = wave_min_u32(n);

  // context states
  uint32_t packed = a.ctx_init[0];
  if (CM != 0) {
    uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
    const uint32_t ngrp = (a.nctx + 3) / 4;
    for (uint32_t g = 0; g < ngrp; g++) st32[g * 64] = a.ctx_init[g];
  }
  const uint32_t row_tab = c_lps_rows[lane], trans_tab = c_next.e[lane];

  EncCore e;
  enc_start(e);
  SlotSink sink;
  sink.slot = reinterpret_cast<uint32_t *>(a.slots + s * a.slot_stride);
  sink.acc = 0;
  sink.nacc = 0;
  sink.wpos = 0;
  sink.on = valid;

  const uint32_t ntiles = (nmax + T - 1) / T;
  uint4 pb[TL::NCH], pc[TL::NCH];
  if (ntiles) {
    TL::load(pb, a.bins, s0, a.S, a.pitch, 0, lane);
    TL::load(pc, a.ctx, s0, a.S, a.pitch, 0, lane);
  }
  wave_lds_sync();
  for (uint32_t t = 0; t < ntiles; t++) {
    TL::to_lds(tile_b, pb, lane);
    TL::to_lds(tile_c, pc, lane);
    wave_lds_sync();
    if (t + 1 < ntiles) {
      TL::load(pb, a.bins, s0, a.S, a.pitch, (t + 1) * T, lane);
      TL::load(pc, a.ctx, s0, a.S, a.pitch, (t + 1) * T, lane);
    }
    for (int j = 0; j < TL::NCH; j++) {
      const uint32_t i0 = t * T + j * 16;
      if (i0 >= nmax) break;
      const uint4 wb = TL::row_chunk(tile_b, lane, j);
      const uint4 wc = TL::row_chunk(tile_c, lane, j);
      const bool full = i0 + 16 <= nmin;
      const bool hi = wave_any(((wc.x | wc.y | wc.z | wc.w) & 0x80808080u) != 0);
      if (full && !hi) {
        // every lane codes 16 context bins: convergent, tables through the crossbar
        if (CM == 0) {
          RegStatesXbar st{&packed, row_tab, trans_tab};
#pragma unroll
          for (int k = 0; k < 16; k++) {
            enc_bin(e, byte_of(wb, k) & 1u, byte_of(wc, k), st);
            if ((k & 3) == 3 && wave_any(e.W >= 21)) enc_flush(e, sink);
          }
        } else {
          LdsStatesXbar st{wbase + lane * 4, row_tab, trans_tab};
#pragma unroll
          for (int k = 0; k < 16; k++) {
            enc_bin(e, byte_of(wb, k) & 1u, byte_of(wc, k), st);
            if ((k & 3) == 3 && wave_any(e.W >= 21)) enc_flush(e, sink);
          }
        }
      } else if (full) {
        // bypass bins (or contexts >= 128) somewhere in the wave: combined step, still convergent
        if (CM == 0) {
          RegStatesXbar st{&packed, row_tab, trans_tab};
#pragma unroll
          for (int k = 0; k < 16; k++) {
            enc_bin_any(e, byte_of(wb, k) & 1u, byte_of(wc, k), st);
            if ((k & 3) == 3 && wave_any(e.W >= 21)) enc_flush(e, sink);
          }
        } else {
          LdsStatesXbar st{wbase + lane * 4, row_tab, trans_tab};
#pragma unroll
          for (int k = 0; k < 16; k++) {
            enc_bin_any(e, byte_of(wb, k) & 1u, byte_of(wc, k), st);
            if ((k & 3) == 3 && wave_any(e.W >= 21)) enc_flush(e, sink);
          }
        }
      } else {
        // ragged end: lanes drop out one by one, tables from LDS memory
#pragma unroll 4
        for (int k = 0; k < 16; k++) {
          if (i0 + k < n) {
            const uint32_t bin = (word_of(wb, k >> 2) >> (8 * (k & 3))) & 1u;
            const uint32_t cx = (word_of(wc, k >> 2) >> (8 * (k & 3))) & 0xFFu;
            if (CM == 0) {
              RegStatesMem st{&packed, tab};
              enc_bin_any(e, bin, cx, st);
            } else {
              LdsStatesMem st{wbase + lane * 4, tab};
              enc_bin_any(e, bin, cx, st);
            }
          }
          if ((k & 3) == 3) enc_flush(e, sink);
        }
      }
    }
    wave_lds_sync();
  }
  enc_flush(e, sink);
  enc_finish(e, sink);
  sink.close();
  if (valid) a.lens[s] = sink.bytes();
}

// ------------------------------------------------------------------------------------------------
// decode kernel
// ------------------------------------------------------------------------------------------------
struct DevMem {
  const uint8_t *base;  // 4-byte aligned address at or below the stream's first byte
  uint64_t avail;       // bytes from base to the end of the payload buffer
  struct alignas(4) Q4 { uint32_t w[4]; };
  struct Pair { uint64_t lo, hi; };
  // the last few bytes of the whole payload buffer: never read past its end
  static __device__ __noinline__ Pair load16_tail(const uint8_t *base, uint64_t avail, uint32_t off) {
    Pair r{0, 0};
    for (uint32_t i = 0; i < 16; i++) {
      const uint64_t b = (uint64_t)off + i < avail ? base[off + i] : 0;
      if (i < 8) r.lo |= b << (8 * i);
      else r.hi |= b << (8 * (i - 8));
    }
    return r;
  }
  __device__ __forceinline__ void load16(uint32_t off, uint64_t &lo, uint64_t &hi) const {
    if ((uint64_t)off + 16 <= avail) {
      const Q4 q = *reinterpret_cast<const Q4 *>(base + off);
      lo = (uint64_t)q.w[0] | ((uint64_t)q.w[1] << 32);
      hi = (uint64_t)q.w[2] | ((uint64_t)q.w[3] << 32);
    } else {
      const Pair r = load16_tail(base, avail, off);
      lo = r.lo;
      hi = r.hi;
    }
  }
};

struct DecArgs {
  const uint8_t *payload, *ctx;
  const uint64_t *offsets;
  const uint32_t *nbins_per_stream;
  uint8_t *bins_out;
  uint32_t *status;
  uint64_t S, payload_bytes;
  uint32_t nbins, pitch, nctx;
  uint32_t ctx_init[64];
};

template <int T, int CM>
__global__ void __launch_bounds__(256) k_decode(DecArgs a) {
  using TL = Tile<T>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
  if (tid < 64) {
    tab[tid] = c_lps_rows[tid];
    tab[64 + tid] = c_next.e[tid];
  }
  __syncthreads();
  const uint64_t s0 = ((uint64_t)blockIdx.x * kWavesPerBlock + wib) * 64;
  if (s0 >= a.S) return;

  constexpr uint32_t SB = CM == 0 ? 0u : (CM == 1 ? 4096u : 16384u);
  uint8_t *wbase = smem + kTabBytes + wib * (SB + 2 * T * 64);
  uint8_t *tile_c = wbase + SB, *tile_o = tile_c + T * 64;

  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  const uint32_t nmin = wave_min_u32(n);

  uint32_t packed = a.ctx_init[0];
  if (CM != 0) {
    uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
    const uint32_t ngrp = (a.nctx + 3) / 4;
    for (uint32_t g = 0; g < ngrp; g++) st32[g * 64] = a.ctx_init[g];
  }
  const uint32_t row_tab = c_lps_rows[lane], trans_tab = c_next.e[lane];

  // payload window
  const uint64_t start = a.offsets[s], end = a.offsets[s + 1];
  const uint32_t len = (uint32_t)(end - start);
  const uint64_t abase = start & ~3ull;
  const uint32_t skew = (uint32_t)(start - abase);
  WindowSrc<DevMem> src;
  src.mem.base = a.payload + abase;
  src.mem.avail = a.payload_bytes > abase ? a.payload_bytes - abase : 0;
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);

  const uint32_t ntiles = (a.nbins + T - 1) / T;
  uint4 pc[TL::NCH];
  if (ntiles) TL::load(pc, a.ctx, s0, a.S, a.pitch, 0, lane);
  wave_lds_sync();
  for (uint32_t t = 0; t < ntiles; t++) {
    TL::to_lds(tile_c, pc, lane);
    wave_lds_sync();
    if (t + 1 < ntiles) TL::load(pc, a.ctx, s0, a.S, a.pitch, (t + 1) * T, lane);
    for (int j = 0; j < TL::NCH; j++) {
      const uint32_t i0 = t * T + j * 16;
      uint4 ob = make_uint4(0, 0, 0, 0);
      if (i0 < a.nbins) {
        const uint4 wc = TL::row_chunk(tile_c, lane, j);
        const bool full = i0 + 16 <= nmin;
        const bool hi = wave_any(((wc.x | wc.y | wc.z | wc.w) & 0x80808080u) != 0);
        uint32_t o[4] = {0, 0, 0, 0};
        src.boundary();
        if (full && !hi) {
          if (CM == 0) {
            RegStatesXbar st{&packed, row_tab, trans_tab};
#pragma unroll
            for (int k = 0; k < 16; k++) {
              if ((k & 3) == 0) dec_topup(d, src);
              if (wave_any(d.sh < 8)) {
                if (d.sh < 8) dec_refill(d, src);
              }
              o[k >> 2] |= dec_bin(d, byte_of(wc, k), st) << (8 * (k & 3));
            }
          } else {
            LdsStatesXbar st{wbase + lane * 4, row_tab, trans_tab};
#pragma unroll
            for (int k = 0; k < 16; k++) {
              if ((k & 3) == 0) dec_topup(d, src);
              if (wave_any(d.sh < 8)) {
                if (d.sh < 8) dec_refill(d, src);
              }
              o[k >> 2] |= dec_bin(d, byte_of(wc, k), st) << (8 * (k & 3));
            }
          }
        } else if (full) {
          if (CM == 0) {
            RegStatesXbar st{&packed, row_tab, trans_tab};
#pragma unroll
            for (int k = 0; k < 16; k++) {
              if ((k & 3) == 0) dec_topup(d, src);
              if (wave_any(d.sh < 8)) {
                if (d.sh < 8) dec_refill(d, src);
              }
              o[k >> 2] |= dec_bin_any(d, byte_of(wc, k), st) << (8 * (k & 3));
            }
          } else {
            LdsStatesXbar st{wbase + lane * 4, row_tab, trans_tab};
#pragma unroll
            for (int k = 0; k < 16; k++) {
              if ((k & 3) == 0) dec_topup(d, src);
              if (wave_any(d.sh < 8)) {
                if (d.sh < 8) dec_refill(d, src);
              }
              o[k >> 2] |= dec_bin_any(d, byte_of(wc, k), st) << (8 * (k & 3));
            }
          }
        } else {
#pragma unroll 4
          for (int k = 0; k < 16; k++) {
            if ((k & 3) == 0) dec_topup(d, src);
            if (i0 + k < n) {
              if (d.sh < 8) dec_refill(d, src);
              const uint32_t cx = (word_of(wc, k >> 2) >> (8 * (k & 3))) & 0xFFu;
              uint32_t bin;
              if (CM == 0) {
                RegStatesMem st{&packed, tab};
                bin = dec_bin_any(d, cx, st);
              } else {
                LdsStatesMem st{wbase + lane * 4, tab};
                bin = dec_bin_any(d, cx, st);
              }
              const uint32_t sel = bin << (8 * (k & 3));
              if ((k >> 2) == 0) o[0] |= sel;
              else if ((k >> 2) == 1) o[1] |= sel;
              else if ((k >> 2) == 2) o[2] |= sel;
              else o[3] |= sel;
            }
          }
        }
        ob = make_uint4(o[0], o[1], o[2], o[3]);
      }
      TL::put_row_chunk(tile_o, lane, j, ob);
    }
    wave_lds_sync();
    TL::store(a.bins_out, tile_o, s0, a.S, a.pitch, t * T, lane);
    wave_lds_sync();
  }
  if (valid && a.status) {
    const uint8_t *p = a.payload + start;
    a.status[s] = dec_finish(d, len, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

// ------------------------------------------------------------------------------------------------
// lens -> offsets: three small kernels (1024 lengths per block)
// ------------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256, kScanItems = 4, kScanBlock = kScanThreads * kScanItems;

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

// exclusive scan over the 256 threads of a block; returns this thread's exclusive prefix, *total = block sum
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *total) {
  __shared__ uint64_t wsum[kScanThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t inc = wave_incl_scan_u64(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; i++) {
    if (i < w) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_partial(const uint32_t *lens, uint64_t S, uint64_t *bsum) {
  const uint64_t i0 = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * kScanItems;
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
    if (i0 + k < S) v += lens[i0 + k];
  uint64_t tot;
  block_excl_scan_u64(v, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_top(uint64_t *bsum, uint64_t nblk) {
  uint64_t carry = 0;
  for (uint64_t b0 = 0; b0 < nblk; b0 += kScanThreads) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nblk ? bsum[i] : 0;
    uint64_t tot;
    const uint64_t ex = block_excl_scan_u64(v, &tot);
    if (i < nblk) bsum[i] = carry + ex;
    carry += tot;
  }
}

__global__ void __launch_bounds__(kScanThreads) k_scan_apply(const uint32_t *lens, uint64_t S, const uint64_t *bsum,
                                                             uint64_t *offsets) {
  const uint64_t i0 = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * kScanItems;
  uint32_t l[kScanItems];
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    l[k] = i0 + k < S ? lens[i0 + k] : 0;
    v += l[k];
  }
  uint64_t tot;
  uint64_t run = bsum[blockIdx.x] + block_excl_scan_u64(v, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (i0 + k < S) offsets[i0 + k] = run;
    run += l[k];
    if (i0 + k + 1 == S) offsets[S] = run;
  }
}

// ------------------------------------------------------------------------------------------------
// slots -> tight payload: 16 lanes per stream
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather(const uint8_t *slots, uint64_t slot_stride, const uint32_t *lens,
                                                const uint64_t *offsets, uint64_t S, uint8_t *payload, uint64_t cap) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t s = gid >> 4;
  const uint32_t li = (uint32_t)gid & 15u;
  if (s >= S) return;
  const uint32_t len = lens[s];
  const uint64_t off = offsets[s];
  if (off + len > cap) return;
  const uint8_t *src = slots + s * slot_stride;
  uint8_t *dst = payload + off;
  // head bytes up to the first 4-byte aligned destination address, then aligned dwords, then the tail
  const uint32_t head = min(len, (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
  if (li < head) dst[li] = src[li];
  const uint32_t nd = (len - head) >> 2;
  const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
  uint32_t *dst32 = reinterpret_cast<uint32_t *>(dst + head);
  for (uint32_t j = li; j < nd; j += 16) {
    // source bytes [head + 4j, head + 4j + 4): head < 4, so they straddle source dwords j and j+1
    const uint32_t lo = src32[j], hi = src32[j + 1];
    dst32[j] = head == 0 ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * head));
  }
  const uint32_t done = head + nd * 4;
  if (done + li < len) dst[done + li] = src[done + li];
}

// ------------------------------------------------------------------------------------------------
// synthetic workload generator; identical to oracle/cabac_oracle.c:orc_synth_fill
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__global__ void __launch_bounds__(256) k_synth(uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins, uint32_t pitch,
                                               uint32_t C, int ep_half, uint8_t *bins, uint8_t *ctx) {
  const uint32_t cpr = pitch / 16;  // 16-byte chunks per row
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t r = gid / cpr;
  const uint32_t ch = (uint32_t)(gid % cpr);
  if (r >= S) return;
  const uint32_t s = (uint32_t)(s0 + r);
  const bool pow2 = (C & (C - 1)) == 0;
  uint32_t ob[4] = {0, 0, 0, 0}, oc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t i = ch * 16 + k;
    uint32_t bin = 0, cb = 0;
    if (i < nbins) {
      const uint32_t h = fmix32(seed ^ (s * 0x9E3779B1u) ^ (i * 0x85EBCA77u));
      const uint32_t h2 = fmix32(h + 0x6A09E667u);
      const uint32_t c = pow2 ? (h & (C - 1)) : (((h & 0xFFFFu) * C) >> 16);
      const uint32_t thr = ((2 * c + 1) << 24) / (2 * C);
      bin = (h2 >> 8) < thr ? 1u : 0u;
      cb = c;
      if (ep_half && (h >> 31)) {
        cb = 0xFFu;
        bin = h2 & 1u;
      }
    }
    ob[k >> 2] |= bin << (8 * (k & 3));
    oc[k >> 2] |= cb << (8 * (k & 3));
  }
  *reinterpret_cast<uint4 *>(bins + r * pitch + ch * 16) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
  *reinterpret_cast<uint4 *>(ctx + r * pitch + ch * 16) = make_uint4(oc[0], oc[1], oc[2], oc[3]);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
uint64_t slot_stride_for(uint32_t nbins) {
  const uint64_t bound = (6ull * nbins + 9 + 7) / 8;
  return (bound + 4 + 15) & ~15ull;
}

static int tile_choice() {
  static int t = [] {
    const char *e = getenv("ISSCABAC_TILE");
    const int v = e ? atoi(e) : 64;
    return (v == 32 || v == 64 || v == 128) ? v : 64;
  }();
  return t;
}

template <int T, int CM>
static size_t coder_lds_bytes() {
  const size_t sb = CM == 0 ? 0 : (CM == 1 ? 4096 : 16384);
  return kTabBytes + kWavesPerBlock * (sb + 2 * T * 64);
}

template <int T, int CM>
static hipError_t run_encode(const EncArgs &a, hipStream_t stream) {
  const size_t lds = coder_lds_bytes<T, CM>();
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode<T, CM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  const uint64_t nblk = (a.S + 64 * kWavesPerBlock - 1) / (64 * kWavesPerBlock);
  hipLaunchKernelGGL((k_encode<T, CM>), dim3((unsigned)nblk), dim3(256), lds, stream, a);
  return hipGetLastError();
}

template <int T, int CM>
static hipError_t run_decode(const DecArgs &a, hipStream_t stream) {
  const size_t lds = coder_lds_bytes<T, CM>();
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode<T, CM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  const uint64_t nblk = (a.S + 64 * kWavesPerBlock - 1) / (64 * kWavesPerBlock);
  hipLaunchKernelGGL((k_decode<T, CM>), dim3((unsigned)nblk), dim3(256), lds, stream, a);
  return hipGetLastError();
}

#define ISSCABAC_DISPATCH(FN, ARGS)                                                  \
  do {                                                                               \
    const int cm = sh.nctx <= 4 ? 0 : (sh.nctx <= 64 ? 1 : 2);                       \
    const int t = tile_choice();                                                     \
    if (t == 32) {                                                                   \
      if (cm == 0) return FN<32, 0>(ARGS, stream);                                   \
      if (cm == 1) return FN<32, 1>(ARGS, stream);                                   \
      return FN<32, 2>(ARGS, stream);                                                \
    } else if (t == 128) {                                                           \
      if (cm == 0) return FN<128, 0>(ARGS, stream);                                  \
      if (cm == 1) return FN<128, 1>(ARGS, stream);                                  \
      return FN<128, 2>(ARGS, stream);                                               \
    } else {                                                                         \
      if (cm == 0) return FN<64, 0>(ARGS, stream);                                   \
      if (cm == 1) return FN<64, 1>(ARGS, stream);                                   \
      return FN<64, 2>(ARGS, stream);                                                \
    }                                                                                \
  } while (0)

hipError_t launch_encode(const BatchShape &sh, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                         uint8_t *d_slots, uint64_t slot_stride, uint32_t *d_lens, hipStream_t stream) {
  EncArgs a;
  a.bins = d_bins;
  a.ctx = d_ctx;
  a.nbins_per_stream = d_nbins;
  a.slots = d_slots;
  a.lens = d_lens;
  a.S = sh.S;
  a.slot_stride = slot_stride;
  a.nbins = sh.nbins;
  a.pitch = sh.pitch;
  a.nctx = sh.nctx;
  std::memcpy(a.ctx_init, sh.ctx_init, 256);
  ISSCABAC_DISPATCH(run_encode, a);
}

hipError_t launch_decode(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                         const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out, uint32_t *d_status,
                         hipStream_t stream) {
  DecArgs a;
  a.payload = d_payload;
  a.ctx = d_ctx;
  a.offsets = d_offsets;
  a.nbins_per_stream = d_nbins;
  a.bins_out = d_bins_out;
  a.status = d_status;
  a.S = sh.S;
  a.payload_bytes = payload_bytes;
  a.nbins = sh.nbins;
  a.pitch = sh.pitch;
  a.nctx = sh.nctx;
  std::memcpy(a.ctx_init, sh.ctx_init, 256);
  ISSCABAC_DISPATCH(run_decode, a);
}

uint64_t scan_tmp_bytes(uint64_t S) { return ((S + kScanBlock - 1) / kScanBlock + 1) * sizeof(uint64_t); }

hipError_t launch_scan(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, hipStream_t stream) {
  const uint64_t nblk = (S + kScanBlock - 1) / kScanBlock;
  uint64_t *bsum = static_cast<uint64_t *>(d_tmp);
  hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nblk), dim3(kScanThreads), 0, stream, d_lens, S, bsum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanThreads), 0, stream, bsum, nblk);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblk), dim3(kScanThreads), 0, stream, d_lens, S, bsum, d_offsets);
  return hipGetLastError();
}

hipError_t launch_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, const uint64_t *d_offsets,
                         uint64_t S, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream) {
  const uint64_t nblk = (S * 16 + 255) / 256;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)nblk), dim3(256), 0, stream, d_slots, slot_stride, d_lens, d_offsets, S,
                     d_payload, payload_cap);
  return hipGetLastError();
}

hipError_t launch_synth(uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx, int ep_half,
                        uint8_t *d_bins, uint8_t *d_ctx, hipStream_t stream) {
  const uint64_t total = S * (pitch / 16);
  const uint64_t nblk = (total + 255) / 256;
  hipLaunchKernelGGL(k_synth, dim3((unsigned)nblk), dim3(256), 0, stream, seed, s0, S, nbins, pitch, nctx, ep_half, d_bins,
                     d_ctx);
  return hipGetLastError();
}

}  // namespace isscabac
