// cabac_symbols.hip -- symbol-level batch kernels: (de)binarisation and context selection fused with the coder
// (cabac_symbols.h + cabac_lane.h).  One lane = one stream (one column of symbols); every loop iteration codes one
// bin per lane, whichever symbol that lane is in, so lanes stay busy although symbols binarise to different lengths.
// Used by BASELINE config 1 (cabacDemo: TU, 3 contexts) and config 4 (ISS W/H columns: EG0, 23 contexts).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "cabac_internal.h"
#include "cabac_lane.h"
#include <mutex>
#include "cabac_symbols.h"

namespace isscabac {

__device__ const uint32_t y_lps_rows[64] = {ISSCABAC_LPS_ROWS};
struct SymNextTab {
  uint32_t e[64];
  uint32_t eb[64];  // successor bytes by (packed state, bin): four per dword
  uint32_t ep[64];  // successor bytes by (packed state, path), path 0 = LPS, 1 = MPS
  constexpr SymNextTab() : e{}, eb{}, ep{} {
    constexpr uint8_t trans[64] = {ISSCABAC_TRANS_LPS};
    for (uint32_t s = 0; s < 64; s++) {
      e[s] = next_state_entry(s, trans[s]);
      const uint32_t a = next_state_entry16(2 * s, trans[s]), b = next_state_entry16(2 * s + 1, trans[s]);
      ep[s] = (a >> 8) | (a & 0xFFu) << 8 | (b >> 8) << 16 | (b & 0xFFu) << 24;
      eb[s] = next_state_by_bin(2 * s, 0, trans[s]) | next_state_by_bin(2 * s, 1, trans[s]) << 8 |
              next_state_by_bin(2 * s + 1, 0, trans[s]) << 16 | next_state_by_bin(2 * s + 1, 1, trans[s]) << 24;
    }
  }
};
__device__ const SymNextTab y_next{};

#ifndef ISSCABAC_SYM_PRIO
#define ISSCABAC_SYM_PRIO 1
#endif
constexpr int kSymWaves = 4;
constexpr uint32_t kSymStateBytes = 4096, kSymStageBytes = 1024;
// the code-tree kernels: 32 bytes of output staging and (encoder) a 32-byte ring of input symbols per lane (see kSymPhase),
// and a state image of the batch's own size (256 bytes per four contexts): config 4's 23 contexts leave room for six
// workgroups on a CU
constexpr uint32_t kWalkStageBytes = 2048, kWalkRingBytes = 2048;
__host__ __device__ constexpr uint32_t sym_walk_state_bytes(uint32_t nctx) { return 256u * ((nctx + 3u) / 4u); }
// LDS of a workgroup: [0,256) rangeTabLps rows = [state][column] bytes, [256,512) next-state entries, [512,768) successor
// bytes by (packed state, bin), [768,1024) by (packed state, path), then {bits, len | np << 8} per symbol value
constexpr uint32_t kSymByBin = 512, kSymByPath = 768, kSymBinTab = 1024, kSymTab = 1024 + 256 * 8;
constexpr uint32_t kSymWaveBytes = kSymStateBytes + kSymStageBytes;
// LDS of a workgroup of the code-tree kernels (cabac_symbols.h: SymTrie), read by absolute address:
//   [0, 1024)     the single-lookup step's 8-byte entries by packed state (cabac_lane.h: enc_bin_row_acc / dec_bin_row_q)
//   [1024, 2048)  per symbol byte: its bin string << 3 (encoder: the bin in turn sits at bit 3, where it is 8 * bin)
//                 (decoder: [0, 2048) the 8-byte entries by RAW state byte, cabac_lane.h: dec_bin_out -- it has no bin strings)
//   [2048, ...)   the walk table: one dword per (node, bin, symbol above) at node * 256 + bin * 128 + up * 4 --
//                 bits 0..13  where the walk stands after this bin: (child node, same symbol above), or after a code word's
//                             last bin (root, the symbol just finished above)
//                 bit 14      no code word goes on here (corrupt stream: the general kernel decodes it again)
//                 bit 15      the bin ended a code word, bits 28..31 its symbol
//                 bits 16..27 LDS offset of the context state the NEXT bin is coded with
//   so that per bin the walk, the context selection and the end-of-symbol test are one LDS read and a few bit operations.
constexpr uint32_t kWalkBits = 1024, kWalkTab = 2048, kWalkSlot = 128, kWalkMaxNodes = 32;
constexpr uint32_t kWalkNextMask = 0x3FFFu, kWalkBadBit = 14, kWalkLeafBit = 15, kWalkOffShift = 16, kWalkOffBits = 12, kWalkSymShift = 28;
// (both kernels index the slot by the bin now: the decoder's step hands back `successor | bin`, cabac_lane.h: dec_bin_nx_abs)
__host__ __device__ constexpr uint32_t walk_node_stride() { return 2 * kWalkSlot; }
// the tables, rounded up to 2 KiB: behind them lie the waves' 2 KiB staging / ring regions (LaneBytes32), whose addresses are
// formed by OR-ing into a 2 KiB-aligned base
__host__ __device__ constexpr uint32_t sym_walk_tab_bytes(uint32_t nodes) { return (kWalkTab + nodes * walk_node_stride() + 2047u) & ~2047u; }
// LDS of a code-tree workgroup: tables | staging of waves 0..3 | (encoder) symbol rings of waves 0..3 | state images
__host__ __device__ constexpr uint32_t sym_walk_regions(bool encoder) { return kSymWaves * (kWalkStageBytes + (encoder ? kWalkRingBytes : 0u)); }
__host__ __device__ constexpr uint32_t sym_walk_lds_bytes(uint32_t nodes, uint32_t nctx, bool encoder) {
  return sym_walk_tab_bytes(nodes) + sym_walk_regions(encoder) + kSymWaves * sym_walk_state_bytes(nctx);
}
static_assert(kWalkSlot == (8u << 4) && kWalkSlot == (1u << 7) && kTrieUp * 4 <= kWalkSlot, "the shifts in the kernels");
static_assert(kWalkMaxNodes * 2 * kWalkSlot <= kWalkNextMask + 1u, "walk positions are 14 bits");
static_assert(kWalkStageBytes == 2048 && kWalkRingBytes == 2048, "LaneBytes32 regions are 2 KiB and 2 KiB-aligned");

// states in LDS (same conflict-free layout as the bin kernels), tables in LDS memory (lanes finish at different times)
struct SymStates {
  uint8_t *base;
  const uint32_t *tab;
  __device__ __forceinline__ uint8_t *addr(uint32_t c) const { return base + (((c & ~3u) << 6) | (c & 3u)); }
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return *addr(c); }
  __device__ __forceinline__ void store(uint32_t c, uint32_t q) const { *addr(c) = (uint8_t)q; }
  __device__ __forceinline__ uint32_t load_at(const uint8_t *a) const { return *a; }
  __device__ __forceinline__ void store_at(uint8_t *a, uint32_t q) const { *a = (uint8_t)q; }
  __device__ __forceinline__ uint32_t lps_row(uint32_t q) const { return tab[q >> 2]; }
  __device__ __forceinline__ uint32_t next_entry(uint32_t q) const { return tab[64 + (q >> 2)]; }
  // byte tables by absolute LDS address, as in the bin kernels (no static LDS in these kernels: checked at launch)
  typedef __attribute__((address_space(3))) uint8_t lds_u8;
  __device__ __forceinline__ uint32_t lps_mem(uint32_t q, uint32_t range) const {
    return *reinterpret_cast<const lds_u8 *>((uintptr_t)((q & 0xFCu) | ((range >> 6) & 3u)));
  }
  __device__ __forceinline__ uint32_t next_mem(uint32_t q_or_bin) const {
    return *reinterpret_cast<const lds_u8 *>((uintptr_t)(kSymByBin + q_or_bin));
  }
  __device__ __forceinline__ uint32_t path_index(uint32_t q) const { return kSymByPath + q; }
  __device__ __forceinline__ uint32_t next_path_mem(uint32_t addr) const { return *reinterpret_cast<const lds_u8 *>((uintptr_t)addr); }
};
// Bin policy of enc_bin_at_mem for one bin in a register
struct SymBin {
  uint32_t bin;
  __device__ __forceinline__ uint32_t xor_q(uint32_t q) const { return (bin << 1) ^ q; }
  __device__ __forceinline__ uint32_t or_q(uint32_t q) const { return q | bin; }
};

struct SymSink {
  uint8_t *stage, *slot;
  uint32_t pos;
  __device__ __forceinline__ void put(uint32_t b) {
    stage[pos & 15u] = (uint8_t)b;
    pos++;
    if ((pos & 15u) == 0) *reinterpret_cast<uint4 *>(slot + pos - 16) = *reinterpret_cast<const uint4 *>(stage);
  }
  __device__ __forceinline__ void close() {
    if (pos & 15u) *reinterpret_cast<uint4 *>(slot + (pos & ~15u)) = *reinterpret_cast<const uint4 *>(stage);
  }
};

struct SymEncArgs {
  const uint8_t *sym;
  const uint32_t *nsym_per_stream;
  uint8_t *slots;
  uint32_t *lens, *nbins_out;
  uint32_t *ctx_hist, *ctx_cost;  // optional statistics (include/isscabac.h: isscabac_symbol_stats)
  uint16_t *sym_bits;
  uint32_t *bin_trace;  // optional: one trace record per coded bin (cabac_stream.hip: trace_record), bin_trace_pitch per stream
  uint32_t bin_trace_pitch;
  uint64_t S, slot_stride;
  uint32_t nsym, pitch, nctx;
  SymParams P;
  uint32_t ctx_init[16];
};

__device__ __forceinline__ void sym_init_common(uint8_t *smem, uint32_t *&tab, uint8_t *&wbase, int tid,
                                                uint32_t tab_bytes = kSymTab, uint32_t wave_bytes = kSymWaveBytes) {
  tab = reinterpret_cast<uint32_t *>(smem);
  if (tid < 64) {
    tab[tid] = y_lps_rows[tid];
    tab[64 + tid] = y_next.e[tid];
    tab[128 + tid] = y_next.eb[tid];
    tab[192 + tid] = y_next.ep[tid];
  }
  __syncthreads();
  wbase = smem + tab_bytes + (tid >> 6) * wave_bytes;
}

// LDSCNT: the two per-context counters of the statistics live in the LDS behind the waves' regions, [counter][context][lane]
// 32-bit words, and go out once at the end -- counted in global memory every bin waits for its own read-modify-write
// (config 4 at its real size: 0.93 ms per encode with the statistics against 0.16 without)
template <bool LDSCNT>
__global__ void __launch_bounds__(256) k_encode_symbols(SymEncArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t *tab;
  uint8_t *wbase;
  {  // every symbol value binarised once per workgroup: the per-symbol work in the loop is then one LDS read
    uint2 *bt = reinterpret_cast<uint2 *>(smem + kSymBinTab);
    const BinString g = binarize(min((uint32_t)tid, a.P.nq - 1), a.P);
    bt[tid] = make_uint2(g.bits, g.len | (g.np << 8));
  }
  sym_init_common(smem, tab, wbase, tid);
  const uint2 *bintab = reinterpret_cast<const uint2 *>(smem + kSymBinTab);
  const uint64_t s = ((uint64_t)blockIdx.x * kSymWaves + (tid >> 6)) * 64 + lane;
  const bool valid = s < a.S;
  uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
  for (uint32_t g = 0; g < (a.nctx + 3) / 4; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
  SymStates st{wbase + lane * 4, tab};
  if (!valid) return;  // the only cross-lane operation below is a ballot over the lanes still coding
  uint32_t n_my = a.nsym_per_stream ? min(a.nsym_per_stream[s], a.nsym) : a.nsym;
  const uint8_t *row = a.sym + s * a.pitch;
  const bool want_stats = a.ctx_hist || a.ctx_cost || a.sym_bits || a.bin_trace;
  uint32_t *btrace = a.bin_trace ? a.bin_trace + s * a.bin_trace_pitch : nullptr;
  EncCore e;
  enc_start(e);
  SymSink sink{wbase + kSymStateBytes + lane * 16, a.slots + s * a.slot_stride, 0};
  SymWalk w;
  w.start();
  uint32_t nbins = 0;
  uint4 buf = make_uint4(0, 0, 0, 0);
  // statistics of cabacEncode.m:39-42,60-67 (debugging aid there, optional here): this lane's rows in global memory
  uint32_t *hist = a.ctx_hist ? a.ctx_hist + s * ISSCABAC_STAT_CTX : nullptr;
  uint32_t *cost = a.ctx_cost ? a.ctx_cost + s * ISSCABAC_STAT_CTX : nullptr;
  uint16_t *hbits = a.sym_bits ? a.sym_bits + s * a.pitch : nullptr;
  uint32_t *cnt = nullptr;
  if constexpr (LDSCNT) {
    cnt = reinterpret_cast<uint32_t *>(smem + kSymTab + kSymWaves * kSymWaveBytes + (tid >> 6) * (a.nctx * 512u)) + lane;
    for (uint32_t c = 0; c < 2 * a.nctx; c++) cnt[c * 64] = 0;
  } else {
    for (uint32_t c = 0; c < ISSCABAC_STAT_CTX; c++) {
      if (hist) hist[c] = 0;
      if (cost) cost[c] = 0;
    }
  }
  uint32_t sym_start = 0;
  while (w.idx < n_my) {
    if (w.n == 1) {
      if ((w.idx & 15u) == 0) buf = *reinterpret_cast<const uint4 *>(row + w.idx);
      const uint32_t wi = (w.idx >> 2) & 3u;
      // (two levels of selects: the chained comparison form compiled to nested branches)
      const uint32_t w_lo = (wi & 1u) ? buf.y : buf.x, w_hi = (wi & 1u) ? buf.w : buf.z;
      const uint32_t word = (wi & 2u) ? w_hi : w_lo;
      const uint2 g = bintab[(word >> (8 * (w.idx & 3u))) & 0xFFu];  // (values >= nq cannot be coded: undefined)
      w.cur.bits = g.x;
      w.cur.len = g.y & 0xFFu;
      w.cur.np = g.y >> 8;
      sym_start = sink.pos;
    }
    const uint32_t c = w.ctx(a.P);
    const uint32_t before = sink.pos;  // 8 * pos = the reference's getNumBits()
    const uint32_t q_before = btrace ? st.load(c) : 0u;
    enc_bin_at_mem(e, SymBin{(w.cur.bits >> (w.n - 1)) & 1u}, st.addr(c), st);
    // the record format of the per-bin handle's traces (isscabac_trace_stats reduces both)
    if (btrace)
      btrace[nbins] = c | (((w.cur.bits >> (w.n - 1)) & 1u) << 10) | (1u << 11) | (packed_from_q(q_before) << 16) |
                      (packed_from_q(st.load(c)) << 24);
    // bytes leave at the reference's cadence (a write-out test per bin) when the statistics are wanted -- they are
    // defined through getNumBits() -- and otherwise only when some lane still coding could overflow `low` with its next
    // bins (W < 39 => W <= 62 after four more): the bytes are the same either way
    if (want_stats) enc_flush(e, sink);
    else if (CABAC_WAVE_ANY(e.W >= 39)) enc_flush(e, sink);
    nbins++;
    if constexpr (LDSCNT) {
      if (c < a.nctx) {
        cnt[c * 64]++;
        cnt[(a.nctx + c) * 64] += 8u * (sink.pos - before);
      }
    } else {
      if (hist) hist[c]++;
      if (cost) cost[c] += 8u * (sink.pos - before);
    }
    if (w.n == w.cur.len) {
      if (hbits) hbits[w.idx] = (uint16_t)(8u * (sink.pos - sym_start));
      w.next_symbol(a.P);
    } else {
      w.n++;
    }
  }
  if (hbits)
    for (uint32_t i = n_my; i < a.nsym; i++) hbits[i] = 0;
  if constexpr (LDSCNT) {
    for (uint32_t c = 0; c < ISSCABAC_STAT_CTX; c++) {
      if (hist) hist[c] = c < a.nctx ? cnt[c * 64] : 0u;
      if (cost) cost[c] = c < a.nctx ? cnt[(a.nctx + c) * 64] : 0u;
    }
  }
  enc_finish(e, sink);
  sink.close();
  a.lens[s] = sink.pos;
  if (a.nbins_out) a.nbins_out[s] = nbins;
}

// ---- a whole matrix as ONE stream, encoder: the walk in parallel, the coder on the bin sequence ---------------------------------
// One stream is one lane, and a lone wave issues an instruction every ~2 ns: the fused symbol kernel spends 150 ns per bin on
// the reference's real ISS format (16 268 bins for a 400 x 20 matrix: 2.5 ms).  But binarisation and context selection never
// look across columns (ref: +coder/cabacContextSelection.m:24-67, cabacEncode.m:52), so ONE workgroup walks all columns at
// once -- lane = column: count the bins of its column, an exclusive scan over the columns, then write its bins and context
// indices at their place in the stream's sequence -- and the bin-level encoder codes that sequence as a batch of one stream
// (cabac_pair.hip: 60 ns per bin).  Same bytes: the sequence is the one the fused kernel codes.
struct MatBinsArgs {
  const uint8_t *sym;
  uint8_t *bins, *ctx;
  uint32_t *nbins_out, *nbins_user;  // the stream's bin count for the encoder / for the caller (nullable)
  uint32_t nsym, col_rows;
  SymParams P;  // col_rows = 0: a lane's column is a stream of its own
};
constexpr uint32_t kMatCols = 1024;
__global__ void __launch_bounds__(kMatCols) k_iss_matrix_bins(MatBinsArgs a) {
  __shared__ uint2 bt[256];
  __shared__ uint32_t offs[kMatCols + 1];
  const uint32_t tid = threadIdx.x;
  if (tid < 256) {
    const BinString g = binarize(min(tid, a.P.nq - 1), a.P);
    bt[tid] = make_uint2(g.bits, g.len | (g.np << 8));
  }
  __syncthreads();
  const uint32_t ncols = (a.nsym + a.col_rows - 1) / a.col_rows;
  const uint32_t first = tid * a.col_rows;
  const uint32_t cnt = tid < ncols ? min(a.col_rows, a.nsym - first) : 0u;
  const uint8_t *col = a.sym + first;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < cnt; i++) nb += bt[col[i]].y & 0xFFu;
  offs[tid] = nb;
  __syncthreads();
  if (tid == 0) {  // (at most 1 024 columns: a serial scan is microseconds)
    uint32_t run = 0;
    for (uint32_t c = 0; c < ncols; c++) {
      const uint32_t v = offs[c];
      offs[c] = run;
      run += v;
    }
    offs[kMatCols] = run;
    *a.nbins_out = run;
    if (a.nbins_user) *a.nbins_user = run;
  }
  __syncthreads();
  uint32_t o = offs[tid < ncols ? tid : 0];
  SymWalk w;
  w.start();
  while (w.idx < cnt) {
    if (w.n == 1) {
      const uint2 g = bt[col[w.idx]];
      w.cur.bits = g.x;
      w.cur.len = g.y & 0xFFu;
      w.cur.np = g.y >> 8;
    }
    a.ctx[o] = (uint8_t)w.ctx(a.P);
    a.bins[o] = (uint8_t)((w.cur.bits >> (w.n - 1)) & 1u);
    o++;
    if (w.n == w.cur.len) w.next_symbol(a.P);
    else w.n++;
  }
}

struct SymDevMem {
  const uint8_t *base;
  uint64_t avail;
  struct alignas(4) Q4 { uint32_t w[4]; };
  __device__ __forceinline__ void load16(uint32_t off, uint64_t &lo, uint64_t &hi) const {
    if ((uint64_t)off + 16 <= avail) {
      const Q4 q = *reinterpret_cast<const Q4 *>(base + off);
      lo = (uint64_t)q.w[0] | ((uint64_t)q.w[1] << 32);
      hi = (uint64_t)q.w[2] | ((uint64_t)q.w[3] << 32);
    } else {  // the last few bytes of the whole payload buffer: never read past its end (rolled: rare, keep the loop small)
      lo = hi = 0;
#pragma unroll 1
      for (uint32_t i = 0; i < 16; i++) {
        const uint64_t b = (uint64_t)off + i < avail ? base[off + i] : 0;
        if (i < 8) lo |= b << (8 * i);
        else hi |= b << (8 * (i - 8));
      }
    }
  }
};

struct SymDecArgs {
  const uint8_t *payload;
  const uint64_t *offsets;
  const uint32_t *nsym_per_stream;
  uint8_t *sym_out;
  uint32_t *status;
  // code-tree path: redo[s] != 0 marks a stream the tree kernel could not finish, *redo_any != 0 says there is one.  The
  // tree kernel writes them; the general kernel, launched behind it, then decodes exactly those streams (and returns at once
  // when there are none)
  uint32_t *redo, *redo_any;
  uint32_t out16;  // sym_out is 16-byte aligned: symbols are stored 16 at a time
  uint64_t S, payload_bytes;
  uint32_t nsym, pitch, nctx;
  SymParams P;
  uint32_t ctx_init[16];
};

__global__ void __launch_bounds__(256) k_decode_symbols(SymDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (a.redo_any && *a.redo_any == 0u) return;  // (uniform) the code-tree kernel finished every stream itself
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t *tab;
  uint8_t *wbase;
  sym_init_common(smem, tab, wbase, tid);
  const uint64_t s = ((uint64_t)blockIdx.x * kSymWaves + (tid >> 6)) * 64 + lane;
  const bool valid = s < a.S;
  uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
  for (uint32_t g = 0; g < (a.nctx + 3) / 4; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
  SymStates st{wbase + lane * 4, tab};
  if (!valid) return;
  if (a.redo && a.redo[s] == 0u) return;  // behind the code-tree kernel: only the streams it handed over
  uint32_t n_my = a.nsym_per_stream ? min(a.nsym_per_stream[s], a.nsym) : a.nsym;
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  const uint64_t abase = start & ~3ull;
  const uint32_t skew = (uint32_t)(start - abase);
  WindowSrc<SymDevMem> src;
  src.mem.base = a.payload + abase;
  src.mem.avail = a.payload_bytes > abase ? a.payload_bytes - abase : 0;
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  // a stream that starts with 0xFF (corrupt input only) is decoded in the reference's own formulation, see RefDec
  const bool exact = needs_ref_decoder(d.v >> 24);
  RefReader rd{a.payload + start, len, 0, 0};
  RefDec r;
  if (exact) ref_dec_start(r, rd);
  uint8_t *row = a.sym_out + s * a.pitch;
  uint8_t *stage = wbase + kSymStateBytes + lane * 16;  // this lane's 16 bytes of output staging
  SymWalk w;
  w.start();
  uint32_t it = 0;
  while (w.idx < n_my) {
    const uint32_t c = w.ctx(a.P);
    uint32_t bin;
    if (!exact) {
      if ((it++ & 15u) == 0) src.boundary();
      if (d.sh < 8) dec_refill(d, src);
      const uint32_t pair = dec_bin_at_mem(d, st.addr(c), st);  // bit 1 = mps, bit 0 = MPS path
      bin = ((pair >> 1) ^ ~pair) & 1u;
    } else {
      bin = ref_dec_bin(r, c, st, rd);
    }
    w.cur = append_bin(w.cur, w.n, bin);
    const bool finished = symbol_finished(w.cur, a.P);
    if (finished || w.n == 32) {  // 32 bins without an end: corrupt stream; the symbol is cut there and reads as 0
      const uint8_t sv = finished ? (uint8_t)debinarize(w.cur, a.P) : (uint8_t)0;
      if (a.out16) {  // symbols leave 16 at a time: a byte store per symbol costs the memory pipe a transaction per lane and pass
        stage[w.idx & 15u] = sv;
        if ((w.idx & 15u) == 15u) *reinterpret_cast<uint4 *>(row + (w.idx & ~15u)) = *reinterpret_cast<const uint4 *>(stage);
      } else {
        row[w.idx] = sv;
      }
      w.next_symbol(a.P);
    } else {
      w.n++;
    }
  }
  uint32_t zfrom = n_my;
  if (a.out16 && (n_my & 15u)) {  // the last, incomplete group of 16 (rows are padded to a multiple of 16)
    for (uint32_t i = n_my & 15u; i < 16u; i++) stage[i] = 0;
    *reinterpret_cast<uint4 *>(row + (n_my & ~15u)) = *reinterpret_cast<const uint4 *>(stage);
    zfrom = (n_my + 15u) & ~15u;
  }
  for (uint32_t i = zfrom; i < a.nsym; i++) row[i] = 0;
  if (a.status) {
    const uint8_t *p = a.payload + start;
    a.status[s] = exact ? ref_dec_finish(r, rd, len)
                        : dec_finish(d, src.bpos - 8u * skew, len, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

// ---- the same two kernels on the code tree (SymTrie): walk, context selection and end-of-symbol test are one table read
// per bin instead of ~100 VALU of per-lane string handling.  Small alphabets only (nq <= 16: every reference
// configuration); no statistics (they want the reference's write-out cadence, see k_encode_symbols).
struct SymTrieEncArgs {
  SymEncArgs a;
  SymTrie trie;
};
struct SymTrieDecArgs {
  SymDecArgs a;
  SymTrie trie;
};
__device__ __forceinline__ uint32_t walk_state_off(uint32_t c) { return ((c & ~3u) << 6) | (c & 3u); }
// (all 256 threads; the caller's barrier follows)
__device__ __forceinline__ void sym_walk_to_lds(uint8_t *smem, const SymTrie &T, int tid) {
  uint32_t *wt = reinterpret_cast<uint32_t *>(smem + kWalkTab);
  constexpr uint32_t stride = walk_node_stride(), per_node = stride / 4;
  for (uint32_t i = tid; i < T.nodes * per_node; i += 256) {
    const uint32_t node = i / per_node, bin = (i / 32u) % (per_node / 32u), up = i & 31u;
    uint32_t E = (1u << kWalkBadBit) | (node * stride + min(up, kTrieNoUp) * 4u);  // stays where it is
    const uint32_t nx = T.child[node * 2 + bin];
    if (up < kTrieUp && nx != kTrieNone) {
      const uint32_t leaf = nx >> 7, v = nx & 0x7Fu;
      // the next bin: under the same symbol above inside a code word, at the root under the symbol just finished after one
      const uint32_t c = leaf ? (v < kTrieUp ? T.ctx[v] : 0u) : T.ctx[nx * kTrieUp + up];
      const uint32_t next = leaf ? min(v, kTrieNoUp) * 4u : nx * stride + up * 4u;
      E = next | (leaf << kWalkLeafBit) | (walk_state_off(c) << kWalkOffShift) | (leaf ? (v & 15u) << kWalkSymShift : 0u);
    }
    wt[i] = E;
  }
}
typedef __attribute__((address_space(3))) uint8_t sym_lds_u8;
typedef __attribute__((address_space(3))) uint32_t sym_lds_u32;
typedef __attribute__((address_space(3))) uint64_t sym_lds_u64;
__device__ __forceinline__ uint32_t walk_entry(uint32_t wa) { return *reinterpret_cast<const sym_lds_u32 *>((uintptr_t)(kWalkTab + wa)); }

// context states of the code-tree kernels: addresses are 32-bit LDS addresses, the coder's tables the 8-byte entries at 0
struct WalkStates {
  __device__ __forceinline__ uint32_t load_at(uint32_t a) const { return *reinterpret_cast<const sym_lds_u8 *>((uintptr_t)a); }
  __device__ __forceinline__ void store_at(uint32_t a, uint32_t q) const { *reinterpret_cast<sym_lds_u8 *>((uintptr_t)a) = (uint8_t)q; }
  __device__ __forceinline__ uint64_t entry_mem(uint32_t q) const { return *reinterpret_cast<const sym_lds_u64 *>((uintptr_t)shl16c<2>(q)); }
  // the decoder's entry by raw state byte (bit 0 = a stale bin): 256 entries at LDS 0
  __device__ __forceinline__ uint64_t entry_raw(uint32_t qb) const { return *reinterpret_cast<const sym_lds_u64 *>((uintptr_t)(qb << 3)); }
};
// Bin policy of enc_bin_row_acc for a bin string kept << 3: the bin at bit 3, where it is the shift that picks the successor
struct WalkBin {
  uint32_t b8;  // bits8 & 8
  __device__ __forceinline__ uint32_t lps_mask(uint32_t, uint32_t t) const { return (uint32_t)__builtin_amdgcn_sbfe(t, 16, 1); }
  __device__ __forceinline__ uint32_t next_of(uint32_t hi) const { return hi >> b8; }
};

// enc_pick (cabac_lane.h) on the flag byte of the shifted entry, as in k_encode: the flag byte (0x00 / 0xFF) sign-extended by
// an SDWA move is the select mask
__device__ __forceinline__ void enc_pick(const WalkBin &, uint32_t, uint32_t t, uint32_t lps, uint32_t rmps, uint32_t &r, uint32_t &a) {
  uint32_t m;
  asm("v_mov_b32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(m) : "v"(t));
  r = select_mask(m, lps, rmps);
  a = rmps & m;
  return;
  asm("v_cmp_ne_u32_sdwa vcc, %5, %2 src0_sel:DWORD src1_sel:BYTE_2\n\t"
      "v_cndmask_b32_e32 %0, %4, %3, vcc\n\t"
      "v_cndmask_b32_e32 %1, 0, %4, vcc"
      : "=&v"(r), "=&v"(a)
      : "v"(t), "v"(lps), "v"(rmps), "s"(0u)
      : "vcc");
}

// 32 bytes per lane in 2 KiB of a wave's LDS, laid out like the context states: byte k of lane l at
// (k / 4) * 256 + l * 4 + k % 4, so that whatever k each of the 64 lanes is at, lane l only touches bank l % 32.  (Lane-major
// -- l * 32 + k -- puts sixteen lanes on every bank: the symbol ring read and the symbol store of every bin then cost the
// LDS pipe sixteen passes each.)
struct LaneBytes32 {
  uint32_t base;  // LDS address of the region (a multiple of 2048) + lane * 4
  // m * 65 = m + (m << 6) for m = (k + d) % 32: bits 0..1 and 8..10 of it are m % 4 and m / 4 -- an and, a shift-add and an
  // and-or (the base has zeros in those five bits) instead of shift, two ands and a three-way add
  __device__ __forceinline__ uint32_t addr(uint32_t k, uint32_t d = 0) const { return (__umul24((k + d) & 31u, 65u) & 0x703u) | base; }
  __device__ __forceinline__ uint32_t get(uint32_t k, uint32_t d = 0) const { return *reinterpret_cast<const sym_lds_u8 *>((uintptr_t)addr(k, d)); }
  __device__ __forceinline__ void put(uint32_t k, uint32_t b) const { *reinterpret_cast<sym_lds_u8 *>((uintptr_t)addr(k)) = (uint8_t)b; }
  // bytes k, k + 1 (k even): they share a dword
  __device__ __forceinline__ void put16(uint32_t k, uint32_t x) const {
    *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uintptr_t)addr(k)) = (uint16_t)x;
  }
  // bytes 16 h .. 16 h + 15 (h = 0, 1)
  __device__ __forceinline__ uint4 piece(uint32_t h) const {
    const sym_lds_u32 *p = reinterpret_cast<const sym_lds_u32 *>((uintptr_t)(base + h * 1024u));
    return make_uint4(p[0], p[64], p[128], p[192]);
  }
  __device__ __forceinline__ void put_piece(uint32_t h, uint4 v) const {
    sym_lds_u32 *p = reinterpret_cast<sym_lds_u32 *>((uintptr_t)(base + h * 1024u));
    p[0] = v.x;
    p[64] = v.y;
    p[128] = v.z;
    p[192] = v.w;
  }
};

// Global memory in phases.  Every lane is somewhere else in its column (symbols binarise to different lengths), so a load or
// store issued when a lane needs it lands in some lane's iteration nearly every time, and the wave's in-order vmcnt makes
// everybody wait for it.  The code-tree kernels touch global memory only at phase boundaries, wave-uniformly: a lane's
// next 16 symbols are requested one phase before they are committed to its LDS ring (a lane takes at most one symbol per
// bin: 32 bytes of ring and a request at <= 16 symbols left never run dry, and the symbol after the current one is always
// there to be looked at), and finished 16-byte pieces of output leave at the boundary after they were completed.
constexpr uint32_t kSymPhase = 8;  // bins per lane between two boundaries of the encoder (the decoder: 16, its window's chunk)

// byte sink of the code-tree encoder: 32 bytes of LDS per lane, whole 16-byte pieces leave at phase boundaries
struct SymPhaseSink {
  LaneBytes32 stage;
  uint8_t *slot;
  uint32_t pos, out;  // bytes produced / bytes stored (a multiple of 16)
  __device__ __forceinline__ void store_piece() {
    *reinterpret_cast<uint4 *>(slot + out) = stage.piece((out >> 4) & 1u);
    out += 16;
  }
  __device__ __forceinline__ void put(uint32_t b) {
    if (pos - out >= 32u) store_piece();  // a long run of pending 0xFF bytes resolved at once: rare
    stage.put(pos & 31u, b);
    pos++;
  }
  // two bytes, high byte first; pos is even whenever this is called (16-bit digits come before any single byte)
  __device__ __forceinline__ void put16(uint32_t x) {
    if (pos - out >= 32u) store_piece();
    stage.put16(pos & 31u, __builtin_amdgcn_perm(x, x, 0x0C0C0001u));
    pos += 2;
  }
  __device__ __forceinline__ void boundary() {
    if (pos - out >= 16u) store_piece();
  }
  __device__ __forceinline__ void close() {
    while (out < pos) store_piece();  // (the slot is padded to whole pieces)
  }
};

template <bool COLS>  // the symbol above restarts with every column (ISS selector with col_rows set)
__global__ void __launch_bounds__(256) k_encode_symbols_trie(SymTrieEncArgs ta) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SymEncArgs &a = ta.a;
  const int tid = threadIdx.x, lane = tid & 63;
  {
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    if (tid < 128) {  // packed state: its row, its successors by bin
      tab[2 * tid] = y_lps_rows[tid >> 1];
      const uint32_t sb = y_next.eb[tid >> 1] >> (16 * (tid & 1));
      tab[2 * tid + 1] = enc_entry_hi(tid, sb & 0xFFu, (sb >> 8) & 0xFFu);
    }
    const BinString g = binarize(min((uint32_t)tid, a.P.nq - 1), a.P);  // (values >= nq cannot be coded: they go out as nq - 1)
    reinterpret_cast<uint32_t *>(smem + kWalkBits)[tid] = g.bits << 3;  // (at most 29 bins: sym_trie_build)
    sym_walk_to_lds(smem, ta.trie, tid);
  }
  __syncthreads();
  const uint32_t sb = sym_walk_state_bytes(a.nctx), tabs = sym_walk_tab_bytes(ta.trie.nodes), wv = (uint32_t)tid >> 6;
  uint8_t *wbase = smem + tabs + sym_walk_regions(true) + wv * sb;  // this wave's state image
  const uint64_t s = ((uint64_t)blockIdx.x * kSymWaves + wv) * 64 + lane;
  const bool valid = s < a.S;
  uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
  for (uint32_t g = 0; g < (a.nctx + 3) / 4; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
  if (!valid) return;  // (the only cross-lane operations below are ballots over the lanes still here)
  const uint32_t n_my = a.nsym_per_stream ? min(a.nsym_per_stream[s], a.nsym) : a.nsym;
  const uint8_t *row_in = a.sym + s * a.pitch;
  WalkStates st;
  const uint32_t sbase = (uint32_t)(uintptr_t)(sym_lds_u8 *)(wbase + lane * 4);  // this lane's states
  EncCore e;
  enc_start(e);
  SymPhaseSink sink{LaneBytes32{tabs + wv * kWalkStageBytes + lane * 4}, a.slots + s * a.slot_stride, 0, 0};
  const LaneBytes32 ring{tabs + (kSymWaves + wv) * kWalkStageBytes + lane * 4};  // symbols [fill - 32, fill) of this column
  const uint32_t n_pad = (n_my + 15u) & ~15u;  // rows are padded to whole 16-byte pieces (pitch)
  uint32_t fill = 0;
  uint4 piece = make_uint4(0, 0, 0, 0);
  bool pending = false;
  for (; fill < min(n_pad, 32u); fill += 16)
    ring.put_piece(fill >> 4, *reinterpret_cast<const uint4 *>(row_in + fill));
  const sym_lds_u32 *bt8 = reinterpret_cast<const sym_lds_u32 *>((uintptr_t)kWalkBits);
  const uint32_t wa_top = kTrieNoUp * 4u;                                      // the root, no symbol above
  const uint32_t sa_top = sbase + walk_state_off(ta.trie.ctx[kTrieNoUp]);       // ... and the context of its bin
  uint32_t idx = 0, wa = wa_top, sa = sa_top, row = 0, nbins = 0;
  uint32_t bits8 = n_my ? bt8[ring.get(0)] : 0u;
  // one bin of this lane's current symbol; guarded: the lane may have run out of symbols
  auto one_bin = [&](EncWordAcc &acc, auto guarded) {
    if (decltype(guarded)::value) {
      if (idx >= n_my) return;
      nbins++;
    }
    // the next symbol's string: wanted if this bin ends the current one
    const uint32_t nb8 = bt8[ring.get(idx, 1u)];
    const uint32_t b8 = opaque(bits8 & 8u);                  // 8 * bin (opaque: stays one AND, shared by both uses)
    const uint32_t E = walk_entry(wa + (b8 << 4));           // (kWalkSlot = 8 << 4)
    enc_bin_row_acc(e, acc, false, WalkBin{b8}, sa, st);
    const uint32_t lastm = (uint32_t)__builtin_amdgcn_sbfe(E, kWalkLeafBit, 1);  // ~0: the symbol's last bin
    bits8 = select_mask(lastm, nb8, bits8 >> 1);
    idx -= lastm;
    wa = E & kWalkNextMask;
    sa = sbase + ((E >> kWalkOffShift) & ((1u << kWalkOffBits) - 1u));
    if (COLS) {
      row -= lastm;
      const bool top = row == a.P.col_rows;  // (only ever true right after a last bin)
      row = top ? 0u : row;
      wa = top ? wa_top : wa;
      sa = top ? sa_top : sa;
    }
  };
  while (CABAC_WAVE_ANY(idx < n_my)) {
    // phase boundary (wave-uniform): commit the piece requested a phase ago, request the next, let finished output go
    if (pending) {
      ring.put_piece((fill >> 4) & 1u, piece);
      fill += 16;
      pending = false;
    }
    if (fill < n_pad && fill - idx <= 16u) {
      piece = *reinterpret_cast<const uint4 *>(row_in + fill);
      pending = true;
    }
    sink.boundary();
    // while every lane has more symbols left than a phase can take, nobody needs a guard
    const bool tail = CABAC_WAVE_ANY(idx + kSymPhase > n_my);
    if (ISSCABAC_SYM_PRIO) __builtin_amdgcn_s_setprio(ISSCABAC_SYM_PRIO);  // preferred while it codes bins (see cabac_kernels.hip)
#pragma unroll
    for (uint32_t g = 0; g < kSymPhase / 4; g++) {
      EncWordAcc acc{0u, 0u};
      if (!tail) {
        nbins += 4;
#pragma unroll
        for (int k = 0; k < 4; k++) one_bin(acc, std::false_type{});
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) one_bin(acc, std::true_type{});
      }
      enc_word_close(e, acc);
      if (CABAC_WAVE_ANY(e.W >= 39)) enc_flush16_wave(e, sink);  // bytes are the same at any cadence
    }
    if (ISSCABAC_SYM_PRIO) __builtin_amdgcn_s_setprio(0);
  }
  enc_finish16(e, sink);
  sink.close();
  a.lens[s] = sink.pos;
  if (a.nbins_out) a.nbins_out[s] = nbins;
}

template <bool COLS>
__global__ void __launch_bounds__(256) k_decode_symbols_trie(SymTrieDecArgs ta) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SymDecArgs &a = ta.a;
  const int tid = threadIdx.x, lane = tid & 63;
  {
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    {  // raw state byte tid (bit 0: a stale bin): its row, `successor | bin` by path (cabac_lane.h: dec_entry_hi_out)
      const uint32_t ps = (uint32_t)tid >> 1;
      const uint32_t sp = y_next.ep[ps >> 1] >> (16 * (ps & 1));
      tab[2 * tid] = y_lps_rows[ps >> 1];
      tab[2 * tid + 1] = dec_entry_hi_out(ps, sp & 0xFFu, (sp >> 8) & 0xFFu);
    }
    sym_walk_to_lds(smem, ta.trie, tid);
  }
  __syncthreads();
  const uint32_t sb = sym_walk_state_bytes(a.nctx), tabs = sym_walk_tab_bytes(ta.trie.nodes), wv = (uint32_t)tid >> 6;
  uint8_t *wbase = smem + tabs + sym_walk_regions(false) + wv * sb;  // this wave's state image
  const uint64_t s = ((uint64_t)blockIdx.x * kSymWaves + wv) * 64 + lane;
  const bool valid = s < a.S;
  uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
  for (uint32_t g = 0; g < (a.nctx + 3) / 4; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
  if (!valid) return;
  WalkStates st;
  const uint32_t sbase = (uint32_t)(uintptr_t)(sym_lds_u8 *)(wbase + lane * 4);  // this lane's states
  const uint32_t n_my = a.nsym_per_stream ? min(a.nsym_per_stream[s], a.nsym) : a.nsym;
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  const uint64_t abase = start & ~3ull;
  const uint32_t skew = (uint32_t)(start - abase);
  WindowSrc<SymDevMem> src;
  src.mem.base = a.payload + abase;
  src.mem.avail = a.payload_bytes > abase ? a.payload_bytes - abase : 0;
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  // streams this kernel does not finish go to the general kernel behind it: a first byte of 0xFF (reference-formulation
  // decoder, see RefDec) and bin strings that are no code word of the alphabet -- corrupt input both
  uint32_t bad = needs_ref_decoder(d.v >> 24) ? (1u << kWalkBadBit) : 0u;
  uint8_t *row_out = a.sym_out + s * a.pitch;
  const LaneBytes32 stage{tabs + wv * kWalkStageBytes + lane * 4};  // 32 bytes of output staging (see kSymPhase)
  const uint32_t wa_top = kTrieNoUp * 4u;
  const uint32_t sa_top = sbase + walk_state_off(ta.trie.ctx[kTrieNoUp]);
  uint32_t idx = 0, wa = wa_top, sa = sa_top, row = 0, out = 0;  // out: symbols stored, a multiple of 16
  // the bin loop's core is the absolute form (cabac_lane.h: DecAbs): the step hands back `successor | bin`
  DecAbs da = dec_to_abs(d);
  auto live = [&]() { return idx < n_my && !(bad >> kWalkBadBit & 1u); };
  auto one_bin = [&](auto guarded, auto first) {
    if (decltype(guarded)::value && !live()) return;
    const uint32_t nx = dec_bin_nx_abs<decltype(first)::value>(da, sa, st.load_at(sa), st);
    const uint32_t E = walk_entry(wa | ((nx & 1u) << 7));  // (kWalkSlot = 1 << 7; wa < 128 within its node)
    // the symbol of a code word that ends here -- anything else is overwritten when its code word does end
    stage.put(idx, E >> kWalkSymShift);
    bad |= E;
    idx += (E >> kWalkLeafBit) & 1u;
    wa = E & kWalkNextMask;
    sa = sbase + ((E >> kWalkOffShift) & ((1u << kWalkOffBits) - 1u));
    if (COLS) {
      row += (E >> kWalkLeafBit) & 1u;
      const bool top = row == a.P.col_rows;  // (only ever true right after a code word's last bin)
      row = top ? 0u : row;
      wa = top ? wa_top : wa;
      sa = top ? sa_top : sa;
    }
  };
  while (CABAC_WAVE_ANY(live())) {
    // chunk boundary (wave-uniform): the window moves on to the piece fetched a chunk ago, and finished 16-symbol pieces
    // leave right behind its wait for that piece (vmcnt counts stores too, and in order)
    src.boundary();
    if (idx - out >= 16u) {
      *reinterpret_cast<uint4 *>(row_out + out) = stage.piece((out >> 4) & 1u);
      out += 16;
    }
    // while every lane has more symbols left than a chunk can finish (and none has left the tree), nobody needs a guard
    const bool tail = CABAC_WAVE_ANY(idx + 16u > n_my || (bad >> kWalkBadBit & 1u));
    if (ISSCABAC_SYM_PRIO) __builtin_amdgcn_s_setprio(ISSCABAC_SYM_PRIO);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      // boundary() has just re-based: its 64 bits cover the top-ups (<= 23 bits each) of words 0 and 1; one check then
      // covers words 2 and 3.  One bit-granular top-up per four bins (sh = 23 afterwards); four context bins shift at most
      // 22 bits (tests/lane_harness.cpp: lane_max_shift_sum), so no bin runs short and none is guarded
      if (g == 2) src.ensure(47);
      dec_topup_abs(da, src);
      if (!tail) {
        one_bin(std::false_type{}, std::true_type{});
        one_bin(std::false_type{}, std::false_type{});
        one_bin(std::false_type{}, std::false_type{});
        one_bin(std::false_type{}, std::false_type{});
      } else {
        // (a lane that sat out a bin has shifted less than the bound allows: never short either)
        one_bin(std::true_type{}, std::true_type{});
        one_bin(std::true_type{}, std::false_type{});
        one_bin(std::true_type{}, std::false_type{});
        one_bin(std::true_type{}, std::false_type{});
      }
    }
    if (ISSCABAC_SYM_PRIO) __builtin_amdgcn_s_setprio(0);
  }
  const bool handed_over = (bad >> kWalkBadBit & 1u) != 0;
  a.redo[s] = handed_over ? 1u : 0u;
  if (handed_over) {
    *a.redo_any = 1u;
    return;
  }
  // what is still staged, the last, incomplete group of 16 padded with zeros (rows are padded to a multiple of 16)
  for (uint32_t i = n_my; i < ((n_my + 15u) & ~15u); i++) stage.put(i, 0u);
  for (; out < n_my; out += 16) *reinterpret_cast<uint4 *>(row_out + out) = stage.piece((out >> 4) & 1u);
  for (uint32_t i = (n_my + 15u) & ~15u; i < a.nsym; i++) row_out[i] = 0;
  if (a.status) {
    const uint8_t *p = a.payload + start;
    dec_from_abs(d, da);
    a.status[s] = dec_finish(d, src.bpos - 8u * skew, len, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

// ---- empirical context initialisation (ISS/+coder/cabacInitContextModel.m) -----------------------------------------
// per bin position n = 1..nlbp: 17 counters, then 4 for the two "rest" contexts
enum { ST_A_PRE, ST_Z_PRE, ST_D_C, ST_J_C0, ST_Y0_C, ST_J_C1, ST_Y1_C, ST_D_B, ST_J_B, ST_X1_B, ST_D_S, ST_Z_S, ST_D_CS,
       ST_J_CS0, ST_Y0_CS, ST_J_CS1, ST_Y1_CS, ST_PER_N };
constexpr int kMaxStat = 8 * ST_PER_N + 4;

// One thread per matrix cell, a grid-stride loop of wave-uniform trip count; every counter is a boolean per cell, so a wave
// adds it up with ONE ballot + population count and lane 0 adds that to the workgroup's LDS histogram -- the first version did
// an LDS atomic per lane and counter (64 lanes on one address serialise) and 70 global atomics per 256 cells: 10 ms for 131 M
// cells against 3.3 ms for coding them (profiles/r05/symbol_host_rate_full.txt).  (The ballot's count travels through lane j of a
// register to counter j, see `count` below.)
__global__ void __launch_bounds__(256) k_iss_ctx_stats(const uint8_t *sym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                                       SymParams P, unsigned long long *counts) {
  __shared__ unsigned int hist[kMaxStat];
  __shared__ uint2 bt[256];  // every symbol value binarised once: {bits, len | np << 8}, np = position of the first 0 (or len)
  const int nstat = (int)P.nlbp * ST_PER_N + 4;
  for (int i = threadIdx.x; i < nstat; i += 256) hist[i] = 0;
  {
    const BinString g = binarize(min((uint32_t)threadIdx.x, P.nq - 1), P);
    uint32_t np = first_zero_1based(g.bits, g.len);
    if (!np) np = g.len;
    bt[threadIdx.x] = make_uint2(g.bits, g.len | (np << 8));
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t total = S * nsym, step = (uint64_t)gridDim.x * 256;
  // counter j of the current bin position += number of lanes with cond (all 64 lanes of the wave get here together): the
  // population count of the ballot goes into LANE j of `vec` (v_writelane), and ONE LDS add per bin position, lane j on
  // counter j, moves the 17 counts to the workgroup's histogram -- three instructions per counter, no branch
  uint32_t vec = 0;
#define ST_COUNT(J, COND)                                                                \
  do {                                                                                   \
    const uint32_t c_ = (uint32_t)__popcll(__ballot(COND)); /* wave-uniform */           \
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(c_), "n"(J));                     \
  } while (0)
  uint32_t r_acc[4] = {0, 0, 0, 0};  // the four "rest" counters are sums: kept per lane, reduced once behind the loop
  auto bit = [](uint32_t bits, uint32_t i) { return (bits >> ((i - 1) & 31u)) & 1u; };  // 1-based (evaluated for every cell: the count's condition says whether i is a position)
  for (uint64_t base = (uint64_t)blockIdx.x * 256; base < total; base += step) {  // (base is the same for the whole workgroup)
    const uint64_t gid = base + threadIdx.x;
    const bool in = gid < total;
    uint32_t cb = 0, cl = 1, np = 1, ub = 0, ul = 1, np_up = 0;
    bool nan_up = true;
    if (in) {
      const uint64_t col = gid / nsym;
      const uint32_t d = (uint32_t)(gid - col * nsym);
      const uint8_t *row = sym + col * pitch;
      const uint2 c = bt[row[d]];
      cb = c.x;
      cl = c.y & 0xFFu;
      np = c.y >> 8;
      // the cell above the first row is a NaN in the reference: length 1, np_up1 = 0, equal to neither 0 nor 1
      nan_up = d == 0;
      if (!nan_up) {
        const uint2 u = bt[row[d - 1]];
        ub = u.x;
        ul = u.y & 0xFFu;
        np_up = u.y >> 8;
      }
    }
    for (uint32_t n = 1; n <= P.nlbp; n++) {
      unsigned int *h = hist + (n - 1) * ST_PER_N;
      vec = 0;
      const bool pre = in && n <= np;
      ST_COUNT(ST_A_PRE, pre);
      ST_COUNT(ST_Z_PRE, pre && bit(cb, n) == 0);
      const bool c2 = pre && n <= np_up;  // np_up == 0 on the first row
      const uint32_t y = bit(ub, n), x = bit(cb, n);
      ST_COUNT(ST_D_C, c2);
      ST_COUNT(ST_Y0_C, c2 && y == 0);
      ST_COUNT(ST_Y1_C, c2 && y != 0);
      ST_COUNT(ST_J_C0, c2 && x == 0 && y == 0);
      ST_COUNT(ST_J_C1, c2 && x == 0 && y == 1);
      const bool b2 = in && n + 1 <= np && np_up < n + 1;
      ST_COUNT(ST_D_B, b2);
      ST_COUNT(ST_X1_B, b2 && x == 1);
      ST_COUNT(ST_J_B, b2 && x == 1 && bit(cb, n + 1) == 0);
      const bool s2 = in && n + np <= cl;
      ST_COUNT(ST_D_S, s2);
      ST_COUNT(ST_Z_S, s2 && bit(cb, n + np) == 0);
      const bool cs = s2 && n + np_up <= ul;
      ST_COUNT(ST_D_CS, cs);
      const bool cs2 = cs && !nan_up;
      const uint32_t ys = bit(ub, n + np_up), xs = bit(cb, n + np);
      ST_COUNT(ST_J_CS0, cs2 && xs == 0 && ys == 0);
      ST_COUNT(ST_J_CS1, cs2 && xs == 0 && ys == 1);
      ST_COUNT(ST_Y0_CS, cs2 && y == 0);  // the reference normalises with y(n), not y(n + np_up1) (:96,:108)
      ST_COUNT(ST_Y1_CS, cs2 && y != 0);
      if (lane < ST_PER_N && vec) atomicAdd(&h[lane], vec);
    }
    const uint32_t n = P.nlbp + 1;
    {
      const bool t = in && n <= np;
      const uint32_t cnt = t ? np - n + 1 : 0u, seg = (cb >> (n - 1)) & low_mask(cnt);
      r_acc[0] += cnt;
      r_acc[1] += t ? cnt - (uint32_t)__popc(seg) : 0u;
    }
    {
      const bool t = in && n > np && n <= cl;
      const uint32_t cnt = t ? cl - n + 1 : 0u, seg = (cb >> (n - 1)) & low_mask(cnt);
      r_acc[2] += cnt;
      r_acc[3] += t ? cnt - (uint32_t)__popc(seg) : 0u;
    }
  }
  {  // (a lane sees at most 2^22 / 256 cells of at most 32 bins: far below 2^32)
    unsigned int *r = hist + P.nlbp * ST_PER_N;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t v = r_acc[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
      if (lane == 0 && v) atomicAdd(&r[i], v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nstat; i += 256)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}
#undef ST_COUNT

static uint32_t max_bins_of(const SymParams &P) {
  if (P.nq < 2 || P.nq > 256) return 0;
  uint32_t mx = 0;
  for (uint32_t v = 0; v < P.nq; v++) {
    uint32_t len;
    if (P.method == 0) {
      len = v == P.nq - 1 ? P.nq - 1 : v + 1;
    } else if (P.method == 2) {
      len = (v >> P.k) + 1 + P.k;
    } else if (P.method == 3) {
      len = 32;
    } else {
      const uint32_t t = (v >> P.k) + 1;
      uint32_t n_p = 0;
      while ((t >> n_p) != 0) n_p++;
      len = n_p + P.k + n_p - 1;
    }
    mx = len > mx ? len : mx;
  }
  return mx;
}

}  // namespace isscabac

using namespace isscabac;

static int sym_check(isscabac_engine *e, const char *who, uint64_t S, uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *p,
                     const uint8_t *ctx_init, int nctx, SymParams *P, uint32_t *init_words, uint32_t *maxbins, bool encoding) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!p || !ctx_init) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: null argument", who);
  if (S == 0 || pitch == 0 || (pitch & 15u) || nsym > pitch) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: bad S / nsym / pitch", who);
  if (p->binarization > ISSCABAC_BIN_FL32 || p->selector > 1 || p->k > 4) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: unsupported binarisation / selector", who);
  if (p->binarization > ISSCABAC_BIN_EGK && !encoding)  // ref: cabacDecodeSymbolFinished.m:12-30 knows TU and EG0..2 only
    return engine_fail(e, ISSCABAC_ERR_ARG, "%s: TRk / FL32 bin strings cannot be decoded (the reference has no end-of-symbol rule for them)", who);
  P->method = p->binarization;
  P->k = p->k;
  P->nq = p->nq;
  P->selector = p->selector;
  P->types = p->iss_types;
  P->nlbp = p->nlbp;
  P->col_rows = p->col_rows;
  *maxbins = max_bins_of(*P);
  if (*maxbins == 0 || *maxbins > 32) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: nq = %u gives %u bins per symbol (supported: 1..32)", who, p->nq, *maxbins);
  const int need = p->selector == ISSCABAC_SEL_DEMO ? 3 : (int)(7 * p->nlbp + 2);
  if (p->selector == ISSCABAC_SEL_ISS && (p->nlbp < 1 || p->nlbp > 8)) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: nlbp must be 1..8", who);
  if (nctx < need || nctx > 64) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: this selector uses %d contexts, nctx is %d (max 64)", who, need, nctx);
  uint8_t init[64];
  memset(init, 1, sizeof init);
  for (int c = 0; c < nctx; c++) {
    if ((ctx_init[c] >> 1) > 62) return engine_fail(e, ISSCABAC_ERR_ARG, "%s: ctx_init[%d] has state 63", who, c);
    init[c] = ctx_init[c];
  }
  memcpy(init_words, init, 64);
  return ISSCABAC_OK;
}

static int sym_grow(isscabac_engine *e, void **p, uint64_t *have, uint64_t need) {
  if (*have >= need) return ISSCABAC_OK;
  if (*p) HIP_TRY(e, hipFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(e, hipMalloc(p, need));
  *have = need;
  return ISSCABAC_OK;
}

// counts -> p(0) exactly as cabacInitContextModel.m forms them (joint / norm in double), and uint8(p * 255)
static void finalize_ctx_stats(const SymParams &P, const unsigned long long *cnt, double *p0, uint8_t *bytes) {
  const int N = (int)P.nlbp, total = 7 * N + 2;
  double tmp[7 * 8 + 2];
  for (int i = 0; i < total; i++) tmp[i] = 0.0;
  auto ratio = [](unsigned long long a, unsigned long long b) { return b ? (double)a / (double)b : 0.0; };
  auto cond = [&](unsigned long long joint, unsigned long long den, unsigned long long ycount) {
    const double norm = (den && ycount) ? (double)ycount / (double)den : 1.0;
    return ratio(joint, den) / norm;
  };
  for (int n = 0; n < N; n++) {
    const unsigned long long *h = cnt + n * ST_PER_N;
    tmp[n] = ratio(h[ST_Z_PRE], h[ST_A_PRE]);
    if (P.types & 1u) tmp[N + n] = cond(h[ST_J_C0], h[ST_D_C], h[ST_Y0_C]);
    if (P.types & 2u) tmp[2 * N + n] = cond(h[ST_J_C1], h[ST_D_C], h[ST_Y1_C]);
    if (P.types & 16u) tmp[3 * N + n] = cond(h[ST_J_B], h[ST_D_B], h[ST_X1_B]);
    tmp[4 * N + n] = ratio(h[ST_Z_S], h[ST_D_S]);
    if (P.types & 4u) tmp[5 * N + n] = cond(h[ST_J_CS0], h[ST_D_CS], h[ST_Y0_CS]);
    if (P.types & 8u) tmp[6 * N + n] = cond(h[ST_J_CS1], h[ST_D_CS], h[ST_Y1_CS]);
  }
  const unsigned long long *r = cnt + N * ST_PER_N;
  tmp[7 * N] = ratio(r[1], r[0]);
  tmp[7 * N + 1] = ratio(r[3], r[2]);
  for (int i = 0; i < total; i++) {
    if (p0) p0[i] = tmp[i];
    if (bytes) {
      double x = std::floor(tmp[i] * 255.0 + 0.5);  // MATLAB uint8(): nearest, ties away from zero, saturating
      bytes[i] = (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x));
    }
  }
}

extern "C" {

int isscabac_iss_init_context_model_dev(isscabac_engine *e, const uint8_t *d_sym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                        const isscabac_symbol_params *p, double *p0_out, uint8_t *p0_bytes_out) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!d_sym || !p || S == 0 || nsym == 0 || nsym > pitch) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_iss_init_context_model_dev: bad argument");
  if (p->nlbp < 1 || p->nlbp > 8 || p->binarization > 1 || p->k > 4) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_iss_init_context_model_dev: unsupported parameters");
  SymParams P{p->binarization, p->k, p->nq, 1u, p->iss_types, p->nlbp, 0u};
  const uint32_t mb = max_bins_of(P);
  if (mb == 0 || mb > 32) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_iss_init_context_model_dev: nq = %u not supported", p->nq);
  ENGINE_ENTER(e);
  int rc = sym_grow(e, &e->scan_tmp, &e->scan_tmp_bytes, kMaxStat * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long *d_counts = static_cast<unsigned long long *>(e->scan_tmp);
  HIP_TRY(e, hipMemsetAsync(d_counts, 0, kMaxStat * sizeof(unsigned long long), e->stream));
  // (a workgroup's LDS counters are 32 bits: at most 2^22 cells per workgroup leave a factor of eight below 2^32 for nlbp + 32 bins)
  const uint64_t cells = S * (uint64_t)nsym, want = (cells + 255) / 256, cap = (uint64_t)e->num_cus * 16;
  const uint64_t nblk = std::max<uint64_t>(std::min(want, cap), (cells + (1ull << 22) - 1) >> 22);
  hipLaunchKernelGGL(k_iss_ctx_stats, dim3((unsigned)nblk), dim3(256), 0, e->stream, d_sym, S, nsym, pitch, P, d_counts);
  HIP_TRY(e, hipGetLastError());
  unsigned long long counts[kMaxStat];
  HIP_TRY(e, hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  finalize_ctx_stats(P, counts, p0_out, p0_bytes_out);
  return ISSCABAC_OK;
}

int isscabac_iss_init_context_model(isscabac_engine *e, const uint8_t *sym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                    const isscabac_symbol_params *p, double *p0_out, uint8_t *p0_bytes_out) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!sym || S == 0 || pitch == 0) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_iss_init_context_model: bad argument");
  ENGINE_ENTER(e);
  int rc = sym_grow(e, &e->stage[0], &e->stage_bytes[0], S * pitch);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->stage[0], sym, S * pitch, hipMemcpyHostToDevice, e->stream));
  return isscabac_iss_init_context_model_dev(e, static_cast<uint8_t *>(e->stage[0]), S, nsym, pitch, p, p0_out, p0_bytes_out);
}

uint32_t isscabac_symbol_max_bins(const isscabac_symbol_params *p) {
  if (!p || p->binarization > ISSCABAC_BIN_FL32) return 0;
  SymParams P{p->binarization, p->k, p->nq, p->selector, p->iss_types, p->nlbp, p->col_rows};
  const uint32_t m = max_bins_of(P);
  return m > 32 ? 0 : m;
}

// the code-tree kernels serve small alphabets (cabac_symbols.h: sym_trie_build); ISSCABAC_SYM_NO_TRIE=1 keeps the general
// kernels for A/B runs
static bool sym_use_trie(const SymParams &P, SymTrie *T) {
  static const bool off = [] {
    const char *v = getenv("ISSCABAC_SYM_NO_TRIE");
    return v && v[0] && v[0] != '0';
  }();
  return !off && sym_trie_build(P, T) && T->nodes <= kWalkMaxNodes;  // (walk positions are 14 bits)
}

// the symbol kernels read their tables by absolute LDS address: their dynamic LDS must start at 0
static int sym_check_no_static_lds(isscabac_engine *e, const void *fn) {
  // asked once per kernel and process (a property of the code object): the query costs microseconds that a 0.1 ms call notices
  static std::mutex mu;
  static const void *seen[8];
  static int nseen = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < nseen; i++)
      if (seen[i] == fn) return ISSCABAC_OK;
  }
  hipFuncAttributes fa;
  HIP_TRY(e, hipFuncGetAttributes(&fa, fn));
  if (fa.sharedSizeBytes != 0) return engine_fail(e, ISSCABAC_ERR_HIP, "internal: symbol kernel has static LDS");
  std::lock_guard<std::mutex> g(mu);
  if (nseen < 8) seen[nseen++] = fn;
  return ISSCABAC_OK;
}

int isscabac_encode_symbols_dev(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S, uint32_t nsym,
                                uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                                uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint64_t *total_bytes,
                                uint32_t *d_nbins_out) {
  return isscabac_encode_symbols_stats_dev(e, d_sym, d_nsym, S, nsym, pitch, params, ctx_init, nctx, d_payload, payload_cap,
                                           d_offsets, total_bytes, d_nbins_out, nullptr);
}

// the encode kernel + the offsets scan + the payload gather of one batch, enqueued on the engine's stream (no synchronisation)
// the engine's four device words and pinned size words (cabac_internal.h: flag, h_total), as cabac_capi.cpp sets them up
static int sym_ensure_words(isscabac_engine *e) {
  if (!e->flag) {
    HIP_TRY(e, hipMalloc(reinterpret_cast<void **>(&e->flag), 16));
    HIP_TRY(e, hipMemset(e->flag, 0, 16));
  }
  if (!e->h_total) HIP_TRY(e, hipHostMalloc(reinterpret_cast<void **>(&e->h_total), 64, hipHostMallocMapped));
  if (!e->h_total_dev) {
    void *dv = nullptr;
    HIP_TRY(e, hipHostGetDevicePointer(&dv, e->h_total, 0));
    e->h_total_dev = static_cast<uint64_t *>(dv);
  }
  return ISSCABAC_OK;
}

// (host_word: which of the engine's pinned size words receives offsets[S], written by the scan kernel itself; < 0: none)
static int enc_symbols_enqueue(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S, uint32_t nsym,
                               uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                               uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint32_t *d_nbins_out,
                               const isscabac_symbol_stats *d_stats, int host_word = -1) {
  // a statistics block with nothing asked for is no statistics block: the code-tree kernel serves the call (the host form
  // always hands one over, and ran the general kernel for that reason alone: config 4 at its real size 0.58 ms per encode)
  if (d_stats && !d_stats->ctx_hist && !d_stats->ctx_cost && !d_stats->sym_bits && !d_stats->bin_trace) d_stats = nullptr;
  SymEncArgs a;
  a.ctx_hist = d_stats ? d_stats->ctx_hist : nullptr;
  a.ctx_cost = d_stats ? d_stats->ctx_cost : nullptr;
  a.sym_bits = d_stats ? d_stats->sym_bits : nullptr;
  uint32_t maxbins = 0;
  a.bin_trace = d_stats ? d_stats->bin_trace : nullptr;
  a.bin_trace_pitch = d_stats ? d_stats->bin_trace_pitch : 0;
  int rc = sym_check(e, "isscabac_encode_symbols_dev", S, nsym, pitch, params, ctx_init, nctx, &a.P, a.ctx_init, &maxbins, true);
  if (!rc && a.bin_trace && a.bin_trace_pitch < (uint64_t)nsym * maxbins)
    rc = engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_symbols_dev: bin_trace_pitch (%u) is below nsym x %u bins", a.bin_trace_pitch, maxbins);
  if (rc) return rc;
  if (!d_sym || !d_payload || !d_offsets || (reinterpret_cast<uintptr_t>(d_sym) & 15u))
    return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_symbols_dev: null or misaligned buffer");
  ENGINE_ENTER(e);
  // a whole matrix as one stream: walk the columns in parallel, code the bin sequence with the bin-level encoder (see
  // k_iss_matrix_bins); the statistics want the fused kernel's per-bin cadence and keep it
  const uint64_t mat_bins = (uint64_t)nsym * maxbins, mat_pitch = (mat_bins + 15) & ~15ull;
  if (S == 1 && nsym != 0 && !d_nsym && !d_stats && a.P.col_rows != 0 && a.P.selector == ISSCABAC_SEL_ISS &&
      (nsym + a.P.col_rows - 1) / a.P.col_rows <= kMatCols && mat_pitch <= ISSCABAC_MAX_PITCH && getenv("ISSCABAC_SYM_NO_MATRIX_SPLIT") == nullptr) {
    const uint64_t stride1 = slot_stride_for((uint32_t)mat_pitch);
    if ((rc = sym_grow(e, reinterpret_cast<void **>(&e->slots), &e->slots_bytes, stride1))) return rc;
    if ((rc = sym_grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, sizeof(uint32_t)))) return rc;
    if ((rc = sym_grow(e, &e->scan_tmp, &e->scan_tmp_bytes, scan_tmp_bytes(1)))) return rc;
    if ((rc = sym_grow(e, &e->mat_bins, &e->mat_bins_bytes, 2 * mat_pitch + 16))) return rc;
    uint8_t *mb = static_cast<uint8_t *>(e->mat_bins);
    MatBinsArgs m;
    m.sym = d_sym;
    m.bins = mb;
    m.ctx = mb + mat_pitch;
    m.nbins_out = reinterpret_cast<uint32_t *>(mb + 2 * mat_pitch);
    m.nbins_user = d_nbins_out;
    m.nsym = nsym;
    m.col_rows = a.P.col_rows;
    m.P = a.P;
    m.P.col_rows = 0;
    hipLaunchKernelGGL(k_iss_matrix_bins, dim3(1), dim3(kMatCols), 0, e->stream, m);
    HIP_TRY(e, hipGetLastError());
    BatchShape sh;
    memset(&sh, 0, sizeof sh);
    sh.S = 1;
    sh.nbins = (uint32_t)mat_bins;
    sh.pitch = (uint32_t)mat_pitch;
    sh.nctx = (uint32_t)nctx;
    sh.latency = sh.latency_enc = true;
    sh.tiny = true;
    sh.pair = e->tuning != ISSCABAC_TUNE_LATENCY_SINGLE;
    sh.d_order = nullptr;
    memcpy(sh.ctx_init, a.ctx_init, 64);
    for (int c = 64; c < 256; c++) sh.ctx_init[c] = 1;
    HIP_TRY(e, launch_encode(sh, m.bins, m.ctx, m.nbins_out, e->slots, stride1, e->lens, e->stream));
    if ((rc = sym_ensure_words(e))) return rc;
    HIP_TRY(e, launch_scan_gather(e->slots, stride1, e->lens, 1, d_offsets, e->scan_tmp, d_payload, payload_cap, e->stream, e->flag + 2,
                                  host_word >= 0 ? e->h_total_dev + host_word : nullptr));
    return ISSCABAC_OK;
  }
  const uint64_t stride = slot_stride_for(nsym * maxbins);
  if ((rc = sym_grow(e, reinterpret_cast<void **>(&e->slots), &e->slots_bytes, S * stride))) return rc;
  if ((rc = sym_grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, S * sizeof(uint32_t)))) return rc;
  if ((rc = sym_grow(e, &e->scan_tmp, &e->scan_tmp_bytes, scan_tmp_bytes(S)))) return rc;
  a.sym = d_sym;
  a.nsym_per_stream = d_nsym;
  a.slots = e->slots;
  a.lens = e->lens;
  a.nbins_out = d_nbins_out;
  a.S = S;
  a.slot_stride = stride;
  a.nsym = nsym;
  a.pitch = pitch;
  a.nctx = (uint32_t)nctx;
  const uint64_t nblk = (S + 64 * kSymWaves - 1) / (64 * kSymWaves);
  SymTrieEncArgs ta;
  if (!d_stats && sym_use_trie(a.P, &ta.trie)) {
    ta.a = a;
    const size_t wlds = sym_walk_lds_bytes(ta.trie.nodes, a.nctx, true);
    if (a.P.col_rows != 0 && a.P.selector != 0) {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_encode_symbols_trie<true>)))) return rc;
      hipLaunchKernelGGL(k_encode_symbols_trie<true>, dim3((unsigned)nblk), dim3(256), wlds, e->stream, ta);
    } else {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_encode_symbols_trie<false>)))) return rc;
      hipLaunchKernelGGL(k_encode_symbols_trie<false>, dim3((unsigned)nblk), dim3(256), wlds, e->stream, ta);
    }
  } else {
    // counters in the LDS when some are asked for and the device grants the room (at most 64 contexts: 150 KiB of its 160)
    const size_t base_lds = kSymTab + kSymWaves * kSymWaveBytes, cnt_lds = base_lds + (size_t)kSymWaves * a.nctx * 512u;
    bool lds_cnt = (a.ctx_hist || a.ctx_cost) && a.nctx <= ISSCABAC_STAT_CTX;
    if (lds_cnt && cnt_lds > e->sym_cnt_lds) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode_symbols<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)cnt_lds) == hipSuccess)
        e->sym_cnt_lds = cnt_lds;
      else {
        (void)hipGetLastError();
        lds_cnt = false;
      }
    }
    if (lds_cnt) {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_encode_symbols<true>)))) return rc;
      hipLaunchKernelGGL(k_encode_symbols<true>, dim3((unsigned)nblk), dim3(256), cnt_lds, e->stream, a);
    } else {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_encode_symbols<false>)))) return rc;
      hipLaunchKernelGGL(k_encode_symbols<false>, dim3((unsigned)nblk), dim3(256), base_lds, e->stream, a);
    }
  }
  HIP_TRY(e, hipGetLastError());
  if ((rc = sym_ensure_words(e))) return rc;
  HIP_TRY(e, launch_scan_gather(e->slots, stride, e->lens, S, d_offsets, e->scan_tmp, d_payload, payload_cap, e->stream, e->flag + 2,
                                host_word >= 0 ? e->h_total_dev + host_word : nullptr));
  return ISSCABAC_OK;
}

int isscabac_encode_symbols_stats_dev(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S,
                                      uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                                      const uint8_t *ctx_init, int nctx, uint8_t *d_payload, uint64_t payload_cap,
                                      uint64_t *d_offsets, uint64_t *total_bytes, uint32_t *d_nbins_out,
                                      const isscabac_symbol_stats *d_stats) {
  // the size word comes back through the engine's pinned words, written by the scan kernel (word 1: word 0 may belong to an
  // asynchronous encode that is still outstanding): no copy command, one synchronisation
  const int rc = enc_symbols_enqueue(e, d_sym, d_nsym, S, nsym, pitch, params, ctx_init, nctx, d_payload, payload_cap, d_offsets,
                                     d_nbins_out, d_stats, 1);
  if (rc) return rc;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const uint64_t total = e->h_total[1];
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap)
    return engine_fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_symbols_dev: payload needs %llu bytes, capacity is %llu",
                       (unsigned long long)total, (unsigned long long)payload_cap);
  return ISSCABAC_OK;
}

static int dec_symbols_enqueue(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                               const uint32_t *d_nsym, uint64_t S, uint32_t nsym, uint32_t pitch,
                               const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *d_sym_out,
                               uint32_t *d_status) {
  SymDecArgs a;
  uint32_t maxbins = 0;
  int rc = sym_check(e, "isscabac_decode_symbols_dev", S, nsym, pitch, params, ctx_init, nctx, &a.P, a.ctx_init, &maxbins, false);
  if (rc) return rc;
  if (!d_payload || !d_offsets || !d_sym_out || (reinterpret_cast<uintptr_t>(d_payload) & 3u))
    return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_symbols_dev: null or misaligned buffer");
  ENGINE_ENTER(e);
  a.payload = d_payload;
  a.offsets = d_offsets;
  a.nsym_per_stream = d_nsym;
  a.sym_out = d_sym_out;
  a.status = d_status;
  a.S = S;
  a.payload_bytes = payload_bytes;
  a.nsym = nsym;
  a.pitch = pitch;
  a.nctx = (uint32_t)nctx;
  const size_t lds = kSymTab + kSymWaves * kSymWaveBytes;
  const uint64_t nblk = (S + 64 * kSymWaves - 1) / (64 * kSymWaves);
  a.redo = nullptr;
  a.redo_any = nullptr;
  a.out16 = (reinterpret_cast<uintptr_t>(d_sym_out) & 15u) == 0 ? 1u : 0u;
  SymTrieDecArgs ta;
  if (a.out16 && sym_use_trie(a.P, &ta.trie)) {  // (it stores 16 symbols at a time)
    // the code-tree kernel first; whatever it hands over (corrupt streams only) is decoded by the general kernel behind it,
    // which finds the hand-over marks on the device and returns at once when there are none
    if ((rc = sym_grow(e, reinterpret_cast<void **>(&e->lens), &e->lens_bytes, S * sizeof(uint32_t)))) return rc;
    if ((rc = sym_ensure_words(e))) return rc;  // (cabac_internal.h: four words, zeroed; word 3 is this path's)
    HIP_TRY(e, hipMemsetAsync(e->flag + 3, 0, sizeof(uint32_t), e->stream));
    a.redo = e->lens;
    a.redo_any = e->flag + 3;
    ta.a = a;
    const size_t wlds = sym_walk_lds_bytes(ta.trie.nodes, a.nctx, false);
    if (a.P.col_rows != 0 && a.P.selector != 0) {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_decode_symbols_trie<true>)))) return rc;
      hipLaunchKernelGGL(k_decode_symbols_trie<true>, dim3((unsigned)nblk), dim3(256), wlds, e->stream, ta);
    } else {
      if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_decode_symbols_trie<false>)))) return rc;
      hipLaunchKernelGGL(k_decode_symbols_trie<false>, dim3((unsigned)nblk), dim3(256), wlds, e->stream, ta);
    }
    HIP_TRY(e, hipGetLastError());
  }
  if ((rc = sym_check_no_static_lds(e, reinterpret_cast<const void *>(&k_decode_symbols)))) return rc;
  hipLaunchKernelGGL(k_decode_symbols, dim3((unsigned)nblk), dim3(256), lds, e->stream, a);
  HIP_TRY(e, hipGetLastError());
  return ISSCABAC_OK;
}

// Asynchronous forms, as isscabac_encode_batch_dev_async / isscabac_decode_batch_dev_async: only enqueue on the engine's stream;
// isscabac_sync() waits and reports the payload size and a capacity overflow of the most recent asynchronous encode
int isscabac_encode_symbols_dev_async(isscabac_engine *e, const uint8_t *d_sym, const uint32_t *d_nsym, uint64_t S, uint32_t nsym,
                                      uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                                      uint8_t *d_payload, uint64_t payload_cap, uint64_t *d_offsets, uint32_t *d_nbins_out) {
  const int rc = enc_symbols_enqueue(e, d_sym, d_nsym, S, nsym, pitch, params, ctx_init, nctx, d_payload, payload_cap, d_offsets,
                                     d_nbins_out, nullptr, 0);
  if (rc) return rc;
  e->pending_total = true;
  e->pending_cap = payload_cap;
  return ISSCABAC_OK;
}
int isscabac_decode_symbols_dev_async(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                                      const uint32_t *d_nsym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                      const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *d_sym_out,
                                      uint32_t *d_status) {
  return dec_symbols_enqueue(e, d_payload, payload_bytes, d_offsets, d_nsym, S, nsym, pitch, params, ctx_init, nctx, d_sym_out, d_status);
}

int isscabac_decode_symbols_dev(isscabac_engine *e, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                                const uint32_t *d_nsym, uint64_t S, uint32_t nsym, uint32_t pitch,
                                const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx, uint8_t *d_sym_out,
                                uint32_t *d_status) {
  const int rc = dec_symbols_enqueue(e, d_payload, payload_bytes, d_offsets, d_nsym, S, nsym, pitch, params, ctx_init, nctx, d_sym_out,
                                     d_status);
  if (rc) return rc;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

// ---- small batches through ONE pinned, device-mapped block (config 4 at its real size: 20 columns) -------------------------
// A call of a few kilobytes is launches and host round trips, not coding: five copies and two synchronisations per encode
// (profiles/r04/symbol_host_rate.txt).  Below kSymSmallBytes the host forms copy the caller's arrays into a pinned block
// the kernels read and write DIRECTLY over the link (zero-copy), enqueue their kernels and synchronise once; the results
// are then ordinary host memory.  No hipMemcpy at all on this path.
constexpr uint64_t kSymSmallBytes = 1ull << 20;
static uint8_t *sym_pin_block(isscabac_engine *e, uint64_t bytes) {
  if (e->sym_pin_bytes >= bytes) return e->sym_pin;
  if (e->sym_pin) (void)hipHostFree(e->sym_pin);
  e->sym_pin = nullptr;
  e->sym_pin_bytes = 0;
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocMapped) != hipSuccess) return nullptr;
  e->sym_pin = static_cast<uint8_t *>(p);
  e->sym_pin_bytes = bytes;
  return e->sym_pin;
}
static inline uint64_t up64(uint64_t x) { return (x + 63) & ~63ull; }

int isscabac_encode_symbols(isscabac_engine *e, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S, uint32_t nsym,
                            uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init, int nctx,
                            uint8_t *payload, uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes, uint32_t *nbins_out) {
  return isscabac_encode_symbols_stats(e, sym, nsym_per_stream, S, nsym, pitch, params, ctx_init, nctx, payload, payload_cap,
                                       offsets, total_bytes, nbins_out, nullptr);
}

// Host-buffer forms.  The caller's arrays are pageable; the copies are plain hipMemcpyAsync calls on the engine's stream.  Moving
// them onto the bin-level calls' pinned ring (cabac_internal.h: HostPipe) was tried and measured SLOWER here (config 4 tiled
// x 16 384, 131 MB in / 30 MB out: encode 4.70 against 3.92 ms, decode 4.16 against 3.44; profiles/r04/symbol_host_rate.txt):
// the runtime's own staging of a pageable copy runs at 35-40 GB/s on this system, and a symbol batch has no stream groups
// whose transfers the ring could overlap with the kernels of their neighbours.
int isscabac_encode_symbols_stats(isscabac_engine *e, const uint8_t *sym, const uint32_t *nsym_per_stream, uint64_t S,
                                  uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params, const uint8_t *ctx_init,
                                  int nctx, uint8_t *payload, uint64_t payload_cap, uint64_t *offsets, uint64_t *total_bytes,
                                  uint32_t *nbins_out, const isscabac_symbol_stats *stats) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!sym || !payload || !offsets || !params) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_symbols: null buffer");
  const uint32_t maxbins = isscabac_symbol_max_bins(params);
  if (maxbins == 0 || S == 0 || pitch == 0) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_encode_symbols: unsupported parameters");
  ENGINE_ENTER(e);
  const uint64_t in_bytes = S * pitch, dev_cap = S * isscabac_stream_bound(nsym * maxbins);
  int rc;
  const bool any_stats = stats && (stats->ctx_hist || stats->ctx_cost || stats->sym_bits || stats->bin_trace);
  // (the statistics come back through the block too -- the kernel counts in device memory, a read-modify-write per bin is no
  // work for the link, and ONE copy brings [hist | cost | bits] over; a per-bin trace is a debugging output of any size and
  // takes the general route)
  const uint64_t st_h = S * ISSCABAC_STAT_CTX * sizeof(uint32_t), st_b = S * pitch * sizeof(uint16_t);
  const uint64_t o_nsym = up64(in_bytes), o_off = o_nsym + up64(S * 4), o_nb = o_off + up64((S + 1) * 8), o_pay = o_nb + up64(S * 4),
                 o_hist = o_pay + up64(dev_cap), o_cost = o_hist + (any_stats ? up64(st_h) : 0),
                 o_bits = o_cost + (any_stats ? up64(st_h) : 0), small_total = o_bits + (any_stats ? up64(st_b) : 0);
  if (!(stats && stats->bin_trace) && small_total <= kSymSmallBytes) {
    uint8_t *pin = sym_pin_block(e, kSymSmallBytes);
    if (pin) {
      void *dv = nullptr;
      HIP_TRY(e, hipHostGetDevicePointer(&dv, pin, 0));
      uint8_t *d = static_cast<uint8_t *>(dv);
      memcpy(pin, sym, in_bytes);
      if (nsym_per_stream) memcpy(pin + o_nsym, nsym_per_stream, S * sizeof(uint32_t));
      const uint64_t h2d = nsym_per_stream ? o_off : in_bytes;  // [symbols | counts]: one copy
      if ((rc = sym_grow(e, &e->stage[0], &e->stage_bytes[0], o_off))) return rc;
      uint8_t *din = static_cast<uint8_t *>(e->stage[0]);
      HIP_TRY(e, hipMemcpyAsync(din, pin, h2d, hipMemcpyHostToDevice, e->stream));
      isscabac_symbol_stats pst = {nullptr, nullptr, nullptr, nullptr, 0};
      if (any_stats) {
        if ((rc = sym_grow(e, &e->stage[1], &e->stage_bytes[1], small_total - o_hist))) return rc;
        uint8_t *b = static_cast<uint8_t *>(e->stage[1]);
        if (stats->ctx_hist) pst.ctx_hist = reinterpret_cast<uint32_t *>(b);
        if (stats->ctx_cost) pst.ctx_cost = reinterpret_cast<uint32_t *>(b + (o_cost - o_hist));
        if (stats->sym_bits) pst.sym_bits = reinterpret_cast<uint16_t *>(b + (o_bits - o_hist));
      }
      rc = enc_symbols_enqueue(e, din, nsym_per_stream ? reinterpret_cast<uint32_t *>(din + o_nsym) : nullptr, S, nsym, pitch, params,
                               ctx_init, nctx, d + o_pay, dev_cap, reinterpret_cast<uint64_t *>(d + o_off),
                               reinterpret_cast<uint32_t *>(d + o_nb), any_stats ? &pst : nullptr);
      if (rc) return rc;
      if (any_stats) HIP_TRY(e, hipMemcpyAsync(pin + o_hist, e->stage[1], small_total - o_hist, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      if (pst.ctx_hist) memcpy(stats->ctx_hist, pin + o_hist, st_h);
      if (pst.ctx_cost) memcpy(stats->ctx_cost, pin + o_cost, st_h);
      if (pst.sym_bits) memcpy(stats->sym_bits, pin + o_bits, st_b);
      const uint64_t *off = reinterpret_cast<const uint64_t *>(pin + o_off);
      const uint64_t total = off[S];
      if (total_bytes) *total_bytes = total;
      if (total > payload_cap)
        return engine_fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_symbols: payload needs %llu bytes, capacity is %llu",
                           (unsigned long long)total, (unsigned long long)payload_cap);
      memcpy(payload, pin + o_pay, total);
      memcpy(offsets, off, (S + 1) * sizeof(uint64_t));
      if (nbins_out) memcpy(nbins_out, pin + o_nb, S * sizeof(uint32_t));
      return ISSCABAC_OK;
    }
  }
  if ((rc = sym_grow(e, &e->stage[0], &e->stage_bytes[0], in_bytes))) return rc;
  if ((rc = sym_grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = sym_grow(e, &e->stage[3], &e->stage_bytes[3], dev_cap))) return rc;
  if ((rc = sym_grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = sym_grow(e, &e->stage[5], &e->stage_bytes[5], S * sizeof(uint32_t)))) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->stage[0], sym, in_bytes, hipMemcpyHostToDevice, e->stream));
  if (nsym_per_stream) HIP_TRY(e, hipMemcpyAsync(e->stage[2], nsym_per_stream, S * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  // statistics: one device block behind the staged symbols' own buffer [hist | cost | sym_bits]
  isscabac_symbol_stats dst = {nullptr, nullptr, nullptr, nullptr, 0};
  const uint64_t hbytes = S * ISSCABAC_STAT_CTX * sizeof(uint32_t), sbytes = (S * pitch * sizeof(uint16_t) + 3) & ~3ull;
  const uint64_t tbytes = stats && stats->bin_trace ? S * (uint64_t)stats->bin_trace_pitch * sizeof(uint32_t) : 0;
  if (stats && (stats->ctx_hist || stats->ctx_cost || stats->sym_bits || stats->bin_trace)) {
    if ((rc = sym_grow(e, &e->stage[1], &e->stage_bytes[1], 2 * hbytes + sbytes + tbytes))) return rc;
    uint8_t *b = static_cast<uint8_t *>(e->stage[1]);
    if (stats->ctx_hist) dst.ctx_hist = reinterpret_cast<uint32_t *>(b);
    if (stats->ctx_cost) dst.ctx_cost = reinterpret_cast<uint32_t *>(b + hbytes);
    if (stats->sym_bits) dst.sym_bits = reinterpret_cast<uint16_t *>(b + 2 * hbytes);
    if (stats->bin_trace) {
      dst.bin_trace = reinterpret_cast<uint32_t *>(b + 2 * hbytes + sbytes);
      dst.bin_trace_pitch = stats->bin_trace_pitch;
    }
  }
  uint64_t total = 0;
  rc = isscabac_encode_symbols_stats_dev(e, static_cast<uint8_t *>(e->stage[0]),
                                         nsym_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr, S, nsym, pitch, params,
                                         ctx_init, nctx, static_cast<uint8_t *>(e->stage[3]), dev_cap,
                                         static_cast<uint64_t *>(e->stage[4]), &total, static_cast<uint32_t *>(e->stage[5]), &dst);
  if (rc) return rc;
  if (dst.ctx_hist) HIP_TRY(e, hipMemcpyAsync(stats->ctx_hist, dst.ctx_hist, hbytes, hipMemcpyDeviceToHost, e->stream));
  if (dst.ctx_cost) HIP_TRY(e, hipMemcpyAsync(stats->ctx_cost, dst.ctx_cost, hbytes, hipMemcpyDeviceToHost, e->stream));
  if (dst.sym_bits) HIP_TRY(e, hipMemcpyAsync(stats->sym_bits, dst.sym_bits, S * pitch * sizeof(uint16_t), hipMemcpyDeviceToHost, e->stream));
  if (dst.bin_trace) HIP_TRY(e, hipMemcpyAsync(stats->bin_trace, dst.bin_trace, tbytes, hipMemcpyDeviceToHost, e->stream));
  if (total_bytes) *total_bytes = total;
  if (total > payload_cap)
    return engine_fail(e, ISSCABAC_ERR_CAPACITY, "isscabac_encode_symbols: payload needs %llu bytes, capacity is %llu",
                       (unsigned long long)total, (unsigned long long)payload_cap);
  HIP_TRY(e, hipMemcpyAsync(payload, e->stage[3], total, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipMemcpyAsync(offsets, e->stage[4], (S + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  if (nbins_out) HIP_TRY(e, hipMemcpyAsync(nbins_out, e->stage[5], S * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

int isscabac_decode_symbols(isscabac_engine *e, const uint8_t *payload, const uint64_t *offsets, const uint32_t *nsym_per_stream,
                            uint64_t S, uint32_t nsym, uint32_t pitch, const isscabac_symbol_params *params,
                            const uint8_t *ctx_init, int nctx, uint8_t *sym_out, uint32_t *status) {
  if (!e) return ISSCABAC_ERR_ARG;
  if (!payload || !offsets || !sym_out || !params || S == 0 || pitch == 0) return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_symbols: null buffer");
  for (uint64_t s = 0; s < S; s++)
    if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] > 0xFFFFFFFFull)
      return engine_fail(e, ISSCABAC_ERR_ARG, "isscabac_decode_symbols: offsets not monotone at stream %llu", (unsigned long long)s);
  ENGINE_ENTER(e);
  const uint64_t out_bytes = S * pitch, total = offsets[S];
  int rc;
  {
    const uint64_t o_off = up64(total + 16), o_nsym = o_off + up64((S + 1) * 8), o_out = o_nsym + up64(S * 4), o_st = o_out + up64(out_bytes),
                   small_total = o_st + up64(S * 4);
    uint8_t *pin = small_total <= kSymSmallBytes ? sym_pin_block(e, kSymSmallBytes) : nullptr;
    if (pin) {
      void *dv = nullptr;
      HIP_TRY(e, hipHostGetDevicePointer(&dv, pin, 0));
      uint8_t *d = static_cast<uint8_t *>(dv);
      memcpy(pin, payload, total);
      memcpy(pin + o_off, offsets, (S + 1) * sizeof(uint64_t));
      if (nsym_per_stream) memcpy(pin + o_nsym, nsym_per_stream, S * sizeof(uint32_t));
      if ((rc = sym_grow(e, &e->stage[3], &e->stage_bytes[3], o_out))) return rc;  // [payload | offsets | counts]: one copy
      uint8_t *din = static_cast<uint8_t *>(e->stage[3]);
      HIP_TRY(e, hipMemcpyAsync(din, pin, nsym_per_stream ? o_out : o_nsym, hipMemcpyHostToDevice, e->stream));
      rc = dec_symbols_enqueue(e, din, total, reinterpret_cast<const uint64_t *>(din + o_off),
                               nsym_per_stream ? reinterpret_cast<const uint32_t *>(din + o_nsym) : nullptr, S, nsym, pitch, params,
                               ctx_init, nctx, d + o_out, reinterpret_cast<uint32_t *>(d + o_st));
      if (rc) return rc;
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      memcpy(sym_out, pin + o_out, out_bytes);
      if (status) memcpy(status, pin + o_st, S * sizeof(uint32_t));
      return ISSCABAC_OK;
    }
  }
  if ((rc = sym_grow(e, &e->stage[0], &e->stage_bytes[0], out_bytes))) return rc;
  if ((rc = sym_grow(e, &e->stage[2], &e->stage_bytes[2], S * sizeof(uint32_t)))) return rc;
  if ((rc = sym_grow(e, &e->stage[3], &e->stage_bytes[3], total + 16))) return rc;
  if ((rc = sym_grow(e, &e->stage[4], &e->stage_bytes[4], (S + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = sym_grow(e, &e->stage[5], &e->stage_bytes[5], S * sizeof(uint32_t)))) return rc;
  if (nsym_per_stream) HIP_TRY(e, hipMemcpyAsync(e->stage[2], nsym_per_stream, S * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->stage[3], payload, total, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->stage[4], offsets, (S + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  rc = isscabac_decode_symbols_dev(e, static_cast<uint8_t *>(e->stage[3]), total, static_cast<uint64_t *>(e->stage[4]),
                                   nsym_per_stream ? static_cast<uint32_t *>(e->stage[2]) : nullptr, S, nsym, pitch, params, ctx_init,
                                   nctx, static_cast<uint8_t *>(e->stage[0]), static_cast<uint32_t *>(e->stage[5]));
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(sym_out, e->stage[0], out_bytes, hipMemcpyDeviceToHost, e->stream));
  if (status) HIP_TRY(e, hipMemcpyAsync(status, e->stage[5], S * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ISSCABAC_OK;
}

}  // extern "C"
