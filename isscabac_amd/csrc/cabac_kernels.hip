// cabac_kernels.hip -- HIP kernels of the batched CABAC engine, written for gfx950 (MI355X, wave64).
//
// Mapping (DESIGN.md has the long version):
//   * one wavefront = 64 consecutive streams, one lane = one bitstream.  The serial range / low /
//     renormalisation dependency of CABAC stays inside a lane; lanes never talk during coding.
//   * bins / ctx arrive stream-major (each stream's bins contiguous).  A wave reads 64 bytes of each of
//     its 64 streams with coalesced 16-byte loads (4 lanes cover one stream's 64 contiguous bytes), moves
//     them through LDS 32 bytes per stream at a time with an XOR swizzle, and every lane pulls its own
//     stream's bytes back 16 at a time with conflict-free ds_read_b128.  The next round's global loads
//     are issued while the current one is coded, so HBM latency hides behind >= 32 bins of arithmetic.
//   * per-stream context states: one byte per (stream, ctx) in LDS at  (ctx/4)*256 + lane*4 + ctx%4 ,
//     so every lane always hits its own bank.
//   * rangeTabLps and the successor states are 256-byte LDS tables read by absolute LDS address
//     ([state][column]; [packed state][bin] for the encoder, [packed state][lps] for the decoder): one
//     address instruction per lookup, no extraction.  Chunks with bypass bins use 64-entry dword tables
//     instead: lane i keeps entry i in a VGPR and a lookup is one ds_bpermute_b32; divergent (ragged)
//     sections use an LDS copy of those because bpermute cannot read inactive lanes.
//   * emitted bytes are staged 16 per lane in LDS and go to a per-stream HBM slot of provable worst-case
//     size with 16-byte stores; a scan over the lengths and
//     a gather kernel then produce the tight payload + offsets (the reference's one-file-per-stream
//     content, concatenated).
//   * the decoder fetches its payload in aligned 16-byte pieces, each byte once, 8 bytes or more ahead of use (PieceSrc), and tops
//     its value register up bit-granularly once per four bins.
//   * per-lane global addresses are a wave-uniform base (SGPRs) plus a 32-bit lane offset.
//
// No MFMA anywhere: this is integer / table work bounded by VALU issue and LDS, then HBM.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "cabac_internal.h"
#include "cabac_lane.h"
#include "cabac_wave.h"

namespace isscabac {

// ------------------------------------------------------------------------------------------------
// The latency-oriented encoder's state pass (k_encode<.., LAT, ..>), issued by hand.
// In the encoder the context states evolve from the bins and contexts alone -- no range, no low -- so the state pass
// (state -> successor -> state ...) and the arithmetic are two separate recurrences.  At one or two waves per SIMD the
// state pass is an LDS round trip per bin that nothing hides unless the wave's own arithmetic sits in its shadow, and the
// compiler's scheduler does not put it there (it issues the independent arithmetic BEFORE the lookup the chain waits
// for).  So the LDS operations of the state pass and the wait for them are inline assembly in a fixed order:
//   step j:  wait | successor(j-1) out of entry(j-1) (one byte, zero-padded: it may be the next table index)
//            q(j) = same context ? successor(j-1) : state(j)
//            read entry(j) by q(j) | store successor(j-1) | read state(j+1)
//   then, in C++, the arithmetic of bin j - 1 (cabac_lane.h: enc_lat_arith) -- behind the reads, in front of the next wait.
// entry = the 8 bytes of the single-lookup step (enc_bin_row_acc): the state's rangeTabLps row and its successors by bin.
// A wave's LDS operations complete in order; the compiler's own LDS operations in between only make the waits here (all
// lgkmcnt(0)) longer.  Registers with a read in flight (entry, qld) are named by nothing but the next step's assembly.
// ------------------------------------------------------------------------------------------------
struct EncLatPipe {
  uint32_t at, at_prev;      // LDS address of the state of bin j / bin j - 1
  uint64_t same;             // lane mask: bin j names the context of bin j - 1
  uint32_t qld;              // state(j), read in flight
  uint32_t q_prev;           // q(j - 1)
  uint64_t ent_prev;         // entry(j - 1), read in flight
  uint32_t q_done, row_done; // state and row of the bin whose arithmetic is due

  __device__ __forceinline__ void start(uint32_t sb32, uint32_t z01) {
    asm volatile(
        "v_add_u32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
        "ds_read_u8 %1, %0"
        : "=&v"(at), "=&v"(qld)
        : "v"(sb32), "v"(z01)
        : "memory");
    at_prev = q_prev = q_done = row_done = 0;
    same = 0;
    ent_prev = 0;
  }
  // w8_prev: the word of bin j - 1, << 3 (byte k = 8 * bin k); zz_next: the CtxQuad half that holds bin j + 1
  // dep: a value the arithmetic of bin j - 2 ends in (the range): keeps that arithmetic in front of this step's wait
  template <int J>
  __device__ __forceinline__ void step(uint32_t sb32, uint32_t w8_prev, uint32_t zz_next, uint32_t &dep) {
    uint32_t q, nx = 0, t, at_next = 0, qld_next = 0;
    uint64_t ent, same_next = 0;
    // gfx950: a VALU result written with a destination select (the zero-padded successor byte) may not be read by the
    // very next VALU instruction -- the address of the next state (or a no-op in the last step) sits in between
    if (J == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "v_add_u32_sdwa %2, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                   : "+v"(qld), "+v"(dep), "=&v"(at_next)
                   : "v"(sb32), "v"(zz_next)
                   : "memory");
      q = qld;
    } else {
#define ISSCABAC_LAT_HEAD(SEL, BETWEEN)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
               "v_lshrrev_b32_sdwa %0, %4, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD\n\t"     \
               BETWEEN "\n\t"                                                                                               \
               "v_cndmask_b32_e64 %1, %6, %0, %7"                                                                            \
               : "=&v"(nx), "=&v"(q), "+v"(dep), "=&v"(at_next)                                                              \
               : "v"(w8_prev), "v"((uint32_t)(ent_prev >> 32)), "v"(qld), "s"(same), "v"(sb32), "v"(zz_next)                 \
               : "memory")
#define ISSCABAC_LAT_AT0 "v_add_u32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
#define ISSCABAC_LAT_AT1 "v_add_u32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
      constexpr int kb = (J - 1) & 3;
      if (J == 15) ISSCABAC_LAT_HEAD("BYTE_2", "s_nop 0");
      else if (kb == 0) ISSCABAC_LAT_HEAD("BYTE_0", ISSCABAC_LAT_AT0);  // J = 1, 5, 9, 13: bin J + 1 is an even bin
      else if (kb == 1) ISSCABAC_LAT_HEAD("BYTE_1", ISSCABAC_LAT_AT1);  // J = 2, 6, 10, 14
      else if (kb == 2) ISSCABAC_LAT_HEAD("BYTE_2", ISSCABAC_LAT_AT0);  // J = 3, 7, 11
      else ISSCABAC_LAT_HEAD("BYTE_3", ISSCABAC_LAT_AT1);               // J = 4, 8, 12
#undef ISSCABAC_LAT_HEAD
#undef ISSCABAC_LAT_AT0
#undef ISSCABAC_LAT_AT1
    }
    asm volatile("v_lshlrev_b16_e32 %1, 2, %2\n\t"  // (full rate; the 32-bit shift is half rate: cabac_lane.h shl16)
                 "ds_read_b64 %0, %1"
                 : "=&v"(ent), "=&v"(t)
                 : "v"(q)
                 : "memory");
    static_assert(kTabRowOff == 0, "the (absent) offset in the assembly above");
    if (J > 0) asm volatile("ds_write_b8 %0, %1" : : "v"(at_prev), "v"(nx) : "memory");
    if (J < 15)
      asm volatile("ds_read_u8 %0, %2\n\t"
                   "v_cmp_eq_u32_e64 %1, %2, %3"
                   : "=&v"(qld_next), "=&s"(same_next)
                   : "v"(at_next), "v"(at)
                   : "memory");
    q_done = q_prev;
    row_done = (uint32_t)ent_prev;  // complete: it came with the high word the head has just used
    q_prev = q;
    ent_prev = ent;
    at_prev = at;
    at = at_next;
    qld = qld_next;
    same = same_next;
  }
  // after step 15: the last successor goes to LDS, the last row is complete
  __device__ __forceinline__ void finish(uint32_t w8_last, uint32_t &dep) {
    uint32_t nx;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshrrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "ds_write_b8 %4, %0"
        : "=&v"(nx), "+v"(dep)
        : "v"(w8_last), "v"((uint32_t)(ent_prev >> 32)), "v"(at_prev)
        : "memory");
    q_done = q_prev;
    row_done = (uint32_t)ent_prev;
  }
};


__device__ uint32_t g_kernel_ticket[kPinTickets];  // (pinned_wave_index: workgroups of fewer than four waves)

// LDS bytes of one wave: context states, two half tiles, (encoder) the output staging, (bucketed launches) the wave's stream ids
template <int CM, bool PERM = false>
struct WaveLds {
  static constexpr uint32_t SB = CM == 1 ? 4096u : CM == 3 ? 8192u : 16384u;  // CM 1: <= 64 contexts, 3: <= 128, 2: <= 255
  static constexpr uint32_t TILE = EncTile<CM, PERM>::H * 64;  // (more than 64 contexts: 16 bytes per stream resident, cabac_wave.h)
  static constexpr uint32_t STAGE = 16 * 64;
  static constexpr uint32_t IDS = PERM ? 4 * 64 : 0;
  // the "context" bypass bins read and write in the mixed step (a byte nobody else looks at).  <= 64 contexts (MIXED3): group 31 of the
  // wave's region -- where a bypass byte's state address lands under the step's 5-bit group mask (enc_bin_mixed3_acc) --, which is a
  // hole in the ctx tile (TileT<.., HOLE>: its rows 56..63 sit behind it).  <= 128 contexts: one more row behind the image's last
  // context group.  Not with 255 contexts: there the mixed step does not run and the row would cost a wave per CU
  static constexpr bool MIXED3 = CM == 1;
  static constexpr uint32_t DUMMY = CM != 2 ? 256u : 0u;
  static constexpr uint32_t DUMMY_OFF = 31 * 256;
  static_assert(!MIXED3 || SB + TILE + 56 * 32 == DUMMY_OFF, "the dummy row is the hole of the ctx tile");
  static constexpr uint32_t BYTES = SB + DUMMY + 2 * TILE + STAGE + IDS;
};

// ------------------------------------------------------------------------------------------------
// encode kernel
// ------------------------------------------------------------------------------------------------


// LAT: the latency-oriented instance for launches that leave fewer than four waves on a SIMD (see enc_bin_lat_acc):
// the same bytes, three more VALU instructions per bin, one LDS round trip per bin instead of three behind each other
// PERM: a length-bucketed launch (cabac_order.hip): the wave's 64 streams are order[s0 .. s0 + 63]
template <int CM, bool LAT, bool PERM>
__global__ void __launch_bounds__(256, 4) k_encode(EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
  for (int ps = tid; ps < 128; ps += blockDim.x) {  // packed state ps = q >> 1: its row, its successors by bin
    const uint32_t sb = c_next.eb[ps >> 1] >> (16 * (ps & 1));
    // state 63 is no context's state: its entries are the all-zero ones bypass lanes look up (enc_bin_mixed_acc)
    tab[kTabRowOff / 4 + 2 * ps] = ps >= 126 ? 0u : c_lps_rows[ps >> 1];
    // ... and packed state 127 (the byte 0xFE) is the bypass entry of the step on a dummy context that holds it (enc_bin_mixed3_acc)
    tab[kTabRowOff / 4 + 2 * ps + 1] = ps == 127 ? kEncBypassHi : ps == 126 ? 0u : enc_entry_hi(ps, sb & 0xFFu, (sb >> 8) & 0xFFu);
  }
  const uint32_t wpb = CM == 1 ? (uint32_t)enc_waves_per_block(1) : a.wpb;  // (> 64 contexts: chosen at launch, waves_per_block_for; a kernel
                                                                            // argument: derived from blockDim.x the compiler took the wave's first stream -- and every
                                                                            // address built on it -- for a per-lane value)
  int wave = tid >> 6;
  if (CM != 1 && wpb < 4) {  // four waves have started, wpb stay: the ones a per-CU ticket names (cabac_wave.h: pinned_wave_index)
    wave = pinned_wave_index(reinterpret_cast<uint32_t *>(smem + kTabBytes), g_kernel_ticket, tid, (int)wpb, true);
    if (wave < 0) return;
  } else {
    __syncthreads();
  }
  const int wib = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform: keeps s0 and everything derived from it in SGPRs
  const uint64_t s0 = ((uint64_t)blockIdx.x * wpb + wib) * 64;
  if (s0 >= a.S) return;

  using WL = WaveLds<CM, PERM>;
  const uint32_t sbytes = CM == 2 ? a.img_bytes : WL::SB;  // the wave's state image
  const uint32_t ctx_lim = min(sbytes >> 6, 255u), ctx_kadd = (256u - ctx_lim) * 0x01010101u;
  uint8_t *wbase = smem + kTabBytes + wib * (WL::BYTES - WL::SB + sbytes);
  // (MIXED3: states | bins tile | ctx tile with the dummy row in its hole | staging; else: states | dummy row | tiles | staging)
  uint8_t *tile_b = wbase + sbytes + (WL::MIXED3 ? 0u : WL::DUMMY), *tile_c = tile_b + WL::TILE, *stage = tile_c + WL::TILE + (WL::MIXED3 ? WL::DUMMY : 0u);

  const bool valid = s0 + lane < a.S;
  const uint64_t slot = valid ? s0 + lane : a.S - 1;       // position in the launch
  const uint64_t s = PERM ? a.order[slot] : slot;           // the stream coded here
  using BTile = EncTile<CM, PERM>;  // the tiles of the two input arrays, bins and ctx: the same load rounds
  using CTile = EncTile<CM, PERM, WL::MIXED3>;
  uint32_t ridb[BTile::NLD] = {}, ridc[CTile::NLD] = {};
  if (PERM) {  // the ids of the rows this lane fetches pieces of, through LDS
    uint32_t *wids = reinterpret_cast<uint32_t *>(stage + WL::STAGE);
    wids[lane] = (uint32_t)s;
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < BTile::NLD; k++) ridb[k] = wids[BTile::row_of(k, lane)];
#pragma unroll
    for (int k = 0; k < CTile::NLD; k++) ridc[k] = wids[CTile::row_of(k, lane)];
  }
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  const uint32_t nmax = wave_max_u32(n), nmin = wave_min_u32(n);

  // context states
  // doubled states q = m_ucState << 1, four per dword
  {
    uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
    // every group of the wave's state image, not only ceil(nctx / 4): a ctx byte in [nctx, image size) passes
    // is_ctx_byte() and must find the reference's default ContextModel() (mps 1, state 0: the C ABI pads ctx_init
    // with it), never a stale LDS byte -- a stale 0xFC..0xFF would be state 63, whose LPS of 2 shifts 7 bits per bin
    // and overruns the slot bound of 6
    const uint32_t ngrp = sbytes / 256;
    for (uint32_t g = 0; g < ngrp; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
    if (WL::MIXED3) st32[WL::DUMMY_OFF / 4] = 0xFEFEFEFEu;  // the bypass bins' dummy context holds the bypass state, and keeps it
    else if (WL::DUMMY) st32[ngrp * 64] = 0;  // ... an even byte, made the bypass state by what the step ORs into it
  }
  LdsStates stx{wbase + lane * 4};

  EncCore e;
  enc_start(e);
  SlotSink sink;
  sink.stage = stage + lane * 16;
  sink.slots = a.slots + (PERM ? s : s0) * a.slot_stride;
  sink.mine = PERM ? 0u : (uint32_t)(s - s0) * (uint32_t)a.slot_stride;
  sink.pos = 0;
  sink.on = valid;

  // A load round is TM bytes of every row of both arrays, coded in parts of PB bytes (what is resident in LDS); the next round is requested when the round's last
  // part has gone to LDS (the staging registers are free again)
  constexpr int TM = BTile::T, PB = BTile::H, NPM = TM / PB;
  const uint32_t nrounds = (nmax + TM - 1) / TM;
  const uint32_t nrows = (uint32_t)min((uint64_t)64, a.S - s0);
  const uint8_t *bins_w = a.bins + s0 * a.pitch, *ctx_w = a.ctx + s0 * a.pitch;
  uint4 pb0[BTile::NLD], pc0[CTile::NLD];
  auto fetch_into = [&](uint4 (&pb)[BTile::NLD], uint4 (&pc)[CTile::NLD], uint32_t off) {
    if (PERM) BTile::template load_rows<kEncNT>(pb, a.bins, ridb, a.pitch, off, lane);
    else BTile::template load<kEncNT>(pb, bins_w, nrows, a.pitch, off, lane);
    if (PERM) CTile::template load_rows<kEncNT>(pc, a.ctx, ridc, a.pitch, off, lane);
    else CTile::template load<kEncNT>(pc, ctx_w, nrows, a.pitch, off, lane);
  };
  if (nrounds) fetch_into(pb0, pc0, 0);
  wave_lds_sync();
  // one round out of the staging registers (pb, pc)
  auto round = [&](uint4 (&pb)[BTile::NLD], uint4 (&pc)[CTile::NLD], uint32_t t) {
#pragma unroll
    for (int h = 0; h < NPM; h++) {
      const uint32_t pos = t * TM + h * PB;
      if (pos >= nmax) return;
      BTile::half_to_lds(tile_b, pb, h, lane);
      CTile::half_to_lds(tile_c, pc, h, lane);
      wave_lds_sync();
      if (h == NPM - 1 && pos + PB < nmax) fetch_into(pb, pc, pos + PB);
      for (int j = 0; j < PB / 16; j++) {
        const uint32_t i0 = pos + j * 16;
        if (i0 >= nmax) break;
        const uint4 wb = BTile::row_chunk(tile_b, lane, j);
        const uint4 wc = CTile::row_chunk(tile_c, lane, j);
        const bool full = i0 + 16 <= nmin;
        const bool hi = chunk_not_plain<CM>(wc, ctx_kadd);
        if (ISSCABAC_ENC_PRIO) __builtin_amdgcn_s_setprio(ISSCABAC_ENC_PRIO);
        if (LAT && full && !hi) {
          // every lane codes 16 context bins, the state pass issued by hand (EncLatPipe): the lookup of bin j goes out as
          // soon as the successor of bin j - 1 is back, and the range / low arithmetic of bin j - 1 runs behind it
          EncWordAcc acc;
          EncLatPipe p;
          const uint32_t sb32 = stx.base32();
          CtxQuad z = stx.prep4(wc.x);
          p.start(sb32, z.z01);
#pragma unroll
          for (int k = 0; k < 16; k++) {
            if ((k & 3) == 3 && k < 15) z = stx.prep4(word_of(wc, (k + 1) >> 2));
            const uint32_t zz = ((k + 1) & 2) ? z.z23 : z.z01;
            const uint32_t w8p = k > 0 ? word_of(wb, (k - 1) >> 2) << 3 : 0u;
            switch (k) {  // k is a constant once the loop is unrolled
#define ISSCABAC_LAT_STEP(K) case K: p.template step<K>(sb32, w8p, zz, e.range); break;
              ISSCABAC_LAT_STEP(0) ISSCABAC_LAT_STEP(1) ISSCABAC_LAT_STEP(2) ISSCABAC_LAT_STEP(3)
              ISSCABAC_LAT_STEP(4) ISSCABAC_LAT_STEP(5) ISSCABAC_LAT_STEP(6) ISSCABAC_LAT_STEP(7)
              ISSCABAC_LAT_STEP(8) ISSCABAC_LAT_STEP(9) ISSCABAC_LAT_STEP(10) ISSCABAC_LAT_STEP(11)
              ISSCABAC_LAT_STEP(12) ISSCABAC_LAT_STEP(13) ISSCABAC_LAT_STEP(14) ISSCABAC_LAT_STEP(15)
#undef ISSCABAC_LAT_STEP
            }
            if (k > 0) {
              const uint32_t v1 = word_of(wb, (k - 1) >> 2), v2 = v1 << 1;
              with_bin_of_word(k - 1, v2, v1, [&](auto bin) { enc_lat_arith(e, acc, ((k - 1) & 3) == 0, bin, p.q_done, p.row_done); return 0; });
              if (((k - 1) & 3) == 3) {
                enc_word_close(e, acc);
                if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
              }
            }
          }
          p.finish(wb.w << 3, e.range);
          {
            const uint32_t v1 = wb.w, v2 = v1 << 1;
            enc_lat_arith(e, acc, false, BinOfWord<3>{v2, v1, 0u}, p.q_done, p.row_done);
            enc_word_close(e, acc);
            if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
          }
        } else if (full && !hi) {
          // every lane codes 16 context bins: convergent
          EncWordAcc acc;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const CtxQuad z = stx.prep4(word_of(wc, k >> 2));  // once per word: the compiler keeps it across the four bins
            const uint32_t w1 = word_of(wb, k >> 2), w2 = w1 << 1, w8 = w1 << 3;
            // `low` takes the word's four additions at once (EncWordAcc): one 64-bit shift + add per word, not per bin
            if ((k & 3) == 0) enc_bin_row_acc(e, acc, true, BinOfWord<0>{w2, w1, w8}, stx.addr_of(z, 0), stx);
            else if ((k & 3) == 1) enc_bin_row_acc(e, acc, false, BinOfWord<1>{w2, w1, w8}, stx.addr_of(z, 1), stx);
            else if ((k & 3) == 2) enc_bin_row_acc(e, acc, false, BinOfWord<2>{w2, w1, w8}, stx.addr_of(z, 2), stx);
            else {
              enc_bin_row_acc(e, acc, false, BinOfWord<3>{w2, w1, w8}, stx.addr_of(z, 3), stx);
              enc_word_close(e, acc);
            }
            // digits leave `low` only when some lane could overflow it within the next four bins (W < 39 here =>
            // W <= 62 four bins later): with ~0.8 bit per bin that is every ~11 bins, and then every lane that has a
            // digit ready (W >= 29) emits it (8-bin cadence measured 3% slower, 2-bin guard with W >= 51 slower still)
            if ((k & 3) == 3 && wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
          }
        } else if (WL::MIXED3 && full &&
                   !wave_any((not_ctx64_nor_bypass(wc.x) | not_ctx64_nor_bypass(wc.y) | not_ctx64_nor_bypass(wc.z) | not_ctx64_nor_bypass(wc.w)) != 0)) {
          // contexts below 64 and bypass bins only (config 5's chunks): the step on a dummy context that holds the bypass state
          // (enc_bin_mixed3_acc) -- under the 5-bit group mask a bypass byte's state address is the dummy row
          EncWordAcc acc;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const uint32_t cw = word_of(wc, k >> 2);
            const uint32_t w1 = word_of(wb, k >> 2), w2 = w1 << 1;
            const uint32_t cwm = (cw >> 7) & 0x01010101u, wbm = w1 & cwm;  // per word: the bypass bits as bytes of 0 / 1, the bypass bins of value 1
            const CtxQuad z = stx.prep4(cw & 0x7F7F7F7Fu);
            with_bin_of_word(k, w2, w1, [&](auto bin) { enc_bin_mixed3_acc(e, acc, (k & 3) == 0, bin, cwm, wbm, stx.addr_of(z, k & 3), stx); return 0; });
            if ((k & 3) == 3) {
              enc_word_close(e, acc);
              if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
            }
          }
        } else if (CM == 3 && full && !wave_any((odd_high_bytes(wc.x) | odd_high_bytes(wc.y) | odd_high_bytes(wc.z) | odd_high_bytes(wc.w)) != 0)) {
          // context-coded and bypass bins only (config 5's chunks): one select-based step per bin for every lane, no lane
          // branches, `low` touched once per word (enc_bin_mixed_acc).  Bit 7 of a ctx byte says "bypass" here -- batches
          // of more than 128 contexts never come this way --, and the state address of a bypass lane is made valid by
          // masking the byte down to a context of the image (its state is written back unchanged).
          constexpr uint32_t kIn = CM == 1 ? 0x3F3F3F3Fu : 0x7F7F7F7Fu;
          EncWordAcc acc;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const uint32_t cw = word_of(wc, k >> 2);
            const uint32_t w1 = word_of(wb, k >> 2), w2 = w1 << 1;
            // per word: the bypass bits as bytes of 0 / 1, of 0 / 0xFE (or-ed into the state: 0xFE is the all-zero table entry),
            // and the bypass bins of value 1; a bypass bin's state address is the dummy context behind the image
            constexpr uint32_t kDummy = 128u * 0x01010101u;
            const uint32_t cwm = (cw >> 7) & 0x01010101u, mbfe = cwm * 0xFEu;
            const CtxQuad z = stx.prep4(select_mask(mbfe | cwm, kDummy, cw & kIn));
            const uint32_t wbm = w1 & cwm;
            with_bin_of_word(k, w2, w1, [&](auto bin) { enc_bin_mixed2_acc(e, acc, (k & 3) == 0, bin, cwm, mbfe, wbm, stx.addr_of(z, k & 3), stx); return 0; });
            if ((k & 3) == 3) {
              enc_word_close(e, acc);
              if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
            }
          }
        } else {
          // fixed-probability bins, contexts >= 128 or streams that end inside this chunk.  The single-lookup
          // bin step (enc_bin_at_row) works under any exec mask, so every lane simply takes its own branch: context-coded bin, bypass bin,
          // or nothing once its stream is over (all sides run when a wave mixes them: ~29 VALU per bin against ~37 for
          // one combined, select-based step).
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const CtxQuad z = stx.prep4(word_of(wc, k >> 2));
            const uint32_t w1 = word_of(wb, k >> 2), w2 = w1 << 1;
            if (full || i0 + k < n) {
              const uint32_t cb = byte_of(wc, k);
              if (is_ctx_byte<CM>(cb, ctx_lim)) {
                const uint32_t w8 = w1 << 3;
                if ((k & 3) == 0) enc_bin_at_row(e, BinOfWord<0>{w2, w1, w8}, stx.addr_of(z, 0), stx);
                else if ((k & 3) == 1) enc_bin_at_row(e, BinOfWord<1>{w2, w1, w8}, stx.addr_of(z, 1), stx);
                else if ((k & 3) == 2) enc_bin_at_row(e, BinOfWord<2>{w2, w1, w8}, stx.addr_of(z, 2), stx);
                else enc_bin_at_row(e, BinOfWord<3>{w2, w1, w8}, stx.addr_of(z, 3), stx);
              } else if (cb == kBypass || cb == kProbBase + 50u) {
                enc_bin_ep(e, (w1 >> (8 * (k & 3))) & 1u);
              } else if (is_prob_ctx_byte<CM>(cb) && prob_valid(cb - kProbBase)) {
                enc_bin_prob(e, (w1 >> (8 * (k & 3))) & 1u, cb - kProbBase, c_prob_rows[prob_row_index(cb - kProbBase)]);
              }
            }
            if ((k & 3) == 3 && wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
          }
        }
        if (ISSCABAC_ENC_PRIO) __builtin_amdgcn_s_setprio(0);
      }
      wave_lds_sync();
    }
  };
  for (uint32_t t = 0; t < nrounds; t++) round(pb0, pc0, t);
  enc_flush16(e, sink);
  enc_finish16(e, sink);
  sink.close();
  if (valid) a.lens[s] = sink.bytes();
}

// ------------------------------------------------------------------------------------------------
// decode kernel
// ------------------------------------------------------------------------------------------------
// the decode kernel's input tile (its one input array: ctx): 128-byte row segments (32 registers); the latency-oriented instances
// keep 64-byte segments (16)
template <bool LAT, bool PERM>
using DecTile = TileT<LAT ? 4 : 8>;
template <int CM, bool PERM>
struct DecLds {
  static constexpr uint32_t SB = WaveLds<CM, PERM>::SB;
  static constexpr uint32_t TILE = 32 * 64;  // the ctx half tile (TileT::H bytes per stream)
  static constexpr uint32_t IDS = PERM ? 4 * 64 : 0;
  // <= 64 contexts: the mixed step's dummy row (cabac_lane.h: dec_bin_mixed2_nx_abs) is group 31 of the wave's region -- where a
  // bypass byte's state address lands under the step's 5-bit group mask --, i.e. the 256 bytes chunk 7 of the output staging
  // would take: that chunk moves up behind it (OutBufT<true>)
  static constexpr bool MIXED2 = CM == 1;
  typedef OutBufT<MIXED2> OBuf;
  static constexpr uint32_t DUMMY_OFF = 31 * 256;
  static_assert(!MIXED2 || SB + TILE + 7 * 256 == DUMMY_OFF, "the dummy row is the staging buffer's chunk-7 row");
  static constexpr uint32_t BYTES = SB + TILE + OBuf::BYTES + IDS;
};

template <int CM, bool LAT, bool PERM>
__global__ void __launch_bounds__(256, 4) k_decode(DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
  for (int qb = tid; qb < 256; qb += blockDim.x) {  // raw state byte qb: packed state qb >> 1, bit 0 is a stale bin
    const uint32_t ps = (uint32_t)qb >> 1;
    const uint32_t sp = c_next.ep[ps >> 1] >> (16 * (ps & 1));  // byte 0: successor on the LPS path, byte 1: on the MPS path
    // raw bytes 254 / 255 (state 63, no context's state): the bypass entry of the mixed step on a dummy context
    tab[2 * qb] = qb >= (int)kDecBypassRaw ? 0u : c_lps_rows[ps >> 1];
    tab[2 * qb + 1] = qb >= (int)kDecBypassRaw ? kDecBypassHi : dec_entry_hi_out(ps, sp & 0xFFu, (sp >> 8) & 0xFFu);
  }
  const uint32_t wpb = CM == 1 ? (uint32_t)dec_waves_per_block(1) : a.wpb;
  int wave = tid >> 6;
  if (CM != 1 && wpb < 4) {  // (as in k_encode)
    wave = pinned_wave_index(reinterpret_cast<uint32_t *>(smem + kDecTabBytes), g_kernel_ticket, tid, (int)wpb, true);
    if (wave < 0) return;
  } else {
    __syncthreads();
  }
  const int wib = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform: keeps s0 and everything derived from it in SGPRs
  const uint64_t s0 = ((uint64_t)blockIdx.x * wpb + wib) * 64;
  if (s0 >= a.S) return;

  using DL = DecLds<CM, PERM>;
  using DTile = DecTile<LAT, PERM>;
  static_assert(DTile::H * 64 == DL::TILE, "half tile");
  const uint32_t sbytes = CM == 2 ? a.img_bytes : DL::SB;  // the wave's state image
  const uint32_t ctx_lim = min(sbytes >> 6, 255u), ctx_kadd = (256u - ctx_lim) * 0x01010101u;
  uint8_t *wbase = smem + kDecTabBytes + wib * (DL::BYTES - DL::SB + sbytes);
  uint8_t *tile_c = wbase + sbytes;
  using OBuf = typename DL::OBuf;
  uint32_t *obuf = reinterpret_cast<uint32_t *>(tile_c + DL::TILE);
  const uint32_t *wids = reinterpret_cast<const uint32_t *>(tile_c + DL::TILE + OBuf::BYTES);  // PERM: the wave's stream ids

  const bool valid = s0 + lane < a.S;
  const uint64_t slot = valid ? s0 + lane : a.S - 1;       // position in the launch
  const uint64_t s = PERM ? a.order[slot] : slot;           // the stream decoded here
  if (PERM) {
    const_cast<uint32_t *>(wids)[lane] = (uint32_t)s;
    wave_lds_sync();
  }
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  // wave-uniform, and known to the compiler as such: chunk classification is scalar compares and branches
  const uint32_t nmin = __builtin_amdgcn_readfirstlane(wave_min_u32(n)), nmax = __builtin_amdgcn_readfirstlane(wave_max_u32(n));

  // doubled states q = m_ucState << 1, four per dword
  {
    uint32_t *st32 = reinterpret_cast<uint32_t *>(wbase) + lane;
    // every group of the wave's state image, not only ceil(nctx / 4): a ctx byte in [nctx, image size) passes
    // is_ctx_byte() and must find the reference's default ContextModel() (mps 1, state 0: the C ABI pads ctx_init
    // with it), never a stale LDS byte -- a stale 0xFC..0xFF would be state 63, whose LPS of 2 shifts 7 bits per bin
    // and overruns the slot bound of 6
    const uint32_t ngrp = sbytes / 256;
    for (uint32_t g = 0; g < ngrp; g++) st32[g * 64] = (a.ctx_init[g] & 0x7F7F7F7Fu) << 1;
    if (DL::MIXED2) st32[DL::DUMMY_OFF / 4] = kDecBypassRaw * 0x01010101u;  // the bypass bins' dummy context: stays a bypass byte (0xFE | bin)
  }
  LdsStates stx{wbase + lane * 4};

  // payload window
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  // Throughput instances: aligned, non-overlapping 16-byte pieces (PieceSrc: every payload byte requested once -- FETCH_SIZE of
  // k_decode 6.45 -> 3.52 GB per config-3 launch, 2.796 -> 2.783 ms; a piece never straddles two 128-byte lines).  The latency-oriented
  // instances keep the window's own piece (WindowSrc: fewer instructions at the chunk boundary -- 0.516 against 0.528 ms at 128 Ki
  // streams, where the payload lines stay in the L2 anyway); profiles/r06/f_ab_pieces.txt
  constexpr bool kPieces = CM == 1 && !LAT;  // (the instances of more than 64 contexts keep WindowSrc too: they are not the ones whose
                                             // traffic matters, and at their 128 registers PieceSrc's seam code spills)
  constexpr uint64_t kPieceAlign = kPieces ? 16 : 4;
  const uint64_t abase = start & ~(kPieceAlign - 1);
  const uint32_t skew = (uint32_t)(start - abase);
  typename std::conditional<kPieces, PieceSrc<DevMem>, WindowSrc<DevMem>>::type src;
  src.mem.payload = a.payload;
  src.mem.total = a.payload_bytes;
  src.mem.abase = abase;
  src.mem.lim = a.payload_bytes >= abase + 16 ? (int32_t)min(a.payload_bytes - 16 - abase, (uint64_t)0x7FFFFFFF) : -1;
  src.len = skew + len;
  src.start(skew);
  src.boundary();
  DecCore d;
  dec_start(d, src);
  // v now holds the stream's first four bytes.  A first byte of 0xFF (corrupt input only) makes the reference's value
  // register overflow: those streams are decoded again, in the reference's formulation, by k_decode_exact
  if (valid && needs_ref_decoder(d.v >> 24) && a.exact_flag) *a.exact_flag = 1u;
  // the loop's core is the absolute form (cabac_lane.h: DecAbs, R = range << sh); the general path converts for its chunk
  DecAbs da = dec_to_abs(d);

  const uint32_t nrows = (uint32_t)min((uint64_t)64, a.S - s0);
  const uint8_t *ctx_w = a.ctx + s0 * a.pitch;
  uint8_t *out_w = a.bins_out + s0 * a.pitch;
  // 128 bins of every stream leave the staging buffer as full 128-byte row segments: lane l of pass k takes chunk l % 8 of
  // row 8 k + l / 8.  Ordinarily -- all 64 rows exist, the block lies inside the row -- the eight stores are issued by
  // hand: one wave-uniform base per pass (SGPRs) + one 32-bit lane offset for all of them (the compiler hoists eight 64-bit
  // lane addresses out of the bin loop otherwise: 16 registers) and no per-store exec masks.
  auto store_out = [&](uint32_t off) {
    if (!PERM && nrows == 64 && off + 128 <= a.pitch) {
      // (formed here, behind an opaque lane id, so that they are not kept in registers across the bin loop)
      const uint32_t ln = opaque((uint32_t)lane);
      const uint32_t st_voff = (ln >> 3) * a.pitch + (ln & 7) * 16;
      const uint32_t st_lds = (uint32_t)(uintptr_t)(LdsStates::lds_u8 *)reinterpret_cast<uint8_t *>(obuf) + 4 * OBuf::idx(ln & 7, ln >> 3);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        // idx(c, 8 k + i) = idx(c, i) ^ 8 k
        const uint4 v = OBuf::unpack(*reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)(st_lds ^ (32u * k))));
        const u32x4 w = {v.x, v.y, v.z, v.w};
        uint8_t *base = out_w + ((uint64_t)(8 * k) * a.pitch + off);
        // (s_nop 1: a store of more than 64 bits must not be followed within two wait states by a VALU write of its data -- the
        // compiler keeps that wait state for its own stores, it cannot see this one)
        if (kDecNTst) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(st_voff), "v"(w), "s"(base) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(st_voff), "v"(w), "s"(base) : "memory");
      }
      return;
    }
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
      const uint32_t row = k * 8 + (lane >> 3), c = lane & 7;
      const uint4 v = OBuf::unpack(obuf[OBuf::idx(c, row)]);
      const uint32_t boff = off + c * 16;
      if (row < nrows && boff < a.pitch) {
        // 32-bit offset from the wave's first row (see Tile::load), or the row of stream wids[row]
        uint8_t *p = PERM ? a.bins_out + ((uint64_t)wids[row] * a.pitch + boff) : out_w + (row * a.pitch + boff);
        if (kDecNT) nt_store16(p, v);
        else *reinterpret_cast<uint4 *>(p) = v;
      }
    }
  };

  const uint32_t nrounds = (a.nbins + DTile::T - 1) / DTile::T;
  uint4 pc[DTile::NLD];
  auto fetch = [&](uint32_t off) {
    if (PERM) DTile::template load_rows_ids<kDecNT>(pc, a.ctx, wids, a.pitch, off, lane);
    else DTile::template load<kDecNT>(pc, ctx_w, nrows, a.pitch, off, lane);
  };
  if (nrounds) fetch(0);
  wave_lds_sync();
  bool have_out = false;
  uint32_t out_off = 0;
  for (uint32_t t = 0; t < nrounds; t++) {
#pragma unroll
    for (int h = 0; h < DTile::NP; h++) {
      DTile::half_to_lds(tile_c, pc, h, lane);
      wave_lds_sync();
      for (int j = 0; j < 2; j++) {
        const uint32_t i0 = t * DTile::T + h * DTile::H + j * 16;
        const uint32_t cidx = (i0 >> 4) & 7u;  // chunk inside the 128-bin output block
        uint32_t od = 0;  // the chunk's 16 bins packed as they come (OutBuf: bit g of byte b = bin 4 g + b)
        const bool live = i0 < nmax;  // past the longest stream of the wave: zeros
        if (live) src.boundary();
        if (j == 0) {
          // global stores and loads go out right AFTER the window boundary: vmcnt is in-order and counts stores, so
          // anything issued just before the boundary's wait for the payload piece would be waited for as well
          if (have_out) {
            store_out(out_off);
            wave_lds_sync();  // the staging buffer is rewritten from here on
            have_out = false;
          }
          if (h == DTile::NP - 1 && t + 1 < nrounds && (t + 1) * DTile::T < nmax) {  // nothing to decode past the wave's longest stream
            fetch((t + 1) * DTile::T);
          }
        }
        if (live) {
        const uint4 wc = DTile::row_chunk(tile_c, lane, j);
        const bool full = i0 + 16 <= nmin;
        const bool hi = chunk_not_plain<CM>(wc, ctx_kadd);
        // One bit-granular top-up per four bins (sh = 23 afterwards).  Four context or bypass bins shift at most 22 bits
        // (tests/lane_harness.cpp: lane_max_shift_sum), so the plain and mixed words never run short and carry no guard;
        // the window is re-based at the chunk boundary and, wave-uniformly, whenever a lane has fewer than the 24 bits a
        // top-up can take.  oa / ob collect the bins of a word as bytes (cabac_lane.h: dec_bin_out).
        uint32_t oa = 0, ob = 0;
        if (ISSCABAC_DEC_PRIO) __builtin_amdgcn_s_setprio(ISSCABAC_DEC_PRIO);
        if (full && !hi) {
          // every lane decodes 16 context bins
          // latency-oriented instance: a bin's entry is read right behind the decision of the bin before it (its state may
          // be that bin's successor: forwarded in a register for a repeated context, else loaded two bins ahead), and the
          // rest of that bin -- state store, next state load, output byte, the positions of the new range -- is issued in
          // the read's shadow (the scheduling barriers keep the compiler from moving it in front of the read)
          uint32_t at_cur = 0, at_nxt = 0, qld_nxt = 0;
          // lane mask: the next bin names this bin's context -- compared in the shadow and kept in SGPRs (inline assembly: the
          // compiler sinks a C++ comparison to its use, behind the decision)
          uint64_t same_nxt = 0;
          uint32_t psel = 0, psh = 0;  // the next bin's positions (abs_positions of the range this bin left)
          uint64_t ent = 0;
          CtxQuad zn = stx.prep4(wc.x);  // the quad of the word bin i + 2 lies in
          if (LAT) {
            at_cur = stx.addr32_of(zn, 0);
            at_nxt = stx.addr32_of(zn, 1);
            const uint32_t q0 = stx.load_at(at_cur);
            qld_nxt = stx.load_at(at_nxt);
            asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(same_nxt) : "v"(at_nxt), "v"(at_cur));
            ent = stx.entry_raw(q0);
          }
#pragma unroll
          for (int g = 0; g < 4; g++) {
            // boundary() has just re-based: its 64 bits cover the top-ups (<= 23 bits each) of words 0 and 1; one check
            // then covers words 2 and 3
            if (g == 2) src.ensure(47);
            dec_topup_abs(da, src);
            if (LAT) {
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
                  if (((i + 2) & 3) == 0) zn = stx.prep4(word_of(wc, ((i + 2) >> 2) & 3));
                  at_n2 = stx.addr32_of(zn, (i + 2) & 3);
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
            } else {
              const CtxQuad z = stx.prep4(word_of(wc, g));
              const uint32_t a0 = stx.addr32_of(z, 0), a1 = stx.addr32_of(z, 1), a2 = stx.addr32_of(z, 2), a3 = stx.addr32_of(z, 3);
              dec_bin_out_abs<0, true>(da, a0, stx.load_at(a0), stx, oa, ob);
              dec_bin_out_abs<1, false>(da, a1, stx.load_at(a1), stx, oa, ob);
              dec_bin_out_abs<2, false>(da, a2, stx.load_at(a2), stx, oa, ob);
              dec_bin_out_abs<3, false>(da, a3, stx.load_at(a3), stx, oa, ob);
            }
            od |= (out_word(oa, ob) & 0x01010101u) << g;
          }
        } else if (DL::MIXED2 && full &&
                   !wave_any((not_ctx64_nor_bypass(wc.x) | not_ctx64_nor_bypass(wc.y) | not_ctx64_nor_bypass(wc.z) | not_ctx64_nor_bypass(wc.w)) != 0)) {
          // contexts below 64 and bypass bins only (config 5's chunks): the step on a dummy context (dec_bin_mixed2_nx_abs) -- under
          // the 5-bit group mask a bypass byte's state address is the dummy row, whose byte finds the table's bypass entry
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (g == 2) src.ensure(47);
            const uint32_t cw = word_of(wc, g);
            dec_topup_abs(da, src);
            const CtxQuad zs = stx.prep4(cw & 0x7F7F7F7Fu);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const uint32_t ak = stx.addr32_of(zs, k);
              const uint32_t nx = k == 0 ? dec_bin_mixed2_nx_abs<true>(da, ak, stx.load_at(ak), stx) : dec_bin_mixed2_nx_abs<false>(da, ak, stx.load_at(ak), stx);
              if (k == 0) put_out_byte<0>(oa, ob, nx);
              else if (k == 1) put_out_byte<1>(oa, ob, nx);
              else if (k == 2) put_out_byte<2>(oa, ob, nx);
              else put_out_byte<3>(oa, ob, nx);
            }
            od |= (out_word(oa, ob) & 0x01010101u) << g;
          }
        } else if (CM == 3 && full &&
                   !wave_any((odd_high_bytes(wc.x) | odd_high_bytes(wc.y) | odd_high_bytes(wc.z) | odd_high_bytes(wc.w)) != 0)) {
          // (<= 128 contexts: a bypass byte's masked address is context 127, a real one) context-coded and bypass bins only: one
          // select-based step for every lane (dec_bin_mixed_nx), see k_encode
          constexpr uint32_t kIn = 0x7F7F7F7Fu;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (g == 2) src.ensure(47);
            const uint32_t cw = word_of(wc, g);
            dec_topup_abs(da, src);
            const CtxQuad zs = stx.prep4(cw & kIn);
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
          // fixed-probability bins, contexts >= 128 beside bypass bins, or streams that end inside this chunk: every lane
          // takes its own branch under the exec mask -- context bin, bypass bin, fixed-probability bin, or nothing (its
          // output reads 0) once its stream is over.  Three bins after a top-up cannot run short (<= 6 bits each); the
          // fourth is guarded, wave-uniformly.
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
                if (is_ctx_byte<CM>(cb, ctx_lim)) {
                  const uint32_t ak = stx.addr32_of(z, k);
                  w = dec_bin_nx(d, ak, stx.load_at(ak), stx);
                } else if (cb == kBypass || cb == kProbBase + 50u) {
                  w = dec_bin_ep(d);
                } else if (is_prob_ctx_byte<CM>(cb) && prob_valid(cb - kProbBase)) {
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
        obuf[OBuf::idx(cidx, lane)] = od;
      }
      wave_lds_sync();
    }
    if (((t + 1) * DTile::T) % 128 == 0 || t + 1 == nrounds) {  // a 128-bin output block is complete (or the stream ends)
      have_out = true;
      out_off = (t * DTile::T) & ~127u;
      if (((t + 1) * DTile::T) % 128 != 0) {
        // the stream ends inside the block: its upper half was never written
        for (uint32_t c = 4; c < 8; c++) obuf[OBuf::idx(c, lane)] = 0;
        wave_lds_sync();
      }
    }
  }
  if (have_out) store_out(out_off);
  if (valid && a.status) {
    uint64_t first = a.offsets[s];  // re-read: keeping it live through the loop costs two registers
    (void)clip_stream(first, first, a.payload_bytes);
    const uint32_t sk = (uint32_t)first & (uint32_t)(kPieceAlign - 1);
    const uint8_t *p = a.payload + first;
    dec_from_abs(d, da);
    a.status[s] = dec_finish(d, src.bpos - 8u * sk, src.len - sk, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

// ------------------------------------------------------------------------------------------------
// Streams that start with 0xFF (see cabac_lane.h: RefDec): one thread per stream, the reference's own formulation,
// context states in a private array, tables from memory.  Slow, and only ever runs on corrupt input.
// ------------------------------------------------------------------------------------------------
struct PrivateStates {
  uint8_t *cm;
  __device__ uint32_t load(uint32_t c) const { return cm[c]; }
  __device__ void store(uint32_t c, uint32_t q) const { cm[c] = (uint8_t)q; }
  __device__ uint32_t lps_row(uint32_t q) const { return c_lps_rows[q >> 2]; }
  __device__ uint32_t next_entry(uint32_t q) const { return c_next.e[q >> 2]; }
};

__global__ void __launch_bounds__(64) k_decode_exact(DecArgs a, uint32_t *next_flag) {
  // (the flag word of the NEXT decode call on this engine, nobody's during this one: zeroed here instead of by a memset per call)
  if (next_flag && blockIdx.x == 0 && threadIdx.x == 0) *next_flag = 0u;
  if (a.exact_flag && *a.exact_flag == 0u) return;  // k_decode saw no such stream: the usual case, a few microseconds
  const uint64_t s = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  if (len != 0 && !needs_ref_decoder(a.payload[start])) return;  // (an empty stream reads 0xFF at EOF)
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  uint8_t cm[256];
  for (uint32_t c = 0; c < 256; c++) cm[c] = (uint8_t)(((a.ctx_init[(c >> 2) & 63] >> (8 * (c & 3))) & 0x7Fu) << 1);
  PrivateStates st{cm};
  RefReader rd{a.payload + start, len, 0, 0};
  RefDec r;
  ref_dec_start(r, rd);
  const uint8_t *ctx = a.ctx + s * a.pitch;
  uint8_t *out = a.bins_out + s * a.pitch;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = ctx[i];
    const bool prob_ok = a.nctx <= 128u && is_prob_byte(c);  // what the batch kernels take as a fixed-probability bin
    if (c == kBypass || (prob_ok && c == kProbBase + 50u)) out[i] = (uint8_t)ref_dec_bin_ep(r, rd);
    else if (prob_ok) out[i] = prob_valid(c - kProbBase) ? (uint8_t)ref_dec_bin_prob(r, c - kProbBase, c_prob_rows[prob_row_index(c - kProbBase)], rd) : 0;
    else out[i] = (uint8_t)ref_dec_bin(r, c, st, rd);
  }
  const uint32_t status = ref_dec_finish(r, rd, len);
  if (a.status) a.status[s] = status;
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
                                                             uint64_t *offsets, uint64_t *host_total) {
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
    if (i0 + k + 1 == S) {
      offsets[S] = run;
      if (host_total) *host_total = run;  // (a pinned, device-mapped word: the caller reads it after its synchronisation, no copy)
    }
  }
}



// ------------------------------------------------------------------------------------------------
// slots -> tight payload: 16 lanes per stream
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kGatherLps = 16;  // lanes per stream (k_gather_scan's block of 16 streams is built on it)
// one stream's bytes from its slot to its place in the payload, by the kGatherLps lanes li = 0 .. kGatherLps - 1
__device__ __forceinline__ void gather_stream(const uint8_t *slots, uint64_t slot_stride, uint64_t s, uint32_t li, uint32_t len,
                                              uint64_t off, uint8_t *payload, uint64_t cap) {
  if (off + len > cap) return;
  const uint8_t *src = slots + s * slot_stride;  // 16-byte aligned
  uint8_t *dst = payload + off;
  // head bytes up to the first 4-byte aligned destination address, then 16 bytes per lane and step (destination dword
  // aligned, source shifted by `head` bytes: five source dwords funnel into four), then single dwords, then the tail
  const uint32_t head = min(len, (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
  if (li < head) dst[li] = src[li];
  const uint32_t nd = (len - head) >> 2;  // whole destination dwords
  const uint32_t nq = nd >> 2;            // ... in groups of four
  const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
  uint32_t *dst32 = reinterpret_cast<uint32_t *>(dst + head);
  const uint32_t sh = 8 * head;
  struct alignas(4) Q4 { uint32_t w[4]; };
  for (uint32_t q = li; q < nq; q += kGatherLps) {
    const uint4 v = *reinterpret_cast<const uint4 *>(src32 + 4 * q);  // the slot is 16-byte aligned and padded to 16
    const uint32_t x = src32[4 * q + 4];
    Q4 o;
    if (head == 0) {
      o.w[0] = v.x; o.w[1] = v.y; o.w[2] = v.z; o.w[3] = v.w;
    } else {
      o.w[0] = (v.x >> sh) | (v.y << (32 - sh));
      o.w[1] = (v.y >> sh) | (v.z << (32 - sh));
      o.w[2] = (v.z >> sh) | (v.w << (32 - sh));
      o.w[3] = (v.w >> sh) | (x << (32 - sh));
    }
    *reinterpret_cast<Q4 *>(dst32 + 4 * q) = o;
  }
  for (uint32_t j = 4 * nq + li; j < nd; j += kGatherLps) {
    const uint32_t lo = src32[j], hi = src32[j + 1];
    dst32[j] = head == 0 ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
  }
  const uint32_t done = head + nd * 4;
  if (done + li < len) dst[done + li] = src[done + li];
}
__global__ void __launch_bounds__(256) k_gather(const uint8_t *slots, uint64_t slot_stride, const uint32_t *lens,
                                                const uint64_t *offsets, uint64_t S, uint8_t *payload, uint64_t cap) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t s = gid / kGatherLps;
  if (s >= S) return;
  gather_stream(slots, slot_stride, s, (uint32_t)gid & (kGatherLps - 1u), lens[s], offsets[s], payload, cap);
}

// Medium batches (kScanBlock < S <= kScanFusedMax): the offsets scan and the gather in TWO launches instead of four.
// k_scan_partial_top: every block leaves its sum, the block that arrives last (a ticket) scans the <= 256 sums in place and
// writes the total; k_gather_scan: a gather block works out the offsets of its 16 streams itself -- the scanned sum of its
// 1024-stream block + the lengths in front of it, at most 4 KiB read from L2 -- and stores them while it moves the bytes.
constexpr uint64_t kScanFusedMax = 256ull * 1024;
__global__ void __launch_bounds__(kScanThreads) k_scan_partial_top(const uint32_t *lens, uint64_t S, uint64_t *bsum, uint32_t nblk,
                                                                   uint32_t *ticket, uint64_t *offsets_S, uint64_t *host_total) {
  __shared__ bool s_last;
  const uint64_t i0 = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * kScanItems;
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
    if (i0 + k < S) v += lens[i0 + k];
  uint64_t tot;
  block_excl_scan_u64(v, &tot);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&bsum[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == nblk - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const uint64_t mine = threadIdx.x < nblk ? __hip_atomic_load(&bsum[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;  // nblk <= 256
  const uint64_t ex = block_excl_scan_u64(mine, &tot);
  if (threadIdx.x < nblk) bsum[threadIdx.x] = ex;
  if (threadIdx.x == 0) {
    *offsets_S = tot;
    if (host_total) *host_total = tot;
    *ticket = 0u;  // for the next call
  }
}
__global__ void __launch_bounds__(256) k_gather_scan(const uint8_t *slots, uint64_t slot_stride, const uint32_t *lens, const uint64_t *bsum,
                                                     uint64_t S, uint64_t *offsets, uint8_t *payload, uint64_t cap) {
  static_assert(kGatherLps == 16 && kScanBlock % 16 == 0, "a gather block's 16 streams lie inside one scan block");
  __shared__ uint64_t s_part[4];
  __shared__ uint32_t s_len[16];
  const uint64_t first = (uint64_t)blockIdx.x * 16, sblk = first / kScanBlock, lo = sblk * kScanBlock;
  const uint32_t cnt = (uint32_t)(first - lo);  // lengths in front of this block's streams inside their scan block: a multiple of 16
  uint64_t v = 0;
  if (4u * threadIdx.x < cnt) {
    const uint4 q = *reinterpret_cast<const uint4 *>(lens + lo + 4u * threadIdx.x);
    v = (uint64_t)q.x + q.y + q.z + q.w;
  }
  v = wave_incl_scan_u64(v, threadIdx.x & 63);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = v;
  const uint64_t s = first + threadIdx.x / 16;
  const uint32_t li = threadIdx.x & 15u, len = s < S ? lens[s] : 0u;
  if (li == 0) s_len[threadIdx.x / 16] = len;
  __syncthreads();
  uint64_t off = bsum[sblk] + s_part[0] + s_part[1] + s_part[2] + s_part[3];
  for (uint32_t j = 0; j < threadIdx.x / 16; j++) off += s_len[j];
  if (s >= S) return;
  if (li == 0) offsets[s] = off;
  gather_stream(slots, slot_stride, s, li, len, off, payload, cap);
}
// Small batches (S <= kScanBlock: the reference's own use -- a handful of streams per call): offsets scan AND gather in one
// single-block launch instead of four, the offsets handed from the scan to the gather through LDS
__global__ void __launch_bounds__(kScanThreads) k_scan_gather_small(const uint8_t *slots, uint64_t slot_stride, const uint32_t *lens,
                                                                    uint64_t S, uint64_t *offsets, uint8_t *payload, uint64_t cap,
                                                                    uint64_t *host_total) {
  __shared__ uint64_t soff[kScanBlock + 1];
  const uint32_t i0 = threadIdx.x * kScanItems;
  uint32_t l[kScanItems];
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    l[k] = i0 + k < S ? lens[i0 + k] : 0;
    v += l[k];
  }
  uint64_t tot;
  uint64_t run = block_excl_scan_u64(v, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (i0 + k < S) {
      offsets[i0 + k] = run;
      soff[i0 + k] = run;
    }
    run += l[k];
  }
  if (threadIdx.x == 0) {
    offsets[S] = tot;
    if (host_total) *host_total = tot;
  }
  __syncthreads();
  const uint32_t li = threadIdx.x & (kGatherLps - 1u);
  for (uint64_t s = threadIdx.x / kGatherLps; s < S; s += kScanThreads / kGatherLps) gather_stream(slots, slot_stride, s, li, lens[s], soff[s], payload, cap);
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
  return (bound + 16 + 15) & ~15ull;  // the sink writes whole 16-byte pieces
}

// Waves per workgroup of the instances with more than 64 contexts: a workgroup's waves share one copy of the tables, and a CU's
// 160 KiB of LDS hold floor(160 KiB / (tables + w x wave)) workgroups of w waves -- the w (1..4) that leaves the most waves on a CU,
// the smallest such w.  Decoder: 127 contexts 11 -> 12 waves (w = 2), 190 contexts 8 -> 9 (w = 3).
// The CU hands its 160 KiB out in 128 granules of 1280 bytes (measured: tools/ubench/lds_occ.hip, profiles/r06/j_lds_occ.txt -- single-
// wave workgroups of 10 240 bytes: 16 on a CU, of 10 496: 14, of 13 312: 11 where 160 KiB / 13 312 = 12.3; the runtime's occupancy
// query does not know and says 12).  The workgroup size (1 .. 4 waves sharing one copy of the tables) that leaves the most waves on a
// CU; workgroups of fewer than four waves start four and keep theirs on the SIMDs a per-CU ticket names (cabac_wave.h:
// pinned_wave_index -- left to the hardware the waves of small workgroups land unevenly, and four-wave workgroups were the only even
// ones: profiles/r06/r_wpb_sweep.txt).  Among sizes that leave the same number of waves, so placed: 2, then 4, then 1, then 3
// (127 contexts, 12 waves each way: 3.29 / 3.14 ms in twos, 3.32 / 3.19 in fours, 3.39 / 3.44 single, 3.45 / 3.24 in threes; 190
// contexts' encoder, 9 waves: 3.6 ms single, 3.85 in threes; 253 contexts' encoder, 8 waves: 3.72 in twos, 3.81 in fours --
// profiles/r06/ae_wpb_pinned.txt).
static int waves_per_block_for(size_t tab_bytes, size_t wave_bytes) {
  constexpr size_t kGranule = 1280, kGranules = 128;
  int best = 2;
  size_t best_waves = 0;
  for (int w : {2, 4, 1, 3}) {
    const size_t waves = kGranules / ((tab_bytes + w * wave_bytes + kGranule - 1) / kGranule) * w;
    if (waves > best_waves) {
      best_waves = waves;
      best = w;
    }
  }
  return best;
}
template <int CM, bool PERM>
static size_t enc_wave_lds_bytes(uint32_t img_bytes) {
  return WaveLds<CM, PERM>::BYTES - WaveLds<CM, PERM>::SB + (CM == 2 ? img_bytes : WaveLds<CM, PERM>::SB);
}
// CM 2: a wave's state image covers the batch's contexts in whole groups of four (a.nctx > 128 there)
static uint32_t image_bytes_for(uint32_t nctx) { return 256u * ((nctx + 3u) / 4u); }

template <int CM, bool LAT, bool PERM>
static hipError_t run_encode(const EncArgs &a, hipStream_t stream) {
  // (more than 64 contexts: waves_per_block_for -- 127 contexts: two waves share the tables, 12 waves per CU, three on every SIMD)
  const int wpb = CM == 1 ? enc_waves_per_block(1) : waves_per_block_for(kTabBytes, enc_wave_lds_bytes<CM, PERM>(a.img_bytes));
  const size_t lds = kTabBytes + wpb * enc_wave_lds_bytes<CM, PERM>(a.img_bytes);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode<CM, LAT, PERM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTabBytes + 4 * enc_wave_lds_bytes<CM, PERM>(16384)));
  if (err != hipSuccess) return err;
  // LdsStates<.., NEXT16> addresses its table by absolute LDS address: the dynamic LDS must start at 0
  static int static_lds = -1;
  if (static_lds < 0) {
    hipFuncAttributes fa;
    err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_encode<CM, LAT, PERM>));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  const uint64_t nblk = (a.S + 64 * wpb - 1) / (64 * wpb);
  EncArgs aw = a;
  aw.wpb = (uint32_t)wpb;
  // (workgroups of fewer than four waves start four: pinned_wave_index)
  hipLaunchKernelGGL((k_encode<CM, LAT, PERM>), dim3((unsigned)nblk), dim3(CM != 1 ? 256 : 64 * wpb), lds, stream, aw);
  return hipGetLastError();
}

template <int CM, bool LAT, bool PERM>
static hipError_t run_decode(const DecArgs &a, hipStream_t stream) {
  using DL = DecLds<CM, PERM>;
  const size_t wave_bytes = CM == 2 ? DL::BYTES - DL::SB + a.img_bytes : DL::BYTES;
  const int wpb = CM == 1 ? dec_waves_per_block(1) : waves_per_block_for(kDecTabBytes, wave_bytes);
  const size_t lds_max = kDecTabBytes + 4 * DL::BYTES, lds = kDecTabBytes + wpb * wave_bytes;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode<CM, LAT, PERM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
  if (err != hipSuccess) return err;
  static int static_lds = -1;  // absolute LDS addresses, as in run_encode
  if (static_lds < 0) {
    hipFuncAttributes fa;
    err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_decode<CM, LAT, PERM>));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  const uint64_t nblk = (a.S + 64 * wpb - 1) / (64 * wpb);
  DecArgs aw = a;
  aw.wpb = (uint32_t)wpb;
  hipLaunchKernelGGL((k_decode<CM, LAT, PERM>), dim3((unsigned)nblk), dim3(CM != 1 ? 256 : 64 * wpb), lds, stream, aw);
  return hipGetLastError();
}

// sh.latency selects the latency-oriented instance (cabac_capi.cpp: small launches, and every launch of more than 128
// contexts, whose occupancy the state image keeps at or below 2.5 waves per SIMD)
// (sh.instance: the name of the instance that ran, for isscabac_last_instance -- tests and the bench line say which kernel they measured)
#define ISSCABAC_RUN(FN, KN, CM, LAT, PERM, ARGS) \
  do {                                           \
    sh.instance = KN "<" #CM "," #LAT "," #PERM ">"; \
    return FN<CM, LAT, PERM>(ARGS, stream);      \
  } while (0)
#define ISSCABAC_DISPATCH(FN, KN, ARGS, LATF)                                                                \
  do {                                                                                                  \
    if (sh.d_order) { /* length-bucketed launch */                                                      \
      if (sh.nctx <= 64) ISSCABAC_RUN(FN, KN, 1, false, true, ARGS);                                    \
      if (sh.nctx <= 128) ISSCABAC_RUN(FN, KN, 3, false, true, ARGS);                                   \
      ISSCABAC_RUN(FN, KN, 2, false, true, ARGS);                                                       \
    }                                                                                                   \
    if (sh.nctx <= 64) {                                                                                \
      if (LATF) ISSCABAC_RUN(FN, KN, 1, true, false, ARGS);                                       \
      ISSCABAC_RUN(FN, KN, 1, false, false, ARGS);                                                      \
    }                                                                                                   \
    if (sh.nctx <= 128) {                                                                               \
      if (LATF) ISSCABAC_RUN(FN, KN, 3, true, false, ARGS);                                       \
      ISSCABAC_RUN(FN, KN, 3, false, false, ARGS);                                                      \
    }                                                                                                   \
    if (LATF) ISSCABAC_RUN(FN, KN, 2, true, false, ARGS);                                         \
    ISSCABAC_RUN(FN, KN, 2, false, false, ARGS);                                                        \
  } while (0)

hipError_t launch_encode(const BatchShape &sh, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                         uint8_t *d_slots, uint64_t slot_stride, uint32_t *d_lens, hipStream_t stream) {
  EncArgs a;
  a.bins = d_bins;
  a.ctx = d_ctx;
  a.nbins_per_stream = d_nbins;
  a.order = sh.d_order;
  a.slots = d_slots;
  a.lens = d_lens;
  a.S = sh.S;
  a.slot_stride = slot_stride;
  a.nbins = sh.nbins;
  a.pitch = sh.pitch;
  a.nctx = sh.nctx;
  a.img_bytes = image_bytes_for(sh.nctx);
  std::memcpy(a.ctx_init, sh.ctx_init, 256);
  // small launches of <= 64 contexts: the producer / consumer wave pair (cabac_pair.hip)
  if (sh.latency && sh.pair && !sh.d_order && sh.nctx <= 64 && encode_pair_enabled()) {
    sh.instance = "k_encode_pair";
    return run_encode_pair(a, stream);
  }
  ISSCABAC_DISPATCH(run_encode, "k_encode", a, sh.latency_enc);
}

static void fill_dec_args(DecArgs &a, const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes,
                          const uint64_t *d_offsets, const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out,
                          uint32_t *d_status, uint32_t *d_flag) {
  a.payload = d_payload;
  a.ctx = d_ctx;
  a.offsets = d_offsets;
  a.nbins_per_stream = d_nbins;
  a.order = sh.d_order;
  a.bins_out = d_bins_out;
  a.status = d_status;
  a.exact_flag = d_flag;
  a.S = sh.S;
  a.payload_bytes = payload_bytes;
  a.nbins = sh.nbins;
  a.pitch = sh.pitch;
  a.nctx = sh.nctx;
  a.img_bytes = image_bytes_for(sh.nctx);
  std::memcpy(a.ctx_init, sh.ctx_init, 256);
}

hipError_t launch_decode_exact(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes,
                               const uint64_t *d_offsets, const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out,
                               uint32_t *d_status, const uint32_t *d_flag, hipStream_t stream, uint32_t *d_next_flag) {
  DecArgs a;
  fill_dec_args(a, sh, d_payload, payload_bytes, d_offsets, d_ctx, d_nbins, d_bins_out, d_status,
                const_cast<uint32_t *>(d_flag));
  hipLaunchKernelGGL(k_decode_exact, dim3((unsigned)((sh.S + 63) / 64)), dim3(64), 0, stream, a, d_next_flag);
  return hipGetLastError();
}

hipError_t launch_decode(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                         const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out, uint32_t *d_status,
                         uint32_t *d_flag, hipStream_t stream) {
  DecArgs a;
  fill_dec_args(a, sh, d_payload, payload_bytes, d_offsets, d_ctx, d_nbins, d_bins_out, d_status, d_flag);
  // small launches of <= 64 contexts: the decoder + helper wave pair (cabac_pair.hip)
  if (sh.latency && sh.tiny && sh.pair && !sh.d_order && sh.nctx <= 64 && decode_pair_enabled()) {
    sh.instance = "k_decode_pair";
    return run_decode_pair(a, stream);
  }
  ISSCABAC_DISPATCH(run_decode, "k_decode", a, sh.latency);
}

// ------------------------------------------------------------------------------------------------
// bins as bits on the PCIe link (isscabac_encode_batch_packed / isscabac_decode_batch_packed): row s of the packed array
// holds bin i of stream s at bit i % 8 of byte i / 8 (pitch / 8 bytes per row); one thread moves 16 bins
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack_bits(const uint8_t *packed, uint8_t *bins, uint64_t S, uint32_t pitch) {
  const uint32_t cpr = pitch / 16;  // 16-bin pieces per row
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t r = gid / cpr;
  const uint32_t c = (uint32_t)(gid % cpr);
  if (r >= S) return;
  const uint32_t w = *reinterpret_cast<const uint16_t *>(packed + r * (pitch / 8) + 2 * c);
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t nib = (w >> (4 * k)) & 0xFu;  // bins 4k .. 4k+3 at bits 0..3 -> one byte each
    o[k] = (nib | nib << 7 | nib << 14 | nib << 21) & 0x01010101u;
  }
  *reinterpret_cast<uint4 *>(bins + r * pitch + 16 * c) = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ void __launch_bounds__(256) k_pack_bits(const uint8_t *bins, uint8_t *packed, uint64_t S, uint32_t pitch) {
  const uint32_t cpr = pitch / 16;
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t r = gid / cpr;
  const uint32_t c = (uint32_t)(gid % cpr);
  if (r >= S) return;
  const uint4 v = *reinterpret_cast<const uint4 *>(bins + r * pitch + 16 * c);
  const uint32_t in[4] = {v.x, v.y, v.z, v.w};
  uint32_t w = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = in[k] & 0x01010101u;  // bytes of 0 / 1 -> a nibble
    w |= ((x | x >> 7 | x >> 14 | x >> 21) & 0xFu) << (4 * k);
  }
  *reinterpret_cast<uint16_t *>(packed + r * (pitch / 8) + 2 * c) = (uint16_t)w;
}
hipError_t launch_unpack_bits(const uint8_t *d_packed, uint8_t *d_bins, uint64_t S, uint32_t pitch, hipStream_t stream) {
  const uint64_t total = S * (pitch / 16);
  hipLaunchKernelGGL(k_unpack_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_packed, d_bins, S, pitch);
  return hipGetLastError();
}
hipError_t launch_pack_bits(const uint8_t *d_bins, uint8_t *d_packed, uint64_t S, uint32_t pitch, hipStream_t stream) {
  const uint64_t total = S * (pitch / 16);
  hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_bins, d_packed, S, pitch);
  return hipGetLastError();
}

uint64_t scan_tmp_bytes(uint64_t S) { return ((S + kScanBlock - 1) / kScanBlock + 1) * sizeof(uint64_t); }

hipError_t launch_scan(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, hipStream_t stream, uint64_t *d_host_total) {
  const uint64_t nblk = (S + kScanBlock - 1) / kScanBlock;
  uint64_t *bsum = static_cast<uint64_t *>(d_tmp);
  hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nblk), dim3(kScanThreads), 0, stream, d_lens, S, bsum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanThreads), 0, stream, bsum, nblk);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblk), dim3(kScanThreads), 0, stream, d_lens, S, bsum, d_offsets, d_host_total);
  return hipGetLastError();
}

bool scan_gather_is_fused(uint64_t S) { return S <= (uint64_t)kScanBlock; }
bool scan_gather_is_paired(uint64_t S, const uint32_t *d_ticket) { return d_ticket && S > (uint64_t)kScanBlock && S <= kScanFusedMax; }
// the two halves of the medium path, for callers that time them apart (scan_gather_is_paired() holds)
hipError_t launch_scan_partial_top(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, uint32_t *d_ticket,
                                   uint64_t *d_host_total, hipStream_t stream) {
  const uint64_t nblk = (S + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(k_scan_partial_top, dim3((unsigned)nblk), dim3(kScanThreads), 0, stream, d_lens, S, static_cast<uint64_t *>(d_tmp),
                     (uint32_t)nblk, d_ticket, d_offsets + S, d_host_total);
  return hipGetLastError();
}
hipError_t launch_gather_scan(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets,
                              void *d_tmp, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream) {
  hipLaunchKernelGGL(k_gather_scan, dim3((unsigned)((S + 15) / 16)), dim3(256), 0, stream, d_slots, slot_stride, d_lens,
                     static_cast<const uint64_t *>(d_tmp), S, d_offsets, d_payload, payload_cap);
  return hipGetLastError();
}
// lens -> offsets and slots -> payload: one launch for S <= kScanBlock streams, scan (3) + gather (1) otherwise
hipError_t launch_scan_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets,
                              void *d_tmp, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream, uint32_t *d_ticket,
                              uint64_t *d_host_total) {
  if (scan_gather_is_fused(S)) {
    hipLaunchKernelGGL(k_scan_gather_small, dim3(1), dim3(kScanThreads), 0, stream, d_slots, slot_stride, d_lens, S, d_offsets, d_payload,
                       payload_cap, d_host_total);
    return hipGetLastError();
  }
  if (scan_gather_is_paired(S, d_ticket)) {
    const hipError_t e1 = launch_scan_partial_top(d_lens, S, d_offsets, d_tmp, d_ticket, d_host_total, stream);
    return e1 != hipSuccess ? e1 : launch_gather_scan(d_slots, slot_stride, d_lens, S, d_offsets, d_tmp, d_payload, payload_cap, stream);
  }
  const hipError_t err = launch_scan(d_lens, S, d_offsets, d_tmp, stream, d_host_total);
  return err != hipSuccess ? err : launch_gather(d_slots, slot_stride, d_lens, d_offsets, S, d_payload, payload_cap, stream);
}

hipError_t launch_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, const uint64_t *d_offsets,
                         uint64_t S, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream) {
  const uint64_t nblk = (S * kGatherLps + 255) / 256;
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
