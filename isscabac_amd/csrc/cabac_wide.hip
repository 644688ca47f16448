// cabac_wide.hip -- batch coding with MORE than 255 contexts: context indices are 16-bit (0xFFFF = bypass bin), up to the
// 999 context models the reference holds per coder (ref: CABAC/CommonDef.h:56, CABAC_ContextModelsInit.cpp:53,84; the
// per-bin commands index them with a double, SimpleCABACMex.cpp:223-224).  gfx950, wave64.
//
// Same mapping as the other batch kernels -- one lane = one bitstream, the per-bin arithmetic of cabac_lane.h -- with the
// state image of the batch's own size in the LDS: 256 bytes per four contexts and wave, at (ctx / 4) * 256 + lane * 4 +
// ctx % 4 like the narrow kernels' (lane l only touches bank l % 32), so 999 contexts are 64 000 bytes per wave.  A
// workgroup is ONE wave -- with the tables (and the encoder's output staging) 64.5 KiB at 999 contexts, two workgroups per
// CU; 500 contexts leave room for four.  The state of the next bin is requested a bin ahead and forwarded in a register
// when it names the same context, so a bin's chain holds one LDS round trip.  A piece of 16 bins in which every lane
// codes 16 context bins (the ordinary case) runs straight-line code: the encoder's single-lookup step accumulated per
// four bins with the wave-uniform digit flush, the decoder with one unchecked top-up per four bins; bypass bins, ragged
// ends and invalid indices take per-lane branches.  Input rows are read per lane, 16 bins a piece, one piece ahead.  (The
// first version kept the image in a device scratch: 31-39 Gbin/s, profiles/r05/wide_rate.txt.)  Batches of up to 255
// contexts never come here.
#include "cabac_wave.h"

namespace isscabac {

// (the initialisation array on the device holds 1 024 packed states; indices are checked against nctx)
constexpr uint32_t kWideBypass = 0xFFFFu;
// LDS of a workgroup (dynamic, no static __shared__): [0, 256) rangeTabLps rows by state, [256, 512) next-state entries,
// [512, 512 + image) the wave's state image, then (encoder) 16 bytes of output staging per lane
constexpr uint32_t kWideTab = 512, kWideStage = 1024;
// the encoder: [0, 1024) its 8-byte entries by packed state (row | enc_entry_hi, cabac_lane.h: enc_bin_row_acc), then the image
constexpr uint32_t kWideEncTab = 1024;
__host__ __device__ constexpr uint32_t wide_image_bytes(uint32_t nctx) { return 256u * ((nctx + 3u) / 4u); }

// States policy over the LDS image: load_at / store_at by context index
struct WideStates {
  uint8_t *img;          // the image + lane * 4
  const uint32_t *tab;   // the two tables
  __device__ __forceinline__ uint8_t *at(uint32_t c) const { return img + (((c & ~3u) << 6) | (c & 3u)); }
  __device__ __forceinline__ uint32_t load_at(uint32_t c) const { return *at(c); }
  __device__ __forceinline__ void store_at(uint32_t c, uint32_t q) const { *at(c) = (uint8_t)q; }
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return load_at(c); }
  __device__ __forceinline__ void store(uint32_t c, uint32_t q) const { store_at(c, q); }
  // (by absolute LDS address -- the tables start at LDS 0, no static __shared__ in these kernels: checked at the launch -- so
  // that the address is ONE instruction on the bin-to-bin chain and the second table's offset folds into the read)
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  __device__ __forceinline__ uint32_t lps_row(uint32_t q) const { return *reinterpret_cast<const lds_u32 *>((uintptr_t)(q & 0xFCu)); }
  __device__ __forceinline__ uint32_t next_entry(uint32_t q) const { return *reinterpret_cast<const lds_u32 *>((uintptr_t)(256u + (q & 0xFCu))); }
};

// The same image with the state of the NEXT bin requested one bin ahead: the read is issued before the
// current bin's store -- LDS operations complete in order, so it sees every store but that one -- and the current bin's
// successor is forwarded in a register when the next bin names the same context.  A bin's chain is then one LDS round trip
// (row + successors of its state) instead of two.
struct WideFwdStates : WideStates {
  uint32_t last_c, last_q, pre_q, cur_q;
  __device__ __forceinline__ WideFwdStates(const WideStates &b) : WideStates(b), last_c(~0u), last_q(0), pre_q(0), cur_q(0) {}
  __device__ __forceinline__ void request(uint32_t c) { pre_q = *at(c); }  // c must be a valid index
  __device__ __forceinline__ void begin(uint32_t c) { cur_q = c == last_c ? last_q : pre_q; }  // before the next request()
  __device__ __forceinline__ uint32_t load_at(uint32_t) const { return cur_q; }
  __device__ __forceinline__ void store_at(uint32_t c, uint32_t q) {
    *at(c) = (uint8_t)q;
    last_c = c;
    last_q = q & 0xFFu;
  }
};
// context of bin k of a 16-bin piece held as two uint4 of 16-bit indices
__device__ __forceinline__ uint32_t wide_ctx_of(const uint4 &c0, const uint4 &c1, int k) {
  const uint32_t cw = word_of(k < 8 ? c0 : c1, (k >> 1) & 3);
  return (k & 1) ? cw >> 16 : cw & 0xFFFFu;
}

// The encoder's policy: the single-lookup step (one 8-byte entry per bin: row + successors by bin + LPS flags) on the same image
struct WideEncStates {
  uint8_t *img;   // the image + lane * 4
  uint32_t two;   // the constant 2 in a register (the SDWA shift's first operand)
  uint32_t last_c, last_q, pre_q, cur_q;
  __device__ __forceinline__ uint8_t *at(uint32_t c) const { return img + (((c & ~3u) << 6) | (c & 3u)); }
  __device__ __forceinline__ void request(uint32_t c) { pre_q = *at(c); }
  __device__ __forceinline__ void begin(uint32_t c) { cur_q = c == last_c ? last_q : pre_q; }
  __device__ __forceinline__ uint32_t load_at(uint32_t) const { return cur_q; }
  __device__ __forceinline__ void store_at(uint32_t c, uint32_t t) {
    *at(c) = (uint8_t)t;
    last_c = c;
    last_q = t;  // (only its low byte is the state: entry_mem reads byte 0)
  }
  // the entry of the doubled state in the LOW BYTE of q, at LDS address (q >> 1) * 8 = byte 0 of q << 2 -- one SDWA shift on the
  // bin-to-bin chain; the table starts at LDS address 0 (no static __shared__ in this kernel: checked at the launch)
  __device__ __forceinline__ uint64_t entry_mem(uint32_t q) const {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(two), "v"(q));
    return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t *>((uintptr_t)a);
  }
};
// Bin policy of enc_bin_row_acc for one bin in a register
struct WideBin {
  uint32_t bin;
  __device__ __forceinline__ uint32_t next_of(uint32_t hi) const { return hi >> (8u * bin); }
  __device__ __forceinline__ uint32_t lps_mask(uint32_t, uint32_t t) const { return (uint32_t)__builtin_amdgcn_sbfe(t, 16, 1); }
};
typedef unsigned short wide_us2 __attribute__((ext_vector_type(2)));
// the largest of the 16 indices of a piece
__device__ __forceinline__ uint32_t wide_ctx_max(const uint4 &c0, const uint4 &c1) {
  auto mx = [](uint32_t a, uint32_t b) {
    const wide_us2 r = __builtin_elementwise_max(__builtin_bit_cast(wide_us2, a), __builtin_bit_cast(wide_us2, b));
    return __builtin_bit_cast(uint32_t, r);
  };
  const uint32_t m = mx(mx(mx(c0.x, c0.y), mx(c0.z, c0.w)), mx(mx(c1.x, c1.y), mx(c1.z, c1.w)));
  return max(m & 0xFFFFu, m >> 16);
}

struct WideEncArgs {
  const uint8_t *bins;
  const uint16_t *ctx;
  const uint32_t *nbins_per_stream;
  uint8_t *slots;
  uint32_t *lens;
  const uint8_t *ctx_init;  // device, 1 024 packed states (mps | state << 1)
  uint64_t S, slot_stride;
  uint32_t nbins, pitch, nctx;
};

// tables and this wave's image; returns the policy.  ctx_init holds 1 024 bytes, so the rounded-up tail is in bounds
__device__ __forceinline__ WideStates wide_setup(uint8_t *smem, const uint8_t *ctx_init, uint32_t nctx, int lane) {
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
  tab[lane] = c_lps_rows[lane];
  tab[64 + lane] = c_next.e[lane];
  uint32_t *img32 = reinterpret_cast<uint32_t *>(smem + kWideTab) + lane;
  const uint32_t *init32 = reinterpret_cast<const uint32_t *>(ctx_init);
  for (uint32_t g = 0; g < (nctx + 3u) / 4u; g++) img32[g * 64] = (init32[g] & 0x7F7F7F7Fu) << 1;  // four contexts a word
  __syncthreads();  // (one wave: orders the table writes before the other lanes' reads)
  return WideStates{smem + kWideTab + lane * 4, tab};
}

__global__ void __launch_bounds__(64) k_encode_wide(WideEncArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  const uint64_t s0 = (uint64_t)blockIdx.x * 64;
  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = valid ? min(n, a.nbins) : 0u;
  {
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int ps = lane + 64 * h;  // packed state
      const uint32_t sb = c_next.eb[ps >> 1] >> (16 * (ps & 1));
      tab[2 * ps] = ps >= 126 ? 0u : c_lps_rows[ps >> 1];
      tab[2 * ps + 1] = ps >= 126 ? 0u : enc_entry_hi(ps, sb & 0xFFu, (sb >> 8) & 0xFFu);
    }
    uint32_t *img32 = reinterpret_cast<uint32_t *>(smem + kWideEncTab) + lane;
    const uint32_t *init32 = reinterpret_cast<const uint32_t *>(a.ctx_init);
    for (uint32_t g = 0; g < (a.nctx + 3u) / 4u; g++) img32[g * 64] = (init32[g] & 0x7F7F7F7Fu) << 1;  // four contexts a word
  }
  __syncthreads();
  WideEncStates fs{smem + kWideEncTab + lane * 4, opaque(2u), ~0u, 0u, 0u, 0u};
  EncCore e;
  enc_start(e);
  SlotSink sink;
  sink.stage = smem + kWideEncTab + wide_image_bytes(a.nctx) + lane * 16;
  sink.slots = a.slots + s0 * a.slot_stride;
  sink.mine = (uint32_t)(s - s0) * (uint32_t)a.slot_stride;
  sink.pos = 0;
  sink.on = valid;
  const uint8_t *brow = a.bins + s * a.pitch;
  const uint16_t *crow = a.ctx + s * a.pitch;
  const uint32_t nmax = __builtin_amdgcn_readfirstlane(wave_max_u32(n)), nmin = __builtin_amdgcn_readfirstlane(wave_min_u32(n));
  // pitch is a multiple of 16: a piece lies inside the row; the next one is requested before this one is coded
  uint4 wb = make_uint4(0, 0, 0, 0), c0 = wb, c1 = wb;
  if (n) {
    wb = *reinterpret_cast<const uint4 *>(brow);
    c0 = *reinterpret_cast<const uint4 *>(crow);
    c1 = *reinterpret_cast<const uint4 *>(crow + 8);
  }
  const uint32_t cmax = a.nctx - 1;
  uint32_t cn = wide_ctx_of(c0, c1, 0);
  fs.request(min(cn, cmax));  // (a bypass marker or an invalid index reads some valid row; the value is not used)
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < nmax; i0 += 16) {
    uint4 nwb = wb, nc0 = c0, nc1 = c1;
    if (i0 + 16 < n) {
      nwb = *reinterpret_cast<const uint4 *>(brow + i0 + 16);
      nc0 = *reinterpret_cast<const uint4 *>(crow + i0 + 16);
      nc1 = *reinterpret_cast<const uint4 *>(crow + i0 + 24);
    }
    // PLAIN piece: every lane has 16 bins and all of them name a context -- straight-line code, the single-lookup step
    // accumulated per four bins, output moved when some lane could overflow `low` within the next four
    if (i0 + 16 <= nmin && !wave_any(wide_ctx_max(c0, c1) > cmax)) {
      EncWordAcc acc;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const uint32_t c = cn;
        cn = k < 15 ? wide_ctx_of(c0, c1, k + 1) : wide_ctx_of(nc0, nc1, 0);  // (the next piece's first bin may be anything)
        fs.begin(c);
        fs.request(k < 15 ? cn : min(cn, cmax));
        enc_bin_row_acc(e, acc, (k & 3) == 0, WideBin{byte_of(wb, k) & 1u}, c, fs);
        if ((k & 3) == 3) {
          enc_word_close(e, acc);
          if (wave_any(e.W >= ISSCABAC_FLUSH_AT)) enc_flush16_wave(e, sink);
        }
      }
    } else {
      // bypass bins, ragged ends, invalid indices: every lane takes its own branch
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const uint32_t c = cn;
        cn = k < 15 ? wide_ctx_of(c0, c1, k + 1) : wide_ctx_of(nc0, nc1, 0);
        fs.begin(c);
        fs.request(min(cn, cmax));
        if (i0 + k < n) {
          const uint32_t bin = byte_of(wb, k) & 1u;
          if (c == kWideBypass) {
            enc_bin_ep(e, bin);
          } else if (c < a.nctx) {
            EncWordAcc acc;
            enc_bin_row_acc(e, acc, true, WideBin{bin}, c, fs);
            enc_word_close(e, acc);
          }
          // (any other value is invalid: skipped, that stream's result is unspecified, include/isscabac.h)
        }
        if ((k & 3) == 3) enc_flush16(e, sink);  // (keeps W below 39 + 4 x 6)
      }
    }
    wb = nwb;
    c0 = nc0;
    c1 = nc1;
  }
  enc_flush16(e, sink);
  enc_finish16(e, sink);
  sink.close();
  if (valid) a.lens[s] = sink.bytes();
}

struct WideDecArgs {
  const uint8_t *payload;
  const uint16_t *ctx;
  const uint64_t *offsets;
  const uint32_t *nbins_per_stream;
  uint8_t *bins_out;
  uint32_t *status;
  const uint8_t *ctx_init;
  uint64_t S, payload_bytes;
  uint32_t nbins, pitch, nctx;
};

__global__ void __launch_bounds__(64) k_decode_wide(WideDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  const uint64_t s0 = (uint64_t)blockIdx.x * 64;
  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  WideStates st = wide_setup(smem, a.ctx_init, a.nctx, lane);
  uint64_t start = valid ? a.offsets[s] : 0;
  const uint32_t len = valid ? clip_stream(start, a.offsets[s + 1], a.payload_bytes) : 0u;
  // a first byte of 0xFF (no encoder emits it; an empty stream reads 0xFF at EOF) goes the reference's own way below
  const bool by_ref = valid && (len != 0 ? needs_ref_decoder(a.payload[start]) : true);
  // the bin counts of the lanes that decode in the main loop, while every lane is still here
  const bool main_lane = valid && !by_ref;
  const uint32_t nmax = __builtin_amdgcn_readfirstlane(wave_max_u32(main_lane ? n : 0u));
  const uint32_t nmin = __builtin_amdgcn_readfirstlane(wave_min_u32(main_lane ? n : 0xFFFFFFFFu));
  if (!valid) return;
  const uint16_t *crow = a.ctx + s * a.pitch;
  uint8_t *orow = a.bins_out + s * a.pitch;
  const uint32_t nrow = min((a.nbins + 15u) & ~15u, a.pitch);  // the row is written up to here, zeros behind the stream's end
  if (by_ref) {
    // the reference's own formulation, so that even this garbage and its status word are the reference's (cabac_lane.h:
    // RefDec, k_decode_exact)
    RefReader rd{a.payload + start, len, 0, 0};
    RefDec r;
    ref_dec_start(r, rd);
    for (uint32_t i = 0; i < nrow; i++) {
      uint32_t bin = 0;
      if (i < n) {
        const uint32_t c = crow[i];
        if (c == kWideBypass) bin = ref_dec_bin_ep(r, rd);
        else if (c < a.nctx) bin = ref_dec_bin(r, c, st, rd);
      }
      orow[i] = (uint8_t)bin;
    }
    if (a.status) a.status[s] = ref_dec_finish(r, rd, len);
    return;
  }
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
  uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
  if (n) {
    c0 = *reinterpret_cast<const uint4 *>(crow);
    c1 = *reinterpret_cast<const uint4 *>(crow + 8);
  }
  WideFwdStates fs(st);
  const uint32_t cmax = a.nctx - 1;
  uint32_t cn = wide_ctx_of(c0, c1, 0);
  fs.request(min(cn, cmax));
  // (from here on the loop is uniform over the lanes still present; ballots see only them)
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < nrow; i0 += 16) {
    uint4 ob = make_uint4(0, 0, 0, 0);
    uint4 nc0 = c0, nc1 = c1;
    if (i0 + 16 < n) {  // the next piece of contexts, requested before this one is decoded
      nc0 = *reinterpret_cast<const uint4 *>(crow + i0 + 16);
      nc1 = *reinterpret_cast<const uint4 *>(crow + i0 + 24);
    }
    if (i0 + 16 <= nmin && !wave_any(wide_ctx_max(c0, c1) > cmax)) {
      // PLAIN piece: 16 context bins for every lane -- no lane branches; a top-up per four bins (three bins after one cannot
      // run short, the fourth is guarded wave-uniformly), the window re-based per piece and once in its middle
      if (i0) src.boundary();
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if ((k & 3) == 0) {
          if (k == 8) src.ensure(47);
          dec_topup_nocheck(d, src);
        }
        if ((k & 3) == 3) dec_guard(d, src, 8);
        const uint32_t c = cn;
        cn = k < 15 ? wide_ctx_of(c0, c1, k + 1) : wide_ctx_of(nc0, nc1, 0);  // (the next piece's first bin may be anything)
        fs.begin(c);
        fs.request(k < 15 ? cn : min(cn, cmax));
        const uint32_t bin = (dec_bin_at_b1(d, c, fs) >> 1) & 1u;
        set_word(ob, k >> 2, word_of(ob, k >> 2) | bin << (8 * (k & 3)));
      }
    } else if (i0 < nmax) {
      if (i0 < n) {
        if (i0) src.boundary();  // adopt the piece requested a chunk ago, re-base, request the next
#pragma unroll
        for (int k = 0; k < 16; k++) {
          uint32_t bin = 0;
          const uint32_t c = cn;
          cn = k < 15 ? wide_ctx_of(c0, c1, k + 1) : wide_ctx_of(nc0, nc1, 0);
          fs.begin(c);
          fs.request(min(cn, cmax));
          if (i0 + k < n) {
            if (c == kWideBypass) {
              if (d.sh < 8) dec_topup(d, src);
              bin = dec_bin_ep(d);
            } else if (c < a.nctx) {
              if (d.sh < 13) dec_topup(d, src);  // a context bin needs sh >= 7 and takes at most 6 bits
              bin = (dec_bin_at_b1(d, c, fs) >> 1) & 1u;
            }
          }
          set_word(ob, k >> 2, word_of(ob, k >> 2) | bin << (8 * (k & 3)));
        }
      }
    }
    *reinterpret_cast<uint4 *>(orow + i0) = ob;
    c0 = nc0;
    c1 = nc1;
  }
  if (a.status) {
    const uint32_t sk = (uint32_t)start & 3u;
    const uint8_t *p = a.payload + start;
    a.status[s] = dec_finish(d, src.bpos - 8u * sk, src.len - sk, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

// dynamic LDS of a workgroup
static size_t wide_lds_bytes(uint32_t nctx, bool encoder) {
  return encoder ? kWideEncTab + wide_image_bytes(nctx) + kWideStage : kWideTab + wide_image_bytes(nctx);
}

hipError_t launch_encode_wide(const uint8_t *d_bins, const uint16_t *d_ctx, const uint32_t *d_nbins, uint64_t S, uint32_t nbins,
                              uint32_t pitch, uint32_t nctx, const uint8_t *d_ctx_init, uint8_t *d_slots, uint64_t slot_stride,
                              uint32_t *d_lens, hipStream_t stream) {
  WideEncArgs a{d_bins, d_ctx, d_nbins, d_slots, d_lens, d_ctx_init, S, slot_stride, nbins, pitch, nctx};
  static int static_lds = -1;  // absolute LDS addresses in the kernel: its dynamic LDS must start at 0
  if (static_lds < 0) {
    hipFuncAttributes fa;
    const hipError_t err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_encode_wide));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  static bool granted[64] = {};  // per device: 999 contexts need 66 048 bytes, above the 64 KiB a kernel gets without asking
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !granted[dev]) {
    const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)wide_lds_bytes(ISSCABAC_MAX_CTX16, true));
    if (err != hipSuccess) return err;
    if (dev >= 0 && dev < 64) granted[dev] = true;
  }
  hipLaunchKernelGGL(k_encode_wide, dim3((unsigned)((S + 63) / 64)), dim3(64), wide_lds_bytes(nctx, true), stream, a);
  return hipGetLastError();
}

hipError_t launch_decode_wide(const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets, const uint16_t *d_ctx,
                              const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx,
                              const uint8_t *d_ctx_init, uint8_t *d_bins_out, uint32_t *d_status, hipStream_t stream) {
  WideDecArgs a{d_payload, d_ctx, d_offsets, d_nbins, d_bins_out, d_status, d_ctx_init, S, payload_bytes, nbins, pitch, nctx};
  static int static_lds = -1;  // absolute LDS addresses in the kernel: its dynamic LDS must start at 0
  if (static_lds < 0) {
    hipFuncAttributes fa;
    const hipError_t err = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_decode_wide));
    if (err != hipSuccess) return err;
    static_lds = (int)fa.sharedSizeBytes;
  }
  if (static_lds != 0) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(k_decode_wide, dim3((unsigned)((S + 63) / 64)), dim3(64), wide_lds_bytes(nctx, false), stream, a);
  return hipGetLastError();
}

}  // namespace isscabac
