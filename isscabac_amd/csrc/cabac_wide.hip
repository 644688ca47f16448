// cabac_wide.hip -- batch coding with MORE than 255 contexts: context indices are 16-bit (0xFFFF = bypass bin), up to the
// 999 context models the reference holds per coder (ref: CABAC/CommonDef.h:56, CABAC_ContextModelsInit.cpp:53,84; the
// per-bin commands index them with a double, SimpleCABACMex.cpp:223-224).  gfx950, wave64.
//
// Same mapping as the other batch kernels -- one lane = one bitstream, the per-bin arithmetic of cabac_lane.h -- but the
// state image of 1 024 bytes per stream does not fit the LDS (64 KiB per wave), so it lives in an engine-owned scratch in
// device memory, [wave][context][lane] (a wave's 64 KiB stay L2-resident while it runs), and the tables are read from
// constant memory.  Every bin is a dependent memory round trip: this path exists for completeness of the API against the
// reference, its rate is on record in DESIGN.md; batches of up to 255 contexts never come here.
#include "cabac_wave.h"

namespace isscabac {

constexpr uint32_t kWideCtx = 1024;          // context rows of a wave's image (indices are checked against nctx first)
constexpr uint32_t kWideBypass = 0xFFFFu;

// States policy over the scratch image: load_at / store_at by context index
struct GlobalStates {
  uint8_t *img;  // this wave's image + lane
  __device__ __forceinline__ uint32_t load_at(uint32_t c) const { return img[(size_t)c * 64]; }
  __device__ __forceinline__ void store_at(uint32_t c, uint32_t q) const { img[(size_t)c * 64] = (uint8_t)q; }
  __device__ __forceinline__ uint32_t load(uint32_t c) const { return load_at(c); }
  __device__ __forceinline__ void store(uint32_t c, uint32_t q) const { store_at(c, q); }
  __device__ __forceinline__ uint32_t lps_row(uint32_t q) const { return c_lps_rows[q >> 2]; }
  __device__ __forceinline__ uint32_t next_entry(uint32_t q) const { return c_next.e[q >> 2]; }
};

struct WideEncArgs {
  const uint8_t *bins;
  const uint16_t *ctx;
  const uint32_t *nbins_per_stream;
  uint8_t *slots, *states;
  uint32_t *lens;
  const uint8_t *ctx_init;  // device, kWideCtx packed states (mps | state << 1)
  uint64_t S, slot_stride;
  uint32_t nbins, pitch, nctx;
};

__device__ __forceinline__ void wide_init_states(uint8_t *img, const uint8_t *ctx_init, int lane) {
  // (the whole image: a context the caller did not initialise is the reference's default ContextModel())
  for (uint32_t c = 0; c < kWideCtx; c++) img[(size_t)c * 64] = (uint8_t)((ctx_init[c] & 0x7Fu) << 1);
}

__global__ void __launch_bounds__(256) k_encode_wide(WideEncArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[4 * 64 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + wv, s0 = w * 64;
  if (s0 >= a.S) return;
  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = valid ? min(n, a.nbins) : 0u;
  GlobalStates st{a.states + w * (uint64_t)(kWideCtx * 64) + lane};
  wide_init_states(st.img, a.ctx_init, lane);
  EncCore e;
  enc_start(e);
  SlotSink sink;
  sink.stage = stage + (wv * 64 + lane) * 16;
  sink.slots = a.slots + s0 * a.slot_stride;
  sink.mine = (uint32_t)(s - s0) * (uint32_t)a.slot_stride;
  sink.pos = 0;
  sink.on = valid;
  const uint8_t *brow = a.bins + s * a.pitch;
  const uint16_t *crow = a.ctx + s * a.pitch;
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < n; i0 += 16) {
    const uint4 wb = *reinterpret_cast<const uint4 *>(brow + i0);  // pitch is a multiple of 16: the piece lies inside the row
    const uint4 c0 = *reinterpret_cast<const uint4 *>(crow + i0), c1 = *reinterpret_cast<const uint4 *>(crow + i0 + 8);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (i0 + k < n) {
        const uint32_t bin = byte_of(wb, k) & 1u;
        const uint32_t cw = word_of(k < 8 ? c0 : c1, (k >> 1) & 3), c = (k & 1) ? cw >> 16 : cw & 0xFFFFu;
        if (c == kWideBypass) enc_bin_ep(e, bin);
        else if (c < a.nctx) enc_bin_at(e, BinB1{bin << 1}, c, st);
        // (any other value is invalid: skipped, that stream's result is unspecified, include/isscabac.h)
      }
      if ((k & 3) == 3) enc_flush16(e, sink);  // (keeps W below 29 + 4 x 6)
    }
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
  uint8_t *bins_out, *states;
  uint32_t *status;
  const uint8_t *ctx_init;
  uint64_t S, payload_bytes;
  uint32_t nbins, pitch, nctx;
};

__global__ void __launch_bounds__(256) k_decode_wide(WideDecArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + wv, s0 = w * 64;
  if (s0 >= a.S) return;
  const bool valid = s0 + lane < a.S;
  const uint64_t s = valid ? s0 + lane : a.S - 1;
  uint32_t n = a.nbins_per_stream ? a.nbins_per_stream[s] : a.nbins;
  n = min(n, a.nbins);
  GlobalStates st{a.states + w * (uint64_t)(kWideCtx * 64) + lane};
  wide_init_states(st.img, a.ctx_init, lane);
  if (!valid) return;  // (no cross-lane operation below)
  uint64_t start = a.offsets[s];
  const uint32_t len = clip_stream(start, a.offsets[s + 1], a.payload_bytes);
  const uint16_t *crow = a.ctx + s * a.pitch;
  uint8_t *orow = a.bins_out + s * a.pitch;
  const uint32_t nrow = min((a.nbins + 15u) & ~15u, a.pitch);  // the row is written up to here, zeros behind the stream's end
  if (len != 0 ? needs_ref_decoder(a.payload[start]) : true) {
    // a first byte of 0xFF (no encoder emits it; an empty stream reads 0xFF at EOF): the reference's own formulation, so
    // that even this garbage and its status word are the reference's (cabac_lane.h: RefDec, k_decode_exact)
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
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < nrow; i0 += 16) {
    uint4 ob = make_uint4(0, 0, 0, 0);
    if (i0 < n) {
      if (i0) src.boundary();  // adopt the piece requested a chunk ago, re-base, request the next
      const uint4 c0 = *reinterpret_cast<const uint4 *>(crow + i0), c1 = *reinterpret_cast<const uint4 *>(crow + i0 + 8);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        uint32_t bin = 0;
        if (i0 + k < n) {
          const uint32_t cw = word_of(k < 8 ? c0 : c1, (k >> 1) & 3), c = (k & 1) ? cw >> 16 : cw & 0xFFFFu;
          if (c == kWideBypass) {
            if (d.sh < 8) dec_topup(d, src);
            bin = dec_bin_ep(d);
          } else if (c < a.nctx) {
            if (d.sh < 13) dec_topup(d, src);  // a context bin needs sh >= 7 and takes at most 6 bits
            bin = (dec_bin_at_b1(d, c, st) >> 1) & 1u;
          }
        }
        set_word(ob, k >> 2, word_of(ob, k >> 2) | bin << (8 * (k & 3)));
      }
    }
    *reinterpret_cast<uint4 *>(orow + i0) = ob;
  }
  if (a.status) {
    const uint32_t sk = (uint32_t)start & 3u;
    const uint8_t *p = a.payload + start;
    a.status[s] = dec_finish(d, src.bpos - 8u * sk, src.len - sk, [p](uint32_t i) { return (uint32_t)p[i]; });
  }
}

hipError_t launch_encode_wide(const uint8_t *d_bins, const uint16_t *d_ctx, const uint32_t *d_nbins, uint64_t S, uint32_t nbins,
                              uint32_t pitch, uint32_t nctx, const uint8_t *d_ctx_init, uint8_t *d_states, uint8_t *d_slots,
                              uint64_t slot_stride, uint32_t *d_lens, hipStream_t stream) {
  WideEncArgs a{d_bins, d_ctx, d_nbins, d_slots, d_states, d_lens, d_ctx_init, S, slot_stride, nbins, pitch, nctx};
  hipLaunchKernelGGL(k_encode_wide, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_decode_wide(const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets, const uint16_t *d_ctx,
                              const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx,
                              const uint8_t *d_ctx_init, uint8_t *d_states, uint8_t *d_bins_out, uint32_t *d_status, hipStream_t stream) {
  WideDecArgs a{d_payload, d_ctx, d_offsets, d_nbins, d_bins_out, d_states, d_status, d_ctx_init, S, payload_bytes, nbins, pitch, nctx};
  hipLaunchKernelGGL(k_decode_wide, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

uint64_t wide_state_bytes(uint64_t S) { return ((S + 63) / 64) * (uint64_t)(kWideCtx * 64); }

}  // namespace isscabac
