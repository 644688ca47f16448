// cabac_internal.h -- launcher interface between the C ABI (cabac_capi.cpp) and the HIP kernels
// (cabac_kernels.hip).  Not installed; the public boundary is include/isscabac.h.
#ifndef ISSCABAC_INTERNAL_H
#define ISSCABAC_INTERNAL_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/isscabac.h"

// the engine handle of include/isscabac.h (host side only)
struct isscabac_engine {
  int device = 0;
  int num_cus = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  int tuning = 0;     // ISSCABAC_TUNE_*
  const char *last_instance[2] = {"", ""};  // kernel instance of the most recent batch encode / decode (isscabac_last_instance)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // stream in use (own_stream or the caller's)
  std::string err;
  // engine-owned device scratch, grow-only
  uint8_t *slots = nullptr;
  uint64_t slots_bytes = 0;
  uint32_t *lens = nullptr;
  uint64_t lens_bytes = 0;
  void *scan_tmp = nullptr;
  uint64_t scan_tmp_bytes = 0;
  uint32_t *order = nullptr;  // length bucketing of ragged batches: the sorted stream ids, and the sort's scratch
  uint64_t order_bytes = 0;
  void *order_tmp = nullptr;
  uint64_t order_tmp_cap = 0;
  // four device words, zeroed at creation: [0], [1] "some stream needs the reference-formulation decoder" (launch_decode_exact) of
  // the device-buffer decode calls, taking turns -- a call's second kernel zeroes the next call's word, no memset per call --,
  // [2] the ticket of the two-launch offsets scan (k_scan_partial_top), [3] the same flag for the host-buffer and symbol calls
  uint32_t *flag = nullptr;
  int flag_turn = 0;
  uint64_t *h_total_dev = nullptr;  // h_total as the device sees it (pinned, mapped)
  // staging for the host-buffer entry points
  void *stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // ([6]: bins as bits, packed forms)
  uint64_t stage_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
  // pinned staging ring of the host-buffer entry points (see HostPipe in cabac_capi.cpp): kPinSlots blocks of kPinBytes,
  // one event each, and two copy streams (H2D and D2H run full duplex beside the compute stream)
  static constexpr int kPinSlots = 6;
  static constexpr uint64_t kPinBytes = 16ull << 20;
  uint8_t *pin[kPinSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pin_ev[kPinSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};  // hand-over points between the copy streams and the compute stream
  // batches of more than 255 contexts (cabac_wide.hip): the 1 024 initial states
  uint8_t *wide_init = nullptr;
  // small symbol batches: one pinned, device-mapped block the kernels read and write directly (cabac_symbols.hip)
  uint8_t *sym_pin = nullptr;
  uint64_t sym_pin_bytes = 0;
  // per-bin handles opened on this engine (cabac_stream.hip): their live kernels are stopped before anything else is
  // launched for the engine -- a polling kernel must never stand in front of another kernel in a hardware queue
  std::vector<isscabac_stream *> handles;
  void *mat_bins = nullptr;  // a whole matrix as one stream: its bin / context sequence for the bin-level encoder (cabac_symbols.hip)
  uint64_t mat_bins_bytes = 0;
  size_t sym_cnt_lds = 0;  // dynamic LDS granted to the statistics kernel so far (cabac_symbols.hip: k_encode_symbols<true>)
  // what isscabac_sync has to report about the most recent (async) encode
  uint64_t *h_total = nullptr;  // pinned host word, target of an async copy of offsets[S]
  uint64_t pending_cap = 0;
  bool pending_total = false;
  // profiling: a ring of event sets, one per batch call, read back at synchronisation points
  static constexpr int kProfRing = 128;
  struct ProfSlot {
    hipEvent_t ev[4];
    int kind;  // 0 encode (ev 0..3: start, kernel end, scan end, gather end), 1 decode (ev 0..1)
  };
  bool profiling = false;
  ProfSlot *prof = nullptr;  // kProfRing slots, created on first use
  int prof_head = 0, prof_count = 0;  // next slot to record into, recorded-but-unread slots
  float last_ms[4] = {0, 0, 0, 0};
  double sum_ms[4] = {0, 0, 0, 0};
  uint32_t sum_n[4] = {0, 0, 0, 0};
  std::vector<float> samples[4];  // every launch's time since the last isscabac_kernel_ms_samples (bounded: kMaxSamples)
  static constexpr size_t kMaxSamples = 65536;
  uint64_t samples_dropped[4] = {0, 0, 0, 0};  // launches whose time did not fit (isscabac_kernel_ms_samples_dropped)
};


namespace isscabac {
int engine_fail(isscabac_engine *e, int code, const char *fmt, ...);
}

#define HIP_TRY(e, call)                                                                                         \
  do {                                                                                                           \
    hipError_t _err = (call);                                                                                    \
    if (_err != hipSuccess)                                                                                      \
      return isscabac::engine_fail(e, _err == hipErrorOutOfMemory ? ISSCABAC_ERR_NOMEM : ISSCABAC_ERR_HIP, "%s: %s", \
                                   #call, hipGetErrorString(_err));                                              \
  } while (0)

namespace isscabac {

struct BatchShape {
  uint64_t S;        // streams
  uint32_t nbins;    // bins per stream when d_nbins is null (<= pitch)
  uint32_t pitch;    // bytes between consecutive streams' rows, multiple of 16
  uint32_t nctx;     // contexts (1..255)
  bool latency;      // run the latency-oriented kernel instance (few waves per SIMD), see isscabac_set_tuning
  bool latency_enc;  // ... the encoder's: also every launch of more than 64 contexts (at most three waves per SIMD)
  bool tiny;         // at most two waves of streams per CU: the decoder + helper wave pair pays (cabac_pair.hip)
  bool pair;         // the wave-pair kernels may run (false: ISSCABAC_TUNE_LATENCY_SINGLE, one wave per 64 streams)
  const uint32_t *d_order;  // device: stream ids sorted by bin count (length-bucketed launch, cabac_order.hip) or null
  uint8_t ctx_init[256];
  mutable const char *instance;  // out: the kernel instance launch_encode / launch_decode chose (a string literal)
};

// bytes of HBM slot reserved per stream by the encode kernel (bound + dword slack, 16-byte aligned)
uint64_t slot_stride_for(uint32_t nbins);

// encode: bins/ctx -> per-stream slots + lens
hipError_t launch_encode(const BatchShape &sh, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                         uint8_t *d_slots, uint64_t slot_stride, uint32_t *d_lens, hipStream_t stream);
// lens -> offsets (exclusive scan, S+1 entries); d_tmp needs scan_tmp_bytes(S)
uint64_t scan_tmp_bytes(uint64_t S);
// d_host_total (nullable): a pinned, device-mapped word that receives offsets[S] as well -- no copy of it afterwards
hipError_t launch_scan(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, hipStream_t stream,
                       uint64_t *d_host_total = nullptr);
// medium batches (1 Ki < S <= 256 Ki, a ticket word given): scan + gather in two launches instead of four
bool scan_gather_is_paired(uint64_t S, const uint32_t *d_ticket);
hipError_t launch_scan_partial_top(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, uint32_t *d_ticket,
                                   uint64_t *d_host_total, hipStream_t stream);
hipError_t launch_gather_scan(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets,
                              void *d_tmp, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream);
// slots -> tight payload (streams whose range would exceed payload_cap are skipped)
bool scan_gather_is_fused(uint64_t S);
hipError_t launch_scan_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets,
                              void *d_tmp, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream, uint32_t *d_ticket = nullptr,
                              uint64_t *d_host_total = nullptr);
hipError_t launch_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, const uint64_t *d_offsets,
                         uint64_t S, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream);
// decode: payload/offsets/ctx -> bins (+status)
// *d_flag (zeroed by the caller) is set when a stream starts with 0xFF: such streams (corrupt input only) are
// decoded again by launch_decode_exact to reproduce the reference's 32-bit register wrap-around
hipError_t launch_decode(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                         const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out, uint32_t *d_status,
                         uint32_t *d_flag, hipStream_t stream);
// one thread per stream, reference formulation (cabac_lane.h: RefDec); touches only streams that start with 0xFF, and
// returns at once when *d_flag (written by launch_decode on the same stream) is 0
hipError_t launch_decode_exact(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes,
                               const uint64_t *d_offsets, const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out,
                               uint32_t *d_status, const uint32_t *d_flag, hipStream_t stream, uint32_t *d_next_flag = nullptr);
// ragged batches: stream ids sorted by min(d_nbins[s], nbins), longest first (cabac_order.hip); d_tmp needs order_tmp_bytes(S)
uint64_t order_tmp_bytes(uint64_t S);
hipError_t launch_order(const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t *d_order, void *d_tmp, uint64_t tmp_bytes,
                        hipStream_t stream);
// bins as bits <-> bins as bytes (rows of pitch / 8 and pitch bytes), for the packed host-buffer entry points
hipError_t launch_unpack_bits(const uint8_t *d_packed, uint8_t *d_bins, uint64_t S, uint32_t pitch, hipStream_t stream);
hipError_t launch_pack_bits(const uint8_t *d_bins, uint8_t *d_packed, uint64_t S, uint32_t pitch, hipStream_t stream);
// per-bin handles (cabac_stream.hip): make the live kernels of the engine's handles (all but `except`) leave and wait until they have
int engine_quiesce_live(isscabac_engine *e, isscabac_stream *except);
void engine_orphan_handles(isscabac_engine *e);  // isscabac_destroy with handles still open
// first statement of every entry point that enqueues work for an engine
#define ENGINE_ENTER(e)                                             \
  do {                                                              \
    HIP_TRY(e, hipSetDevice((e)->device));                          \
    if (!(e)->handles.empty()) {                                    \
      const int rcq_ = isscabac::engine_quiesce_live((e), nullptr); \
      if (rcq_) return rcq_;                                        \
    }                                                               \
  } while (0)

// batches of more than 255 contexts: 16-bit context indices, one wave per workgroup with the state image in its LDS (cabac_wide.hip)
hipError_t launch_encode_wide(const uint8_t *d_bins, const uint16_t *d_ctx, const uint32_t *d_nbins, uint64_t S, uint32_t nbins,
                              uint32_t pitch, uint32_t nctx, const uint8_t *d_ctx_init, uint8_t *d_slots, uint64_t slot_stride,
                              uint32_t *d_lens, hipStream_t stream);
hipError_t launch_decode_wide(const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets, const uint16_t *d_ctx,
                              const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx,
                              const uint8_t *d_ctx_init, uint8_t *d_bins_out, uint32_t *d_status, hipStream_t stream);
// synthetic workload generator
hipError_t launch_synth(uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx, int ep_half,
                        uint8_t *d_bins, uint8_t *d_ctx, hipStream_t stream);


// ---- pinned staging ring of the host-buffer entry points (cabac_capi.cpp, cabac_symbols.hip) -----------------------------------
void parallel_copy(void *dst, const void *src, size_t n);  // a few host threads: one memcpy stream tops out near 10 GB/s
int pipe_init(isscabac_engine *e);                          // the ring, the two copy streams and their events, once per engine
// The pinned ring, used by one direction at a time per call site.  H2D: the host fills a slot, the copy is enqueued, the
// slot's event marks it free again.  D2H: the copy into a slot is enqueued, and when its event has fired the host moves
// the block to the caller's memory.  Slots [first, first + count) of the engine's ring.
struct HostPipe {
  isscabac_engine *e;
  int first, count, next = 0;
  // pending D2H blocks (slot -> where its bytes go)
  struct Out {
    void *dst;
    size_t n;
    bool live;
  } out[isscabac_engine::kPinSlots] = {};

  int upload(void *d_dst, const void *h_src, uint64_t bytes) {
    const uint64_t B = isscabac_engine::kPinBytes;
    for (uint64_t off = 0; off < bytes; off += B) {
      const uint64_t n = std::min(B, bytes - off);
      const int slot = first + next;
      next = (next + 1) % count;
      HIP_TRY(e, hipEventSynchronize(e->pin_ev[slot]));  // the DMA that last read this slot is done
      parallel_copy(e->pin[slot], (const char *)h_src + off, n);
      HIP_TRY(e, hipMemcpyAsync((char *)d_dst + off, e->pin[slot], n, hipMemcpyHostToDevice, e->h2d_stream));
      HIP_TRY(e, hipEventRecord(e->pin_ev[slot], e->h2d_stream));
    }
    return ISSCABAC_OK;
  }
  int drain_slot(int slot) {
    if (!out[slot].live) return ISSCABAC_OK;
    HIP_TRY(e, hipEventSynchronize(e->pin_ev[slot]));
    parallel_copy(out[slot].dst, e->pin[slot], out[slot].n);
    out[slot].live = false;
    return ISSCABAC_OK;
  }
  // enqueue d_src -> h_dst on the D2H stream (which the caller has made wait for the producing kernel)
  int download(void *h_dst, const void *d_src, uint64_t bytes) {
    const uint64_t B = isscabac_engine::kPinBytes;
    for (uint64_t off = 0; off < bytes; off += B) {
      const uint64_t n = std::min(B, bytes - off);
      const int slot = first + next;
      next = (next + 1) % count;
      const int rc = drain_slot(slot);
      if (rc) return rc;
      HIP_TRY(e, hipMemcpyAsync(e->pin[slot], (const char *)d_src + off, n, hipMemcpyDeviceToHost, e->d2h_stream));
      HIP_TRY(e, hipEventRecord(e->pin_ev[slot], e->d2h_stream));
      out[slot] = Out{(char *)h_dst + off, (size_t)n, true};
    }
    return ISSCABAC_OK;
  }
  int drain() {
    for (int k = 0; k < count; k++) {  // oldest first
      const int rc = drain_slot(first + (next + k) % count);
      if (rc) return rc;
    }
    return ISSCABAC_OK;
  }
};

}  // namespace isscabac
#endif
