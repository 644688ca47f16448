// cabac_internal.h -- launcher interface between the C ABI (cabac_capi.cpp) and the HIP kernels
// (cabac_kernels.hip).  Not installed; the public boundary is include/isscabac.h.
#ifndef ISSCABAC_INTERNAL_H
#define ISSCABAC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isscabac {

struct BatchShape {
  uint64_t S;        // streams
  uint32_t nbins;    // bins per stream when d_nbins is null (<= pitch)
  uint32_t pitch;    // bytes between consecutive streams' rows, multiple of 16
  uint32_t nctx;     // contexts (1..255)
  uint8_t ctx_init[256];
};

// bytes of HBM slot reserved per stream by the encode kernel (bound + dword slack, 16-byte aligned)
uint64_t slot_stride_for(uint32_t nbins);

// encode: bins/ctx -> per-stream slots + lens
hipError_t launch_encode(const BatchShape &sh, const uint8_t *d_bins, const uint8_t *d_ctx, const uint32_t *d_nbins,
                         uint8_t *d_slots, uint64_t slot_stride, uint32_t *d_lens, hipStream_t stream);
// lens -> offsets (exclusive scan, S+1 entries); d_tmp needs scan_tmp_bytes(S)
uint64_t scan_tmp_bytes(uint64_t S);
hipError_t launch_scan(const uint32_t *d_lens, uint64_t S, uint64_t *d_offsets, void *d_tmp, hipStream_t stream);
// slots -> tight payload (streams whose range would exceed payload_cap are skipped)
hipError_t launch_gather(const uint8_t *d_slots, uint64_t slot_stride, const uint32_t *d_lens, const uint64_t *d_offsets,
                         uint64_t S, uint8_t *d_payload, uint64_t payload_cap, hipStream_t stream);
// decode: payload/offsets/ctx -> bins (+status)
hipError_t launch_decode(const BatchShape &sh, const uint8_t *d_payload, uint64_t payload_bytes, const uint64_t *d_offsets,
                         const uint8_t *d_ctx, const uint32_t *d_nbins, uint8_t *d_bins_out, uint32_t *d_status,
                         hipStream_t stream);
// synthetic workload generator
hipError_t launch_synth(uint32_t seed, uint64_t s0, uint64_t S, uint32_t nbins, uint32_t pitch, uint32_t nctx, int ep_half,
                        uint8_t *d_bins, uint8_t *d_ctx, hipStream_t stream);

}  // namespace isscabac
#endif
