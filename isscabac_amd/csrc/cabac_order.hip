// cabac_order.hip -- length bucketing for ragged batches.
//
// One wavefront codes 64 streams in lockstep and runs until its longest one is done, so a batch whose streams differ in
// length (d_nbins given) wastes the lanes of the short streams and sends every chunk past the wave's shortest stream down
// the masked general path.  Streams are independent: which 64 of them share a wave is free.  The batch calls therefore
// sort the stream ids by bin count (longest first) and hand the kernels that order; lane l of wave w then codes stream
// order[64 w + l], its neighbours are streams of (nearly) the same length, and the results -- slots, lengths, decoded
// rows, status words -- are written per stream id exactly as without the order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "cabac_internal.h"

namespace isscabac {

__global__ void __launch_bounds__(256) k_order_keys(const uint32_t *nbins_per_stream, uint32_t S, uint32_t nbins, uint32_t *keys,
                                                    uint32_t *ids) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S) return;
  keys[i] = min(nbins_per_stream[i], nbins);
  ids[i] = i;
}

uint64_t order_tmp_bytes(uint64_t S) {
  size_t tmp = 0;
  uint32_t *p = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp, p, p, p, p, (int)S, 0, 32, nullptr);
  // layout: keys | keys_out | ids | cub temp
  return 3 * ((S * sizeof(uint32_t) + 255) & ~255ull) + tmp + 256;
}

hipError_t launch_order(const uint32_t *d_nbins, uint64_t S, uint32_t nbins, uint32_t *d_order, void *d_tmp, uint64_t tmp_bytes,
                        hipStream_t stream) {
  const uint64_t part = (S * sizeof(uint32_t) + 255) & ~255ull;
  uint8_t *b = static_cast<uint8_t *>(d_tmp);
  uint32_t *keys = reinterpret_cast<uint32_t *>(b), *keys_out = reinterpret_cast<uint32_t *>(b + part),
           *ids = reinterpret_cast<uint32_t *>(b + 2 * part);
  size_t cub_bytes = (size_t)(tmp_bytes - 3 * part);
  hipLaunchKernelGGL(k_order_keys, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, d_nbins, (uint32_t)S, nbins, keys, ids);
  int end_bit = 1;
  while (end_bit < 32 && (nbins >> end_bit) != 0) end_bit++;  // the keys are <= nbins
  return hipcub::DeviceRadixSort::SortPairsDescending(b + 3 * part, cub_bytes, keys, keys_out, ids, d_order, (int)S, 0, end_bit, stream);
}

}  // namespace isscabac
