// tile_bw.hip -- what HBM bandwidth does the engine's access pattern get?  A wave reads SEG contiguous bytes from each
// of 64 rows that are `pitch` bytes apart (stream-major bins/ctx), round after round along the rows; optionally writes
// SEG-byte segments back the same way.  Compare SEG = 32/64/128/256 with a fully contiguous copy.
//   hipcc --offload-arch=gfx950 -O3 -o tile_bw tools/ubench/tile_bw.hip && ./tile_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SEG, bool STORE>
__global__ void __launch_bounds__(256) k_rows(const uint8_t *in, uint8_t *out, uint64_t S, uint32_t pitch, unsigned *sink) {
  const int lane = threadIdx.x & 63;
  const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (s0 >= S) return;
  constexpr int LPR = SEG / 16;      // lanes per row
  constexpr int RPI = 64 / LPR;      // rows per instruction
  constexpr int NI = 64 / RPI;       // instructions per round
  unsigned acc = 0;
  for (uint32_t off = 0; off < pitch; off += SEG) {
    uint4 r[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint64_t row = s0 + k * RPI + lane / LPR;
      r[k] = *reinterpret_cast<const uint4 *>(in + row * pitch + off + (lane % LPR) * 16);
    }
#pragma unroll
    for (int k = 0; k < NI; k++) {
      acc += r[k].x ^ r[k].y ^ r[k].z ^ r[k].w;
      if (STORE) {
        const uint64_t row = s0 + k * RPI + lane / LPR;
        r[k].x += 1;
        *reinterpret_cast<uint4 *>(out + row * pitch + off + (lane % LPR) * 16) = r[k];
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <bool STORE>
__global__ void __launch_bounds__(256) k_linear(const uint4 *in, uint4 *out, uint64_t n, unsigned *sink) {
  unsigned acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    uint4 v = in[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
    if (STORE) {
      v.x += 1;
      out[i] = v;
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const uint64_t S = 1 << 20;
  const uint32_t pitch = 4096;
  const uint64_t bytes = S * pitch;
  uint8_t *in, *out;
  unsigned *sink;
  hipMalloc(&in, bytes);
  hipMalloc(&out, bytes);
  hipMalloc(&sink, 4);
  hipMemset(in, 1, bytes);
  hipMemset(out, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, double moved, auto launch) {
    launch();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, moved / best / 1e9);
  };
  const unsigned nb = (unsigned)(S / 256);
#define ROWS(SEG)                                                                                                     \
  time("rows: load " #SEG " B segments", (double)bytes, [&] { hipLaunchKernelGGL((k_rows<SEG, false>), dim3(nb), dim3(256), 0, 0, in, out, S, pitch, sink); }); \
  time("rows: load+store " #SEG " B segments", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_rows<SEG, true>), dim3(nb), dim3(256), 0, 0, in, out, S, pitch, sink); });
  ROWS(32)
  ROWS(64)
  ROWS(128)
  ROWS(256)
  time("linear: load", (double)bytes, [&] { hipLaunchKernelGGL((k_linear<false>), dim3(8192), dim3(256), 0, 0, (const uint4 *)in, (uint4 *)out, bytes / 16, sink); });
  time("linear: load+store", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_linear<true>), dim3(8192), dim3(256), 0, 0, (const uint4 *)in, (uint4 *)out, bytes / 16, sink); });
  return 0;
}
