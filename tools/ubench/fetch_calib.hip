// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the coder kernels' request sizes?  Pure streaming reads
// of a buffer far larger than the Infinity Cache, every byte exactly once, so the true byte count is known:
//   seg64     4 lanes x 16 B per row and instruction: 64-byte row segments (k_encode's bins / ctx tiles, TileT<4>)
//   seg128    8 lanes x 16 B: 128-byte row segments (k_decode's ctx tile, TileT<8>)
//   lane16    every lane reads 16-byte pieces of ITS OWN row, one piece per instruction (k_decode's WindowSrc payload feed):
//             each 128-byte line is touched by eight consecutive instructions of one lane
//   wide      64 lanes x 16 B contiguous: 1 KiB per wave instruction (the guide's calibrated pattern: counter = 1/2)
// and the same four as stores (WRITE_SIZE).  Rows have the coder kernels' 4 KiB pitch; 16 384 waves x 64 rows = 4 GiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kPitch = 4096;
template <int MODE, bool NT, bool ST>
__global__ void __launch_bounds__(256, 4) k_calib(uint8_t *buf, unsigned *sink) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  uint8_t *rows = buf + w * 64 * kPitch;
  unsigned acc = 0;
  auto go = [&](uint8_t *p) {
    if (ST) {
      u32x4 v = {acc, (unsigned)lane, 3u, 4u};
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
      else *reinterpret_cast<u32x4 *>(p) = v;
    } else {
      const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)) : *reinterpret_cast<const u32x4 *>(p);
      acc += v.x ^ v.w;
    }
  };
  if (MODE == 0) {
    for (uint32_t off = 0; off < kPitch; off += 64)
#pragma unroll
      for (int k = 0; k < 4; k++) go(rows + (k * 16 + (lane >> 2)) * kPitch + off + (lane & 3) * 16);
  } else if (MODE == 1) {
    for (uint32_t off = 0; off < kPitch; off += 128)
#pragma unroll
      for (int k = 0; k < 8; k++) go(rows + (k * 8 + (lane >> 3)) * kPitch + off + (lane & 7) * 16);
  } else if (MODE == 2) {
    for (uint32_t off = 0; off < kPitch; off += 16) go(rows + lane * kPitch + off);
  } else if (MODE == 3) {
    for (uint32_t off = 0; off < 64 * kPitch; off += 1024) go(rows + off + lane * 16);
  } else if (MODE == 4) {  // seg64, but only the FIRST 64-byte half of every 128-byte line: what does a 64-byte miss fetch?
    for (uint32_t off = 0; off < kPitch; off += 128)
#pragma unroll
      for (int k = 0; k < 4; k++) go(rows + (k * 16 + (lane >> 2)) * kPitch + off + (lane & 3) * 16);
  } else if (MODE == 5) {  // 32-byte pieces (2 lanes x 16 B per row), only the first 32 bytes of every 128-byte line
    for (uint32_t off = 0; off < kPitch; off += 128)
#pragma unroll
      for (int k = 0; k < 2; k++) go(rows + (k * 32 + (lane >> 1)) * kPitch + off + (lane & 1) * 16);
  } else if (MODE == 6) {  // lane16, only the first 16 bytes of every 128-byte line
    for (uint32_t off = 0; off < kPitch; off += 128) go(rows + lane * kPitch + off);
  } else if (MODE == 7) {  // lane16, only the first 16 bytes of every 64-byte half line
    for (uint32_t off = 0; off < kPitch; off += 64) go(rows + lane * kPitch + off);
  } else {  // MODE 8: seg64, both halves of every line but the second one a whole sweep later (an L2 lifetime apart)
    for (int half = 0; half < 2; half++)
      for (uint32_t off = 0; off < kPitch; off += 128)
#pragma unroll
        for (int k = 0; k < 4; k++) go(rows + (k * 16 + (lane >> 2)) * kPitch + off + half * 64 + (lane & 3) * 16);
  }
  if (!ST && acc == 0x12345678u) *sink = acc;
}
template <int MODE, bool NT, bool ST>
static void run(const char *name, uint8_t *buf, unsigned *sink, uint32_t waves, double useful = 1.0) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_calib<MODE, NT, ST>), dim3(waves / 4), dim3(256), 0, 0, buf, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)waves * 64 * kPitch * useful;  // bytes the lanes asked for
  printf("%-34s %.3f ms  %.3f GB requested  %.2f TB/s of requested bytes\n", name, ms, bytes / 1e9, bytes / ms / 1e9);
}
int main() {
  const uint32_t waves = 16384;
  uint8_t *buf;
  unsigned *sink;
  if (hipMalloc(&buf, (size_t)waves * 64 * kPitch) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, (size_t)waves * 64 * kPitch);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {  // the second pass of every pattern is the one to read (same template = same kernel name)
    run<0, false, false>("load seg64 plain", buf, sink, waves);
    run<0, true, false>("load seg64 nt", buf, sink, waves);
    run<1, false, false>("load seg128 plain", buf, sink, waves);
    run<1, true, false>("load seg128 nt", buf, sink, waves);
    run<2, false, false>("load lane16 plain", buf, sink, waves);
    run<2, true, false>("load lane16 nt", buf, sink, waves);
    run<3, false, false>("load wide plain", buf, sink, waves);
    run<3, true, false>("load wide nt", buf, sink, waves);
    run<0, false, true>("store seg64 plain", buf, sink, waves);
    run<1, true, true>("store seg128 nt", buf, sink, waves);
    run<2, false, true>("store lane16 plain", buf, sink, waves);
    run<3, false, true>("store wide plain", buf, sink, waves);
    // what a miss fetches: pieces of a line only.  If the time follows the lines touched (4.3 GB) and not the bytes asked
    // for, the memory side moves whole 128-byte lines whatever the request
    run<4, false, false>("load first 64 B of each line", buf, sink, waves, 0.5);
    run<4, true, false>("load first 64 B of each line nt", buf, sink, waves, 0.5);
    run<5, false, false>("load first 32 B of each line", buf, sink, waves, 0.25);
    run<6, false, false>("load first 16 B of each line", buf, sink, waves, 0.125);
    run<7, false, false>("load first 16 B of each 64 B", buf, sink, waves, 0.25);
    run<8, false, false>("load seg64 halves a sweep apart", buf, sink, waves);
  }
  return 0;
}
