// lds_tp.hip -- LDS instruction throughput of a CU (all four SIMDs share one LDS pipe) by access width and pattern.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_tp tools/ubench/lds_tp.hip && tools/ubench/lds_tp
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
// KIND 0: ds_read_u8, lane-private banks (address = lane * 4 + const)   1: ds_read_b32 same   2: ds_read_b64 (lane * 8)
//      3: ds_read_b128 (lane * 16)   4: ds_write_b8 lane-private   5: ds_read_u8, random byte of a 256-byte table
//      6: ds_read_b32, random dword of a 256-byte table   7: ds_read_b64, random 8-byte entry of a 1 KiB table
//      8: ds_read_u8 lane-private + 4 VALU per read (does LDS issue overlap VALU issue?)
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned seed, int iters) {
  extern __shared__ unsigned smem[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) smem[i] = i * 2654435761u;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned r = (lane * 2654435761u + seed) >> 7;
  unsigned a = KIND == 2 ? lane * 8 : KIND == 3 ? lane * 16 : KIND == 5 ? (r & 255) : KIND == 6 ? (r & 252) : KIND == 7 ? (r & 1016) : lane * 4;
  a += (KIND >= 5 && KIND <= 7) ? 0 : wave * 1024;
  unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0, x = seed, y = 1;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0 || KIND == 5) { REP8(asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:1\n ds_read_u8 %2, %4 offset:2\n ds_read_u8 %3, %4 offset:3\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");) }
    if (KIND == 1 || KIND == 6) { REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");) }
    if (KIND == 2 || KIND == 7) { unsigned long long w0, w1, w2, w3; REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a) : "memory");) v0 += (unsigned)w0; }
    if (KIND == 3) { uint4 w0, w1, w2, w3; REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a) : "memory");) v0 += w0.x; }
    if (KIND == 4) { REP8(asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:1\n ds_write_b8 %0, %1 offset:2\n ds_write_b8 %0, %1 offset:3\n s_waitcnt lgkmcnt(0)" : : "v"(a), "v"(x) : "memory");) }
    if (KIND == 8) { REP8(asm volatile("ds_read_u8 %0, %4\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n ds_read_u8 %1, %4 offset:1\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n ds_read_u8 %2, %4 offset:2\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n ds_read_u8 %3, %4 offset:3\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %5, %5, %6\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a), "v"(x), "v"(y) : "memory");) }
  }
  if (v0 + v1 + v2 + v3 + x == 0x12345) out[0] = 1;
}
template <int KIND>
static void run(const char *name, int waves_per_cu) {
  unsigned *d;
  (void)hipMalloc(&d, 64);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, threads = 64 * waves_per_cu;
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 100 * 1024, 0, d, 3u, iters);  // 100 KiB: one workgroup per CU
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 100 * 1024, 0, d, 3u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 32;  // LDS instructions a wave executes
  printf("%-44s %2d waves/CU: %7.2f ms  %6.2f ns per LDS instr of a wave, %5.2f ns per LDS instr of the CU\n", name, waves_per_cu, ms,
         ms * 1e6 / per_wave, ms * 1e6 / per_wave / waves_per_cu);
  (void)hipFree(d);
}
int main() {
  for (int w : {4, 8, 16}) {
    run<0>("ds_read_u8 lane-private bank", w);
    run<1>("ds_read_b32 lane-private bank", w);
    run<2>("ds_read_b64 lane * 8", w);
    run<3>("ds_read_b128 lane * 16", w);
    run<4>("ds_write_b8 lane-private bank", w);
    run<5>("ds_read_u8 random in 256 B table", w);
    run<6>("ds_read_b32 random in 256 B table", w);
    run<7>("ds_read_b64 random in 1 KiB table", w);
    run<8>("ds_read_u8 + 4 VALU each", w);
  }
  return 0;
}
