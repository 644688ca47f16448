// valu_tp.hip -- VALU issue rate of a SIMD against the number of waves on it (1, 2, 4, 8), independent and dependent streams.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_tp tools/ubench/valu_tp.hip && tools/ubench/valu_tp
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned seed, int iters) {
  unsigned x = threadIdx.x + seed, z = x * 3, u0 = x * 5, u1 = x * 7, y = seed | 1;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1) : "v"(y));) }
    if (KIND == 1) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (KIND == 2) { REP16(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %2\n s_add_u32 s21, s21, 1" : "+v"(x), "+v"(z) : "v"(y) : "s20", "s21", "scc");) }
  }
  if (x + z + u0 + u1 == 0x12345) out[0] = 1;
}
template <int KIND>
static void run(const char *name, int waves_per_simd) {
  unsigned *d;
  (void)hipMalloc(&d, 64);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int blocks = cus * (256 * waves_per_simd / threads);
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 64;  // instructions (VALU + SALU) a wave executes
  printf("%-34s %d waves/SIMD: %7.2f ms  %.3f ns per wave instruction, %.3f ns per SIMD instruction\n", name, waves_per_simd, ms,
         ms * 1e6 / per_wave, ms * 1e6 / per_wave / waves_per_simd);
  (void)hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("4 independent v_add_u32 streams", w);
    run<1>("1 dependent v_add_u32 stream", w);
    run<2>("v_add / s_add alternating", w);
  }
  return 0;
}
