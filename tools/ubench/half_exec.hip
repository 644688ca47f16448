// half_exec.hip -- does a wave64 instruction cost less when only its low 32 (16) lanes are active?  VALU (a full-rate and a
// half-rate form) and LDS (byte read, 8-byte read, byte write) with EXEC = all 64 / low 32 / low 16 lanes, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/half_exec tools/ubench/half_exec.hip && tools/ubench/half_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned seed, int iters, int active) {
  __shared__ unsigned lds[1024 * 4];
  unsigned x = threadIdx.x + seed, z = x * 3, u0 = x * 5, u1 = x * 7, y = seed | 1;
  const unsigned a0 = (threadIdx.x * 4) * 4;  // this lane's own 16 bytes
  lds[threadIdx.x * 4] = x; lds[threadIdx.x * 4 + 1] = z;
  __syncthreads();
  if ((int)(threadIdx.x & 63) < active) {  // the whole loop runs under the reduced EXEC mask
    for (int i = 0; i < iters; i++) {
      if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1) : "v"(y));) }
      if (KIND == 1) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1));) }
      if (KIND == 2) { REP16(asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:1\n ds_read_u8 %2, %4 offset:2\n ds_read_u8 %3, %4 offset:3\n s_waitcnt lgkmcnt(0)" : "=v"(x), "=v"(z), "=v"(u0), "=v"(u1) : "v"(a0) : "memory");) }
      if (KIND == 3) { unsigned long long p, q; REP16(asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(p), "=v"(q) : "v"(a0) : "memory"); x += (unsigned)p; z += (unsigned)q;) }
      if (KIND == 4) { REP16(asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:1\n ds_write_b8 %0, %1 offset:2\n ds_write_b8 %0, %1 offset:3" : : "v"(a0), "v"(x) : "memory");) }
    }
  }
  if (x + z + u0 + u1 == 0x12345) out[0] = 1;
}
template <int KIND>
static void run(const char *name, int per_iter, int active) {
  unsigned *d;
  (void)hipMalloc(&d, 64);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, threads = 1024, blocks = cus;  // 16 waves per CU = 4 per SIMD
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters, active);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters, active);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 16 * per_iter;
  printf("%-28s %2d active lanes: %8.3f ms  %.3f ns per SIMD instruction (4 waves per SIMD)\n", name, active, ms, ms * 1e6 / per_wave / 4);
  (void)hipFree(d);
}
int main() {
  for (int act : {64, 32, 16}) {
    run<0>("v_add_u32 (full rate)", 4, act);
    run<1>("v_lshlrev_b32 (half rate)", 4, act);
    run<2>("ds_read_u8", 4, act);
    run<3>("ds_read_b64", 2, act);
    run<4>("ds_write_b8", 4, act);
  }
  return 0;
}
