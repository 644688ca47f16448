// valu_half.hip -- does a wave64 VALU instruction whose EXEC has only lanes 0..31 set issue in one pass on gfx950?
// (If it did, a 32-streams-per-wave coder instance would give launches of <= 128 Ki streams four waves per SIMD at the
// same bins per SIMD.)  Independent streams of a full-rate op (v_add_u32), a half-rate-class op (v_lshlrev_b32) and an LDS
// read, with EXEC = all 64 lanes, lanes 0..31, lanes 0..15 and even lanes, at 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_half tools/ubench/valu_half.hip && tools/ubench/valu_half
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned seed, int iters, unsigned long long mask) {
  __shared__ unsigned lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned x = threadIdx.x + seed, z = x * 3, u0 = x * 5, u1 = x * 7, y = seed | 1, a = (threadIdx.x & 63) * 4;
  const unsigned long long keep = __builtin_amdgcn_read_exec();
  asm volatile("s_mov_b64 exec, %0" : : "s"(mask));
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1) : "v"(y));) }
    if (KIND == 1) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1));) }
    if (KIND == 2) { REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n s_waitcnt lgkmcnt(0)" : "=v"(x), "=v"(z), "=v"(u0), "=v"(u1) : "v"(a));) }
  }
  asm volatile("s_mov_b64 exec, %0" : : "s"(keep));
  if (x + z + u0 + u1 == 0x12345) out[0] = 1;
}
template <int KIND>
static void run(const char *name, int waves_per_simd, unsigned long long mask, const char *mname) {
  unsigned *d;
  (void)hipMalloc(&d, 64);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int blocks = cus * (256 * waves_per_simd / threads);
  const int iters = 50000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters, mask);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 3u, iters, mask);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 64;
  printf("%-14s exec %-10s %d waves/SIMD: %7.2f ms  %.3f ns per wave instruction, %.3f ns per SIMD instruction\n", name, mname,
         waves_per_simd, ms, ms * 1e6 / per_wave, ms * 1e6 / per_wave / waves_per_simd);
  (void)hipFree(d);
}
int main() {
  const unsigned long long masks[4] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x5555555555555555ull};
  const char *names[4] = {"all 64", "lanes 0-31", "lanes 0-15", "even lanes"};
  for (int m = 0; m < 4; m++)
    for (int w : {1, 2, 4}) {
      run<0>("v_add_u32", w, masks[m], names[m]);
      run<1>("v_lshlrev_b32", w, masks[m], names[m]);
      run<2>("ds_read_b32", w, masks[m], names[m]);
    }
  return 0;
}
