// lat_chain.hip -- latencies a single wave sees on gfx950: dependent VALU, dependent LDS reads, and the encoder's state step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lat_chain tools/ubench/lat_chain.hip && tools/ubench/lat_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITER 256
#define REP16(X) X X X X X X X X X X X X X X X X

// one wave per workgroup (WAVES = 1) or two waves on a SIMD: 8 waves of 64 = 512 threads -> 2 per SIMD
template <int KIND>
__global__ void k(unsigned long long *out, unsigned seed) {
  extern __shared__ unsigned smem[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = (i * 4 * 7 + 64) & 0x3ffc;  // a byte address, dword aligned
  __syncthreads();
  unsigned x = (threadIdx.x * 4 + seed) & 0x3ffc, y = seed | 1, z = 0, u0 = 0, u1 = 0;
  const unsigned lane4 = (threadIdx.x & 63) * 4;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < ITER; i++) {
    if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (KIND == 1) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : : "memory");) }
    if (KIND == 2) { REP16(asm volatile("ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : : "memory");) }
    if (KIND == 3) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3ffc, %0" : "+v"(x) : : "memory");) }
    // read + 3 dependent VALU (the state step's chain)
    if (KIND == 4) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %0, vcc\n v_or_b32 %0, %0, %2" : "+v"(x) : "v"(y), "v"(z) : "memory", "vcc");) }
    // the same with 16 independent VALU between the read and the wait
    if (KIND == 5) { REP16(asm volatile("ds_read_b32 %0, %0\n" REP16("v_add_u32 %1, %1, %2\n") "s_waitcnt lgkmcnt(0)\n v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %0, vcc\n v_and_b32 %0, 0x3ffc, %0" : "+v"(x), "+v"(y) : "v"(z) : "memory", "vcc");) }
    // read, then 3 more LDS operations queued behind it, then the wait for the first only
    if (KIND == 6) { REP16(asm volatile("ds_read_b32 %0, %0\n ds_write_b8 %5, %2 offset:16384\n ds_read_u8 %3, %5 offset:16384\n ds_read_b32 %4, %5 offset:16388\n s_waitcnt lgkmcnt(3)\n v_and_b32 %0, 0x3ffc, %0" : "+v"(x), "+v"(y), "+v"(z), "=v"(u0), "=v"(u1) : "v"(lane4) : "memory");) }
    if (KIND == 8) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x), "+v"(z), "+v"(u0), "+v"(u1) : "v"(y));) }
    if (KIND == 9) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x), "+v"(z) : "v"(y));) }
    if (KIND == 7) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(x) : "v"(y));) }
  }
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  if (x + y + z + u0 + u1 == 0x12345) out[0] = 1;
}

template <int KIND>
static void run(const char *name, int threads, int per_iter) {
  unsigned long long *d;
  const int blocks = 256;
  hipMalloc(&d, blocks * 8 * 8);
  for (int r = 0; r < 40; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 32768, 0, d, 3u);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-46s %3d threads: %6.1f ns per step (median; wall_clock64 = 100 MHz)\n", name, threads, 10.0 * (double)h[h.size() / 2] / (ITER * 16.0));
  hipFree(d);
}
int main() {
  // ns per step; a dependent v_add_u32 is 4 shader cycles, which gives the clock
  for (int threads : {64, 512}) {
    run<8>("4 independent v_add_u32 (per 4)", threads, 1);
    run<9>("2 independent v_add_u32 (per 2)", threads, 1);
    run<0>("dependent v_add_u32", threads, 1);
    run<7>("dependent v_add_u32 + s_nop 0", threads, 1);
    run<1>("dependent ds_read_b32", threads, 1);
    run<2>("dependent ds_read_u8", threads, 1);
    run<3>("dependent ds_read_b32 + v_and", threads, 1);
    run<4>("ds_read_b32 + cmp + cndmask + or", threads, 1);
    run<5>("ds_read_b32 + 16 indep VALU + cmp,cndmask,and", threads, 1);
    run<6>("ds_read_b32 + 3 queued LDS ops + and", threads, 1);
  }
  return 0;
}
