// How many single-wave workgroups with L bytes of dynamic LDS does a CU hold?  The runtime's answer
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) and a measured one: every workgroup bumps a per-CU counter on entry, spins ~200 us,
// and the maximum a CU's counter reached is what was resident at once.   hipcc --offload-arch=gfx950 -O2 lds_occ.hip -o lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern __shared__ unsigned char smem[];
__global__ void __launch_bounds__(64) k_occ(unsigned *cnt, unsigned *peak, unsigned lds_bytes, long long spin) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = ((xcc & 0xF) << 8) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xF);  // xcc | se | cu
  if (threadIdx.x == 0) {
    smem[lds_bytes - 1] = 1;
    const unsigned now = atomicAdd(&cnt[cu], 1u) + 1;
    atomicMax(&peak[cu], now);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    atomicSub(&cnt[cu], 1u);
  }
}
int main() {
  unsigned *cnt, *peak;
  hipMalloc(&cnt, 4096 * 4);
  hipMalloc(&peak, 4096 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_occ), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (unsigned lds = 9 * 1024; lds <= 24 * 1024; lds += 256) {
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_occ, 64, lds);
    hipMemset(cnt, 0, 4096 * 4);
    hipMemset(peak, 0, 4096 * 4);
    hipLaunchKernelGGL(k_occ, dim3(256 * 40), dim3(64), lds, 0, cnt, peak, lds, 20000LL);  // 100 MHz clock: 200 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(4096);
    hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    unsigned mx = 0, cus = 0;
    for (unsigned v : h) { if (v) cus++; if (v > mx) mx = v; }
    printf("lds %6u B: runtime says %2d workgroups per CU, measured peak %2u on %u CUs (160 KiB / lds = %.2f)\n", lds, api, mx, cus, 163840.0 / lds);
  }
  return 0;
}
