// Where do the waves of small workgroups land?  Workgroups of W waves (LDS-limited to G per CU) spin for ~200 us; every wave counts
// itself in and out of a per-(CU, SIMD) counter; printed: how many waves sat on each of a CU's four SIMDs at the peak (averaged
// over the CUs) and the most any SIMD held.   hipcc --offload-arch=gfx950 -O2 simd_place.hip -o simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ unsigned char smem[];
__global__ void k_place(unsigned *cnt, unsigned *peak, unsigned lds_bytes, long long spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = ((xcc & 0xF) << 7) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xF);  // xcc | se | cu
  const unsigned simd = (hw >> 4) & 3;
  if ((threadIdx.x & 63) == 0) {
    smem[lds_bytes - 1] = 1;
    const unsigned now = atomicAdd(&cnt[cu * 4 + simd], 1u) + 1;
    atomicMax(&peak[cu * 4 + simd], now);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    atomicSub(&cnt[cu * 4 + simd], 1u);
  }
}
int main() {
  const int N = 2048 * 4;
  unsigned *cnt, *peak;
  (void)hipMalloc(&cnt, N * 4);
  (void)hipMalloc(&peak, N * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_place), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct Case { int waves, wgs_per_cu; };
  for (Case c : {Case{1, 12}, Case{1, 11}, Case{1, 9}, Case{2, 6}, Case{2, 4}, Case{2, 8}, Case{3, 4}, Case{3, 3}, Case{4, 3}, Case{4, 2}, Case{4, 4}}) {
    const unsigned gran = 128 / c.wgs_per_cu, lds = gran * 1280;
    (void)hipMemset(cnt, 0, N * 4);
    (void)hipMemset(peak, 0, N * 4);
    hipLaunchKernelGGL(k_place, dim3(256 * c.wgs_per_cu * 6), dim3(64 * c.waves), lds, 0, cnt, peak, lds, 20000LL);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(N);
    (void)hipMemcpy(h.data(), peak, N * 4, hipMemcpyDeviceToHost);
    double sum[4] = {0, 0, 0, 0};
    unsigned mx = 0, cus = 0;
    for (int cu = 0; cu < N / 4; cu++) {
      if (!(h[cu * 4] | h[cu * 4 + 1] | h[cu * 4 + 2] | h[cu * 4 + 3])) continue;
      cus++;
      for (int s = 0; s < 4; s++) { sum[s] += h[cu * 4 + s]; if (h[cu * 4 + s] > mx) mx = h[cu * 4 + s]; }
    }
    printf("%d-wave workgroups, %2d per CU (%2d waves): peak waves on SIMD 0..3, mean over %u CUs: %.2f %.2f %.2f %.2f; most on one SIMD: %u\n",
           c.waves, c.wgs_per_cu, c.waves * c.wgs_per_cu, cus, sum[0] / cus, sum[1] / cus, sum[2] / cus, sum[3] / cus, mx);
  }
  return 0;
}
