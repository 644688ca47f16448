// valu_rates.hip -- calibration microbenchmark: cycles per wave64 instruction on one SIMD of gfx950 for the
// integer / LDS instructions the CABAC kernels are made of, at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip && ./valu_rates
// Each kernel runs ITER x 16 instructions of one kind per wave (8 independent registers -> independent issue, or one
// register -> dependent chain) and stamps s_memtime around the loop.  Reported: shader cycles per instruction as seen
// by one SIMD (wave time / instructions / waves-per-SIMD is NOT applied: the number printed is wave-cycles per
// instruction; divide by the wave count for the SIMD-level cost).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define ITER 2048

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned long long *out, unsigned seed) {
  __shared__ unsigned lds[4096];
  unsigned x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 ^ 0x55, x5 = x0 + 9, x6 = x0 * 11, x7 = ~x0;
  unsigned long long q0 = x0, q1 = x1;
  unsigned y = seed | 1, z = threadIdx.x * 4;
  lds[threadIdx.x] = x0;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < ITER; i++) {
    if (KIND == 0) {  // independent v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
    } else if (KIND == 1) {  // dependent v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         : "+v"(x0) : "v"(y));)
    } else if (KIND == 2) {  // independent v_bfe_u32 (VOP3)
      REP16(asm volatile("v_bfe_u32 %0, %0, %8, 8\n v_bfe_u32 %1, %1, %8, 8\n v_bfe_u32 %2, %2, %8, 8\n v_bfe_u32 %3, %3, %8, 8\n"
                         "v_bfe_u32 %4, %4, %8, 8\n v_bfe_u32 %5, %5, %8, 8\n v_bfe_u32 %6, %6, %8, 8\n v_bfe_u32 %7, %7, %8, 8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
    } else if (KIND == 3) {  // independent v_lshlrev_b64
      REP16(asm volatile("v_lshlrev_b64 %0, %2, %0\n v_lshlrev_b64 %1, %2, %1\n v_lshlrev_b64 %0, %2, %0\n v_lshlrev_b64 %1, %2, %1\n"
                         "v_lshlrev_b64 %0, %2, %0\n v_lshlrev_b64 %1, %2, %1\n v_lshlrev_b64 %0, %2, %0\n v_lshlrev_b64 %1, %2, %1\n"
                         : "+v"(q0), "+v"(q1) : "v"(y));)
    } else if (KIND == 4) {  // v_lshl_add_u64
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n"
                         "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n"
                         : "+v"(q0), "+v"(q1));)
    } else if (KIND == 5) {  // v_cmp + v_cndmask pairs (vcc)
      REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n"
                         "v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %1, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");)
    } else if (KIND == 6) {  // independent v_ffbh_u32
      REP16(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3\n"
                         "v_ffbh_u32 %4, %4\n v_ffbh_u32 %5, %5\n v_ffbh_u32 %6, %6\n v_ffbh_u32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 7) {  // ds_bpermute_b32, independent, waited once per 8
      REP16(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                         "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(z));)
    } else if (KIND == 8) {  // ds_read_u8 conflict-free, independent
      REP16(asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:256\n ds_read_u8 %2, %8 offset:512\n ds_read_u8 %3, %8 offset:768\n"
                         "ds_read_u8 %4, %8 offset:1024\n ds_read_u8 %5, %8 offset:1280\n ds_read_u8 %6, %8 offset:1536\n ds_read_u8 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(z));)
    } else if (KIND == 9) {  // ds_write_b8 conflict-free
      REP16(asm volatile("ds_write_b8 %8, %0\n ds_write_b8 %8, %1 offset:256\n ds_write_b8 %8, %2 offset:512\n ds_write_b8 %8, %3 offset:768\n"
                         "ds_write_b8 %8, %4 offset:1024\n ds_write_b8 %8, %5 offset:1280\n ds_write_b8 %8, %6 offset:1536\n ds_write_b8 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(z) : "memory");)
    } else if (KIND == 10) {  // dependent ds_read_u8 -> ds_bpermute chain (latency)
      REP16(asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0\n ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n"
                         "ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n"
                         : "+v"(x0) : "v"(z));)
    } else if (KIND == 11) {  // v_cndmask with sgpr-pair condition (VOP3)
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                         "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "s"(0x5555555555555555ull));)
    } else if (KIND == 12) {  // SDWA byte-select add
      REP16(asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         "v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
    }
  }
  const unsigned long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (unsigned)q0 + (unsigned)q1 == 0x12345) out[0] = lds[z & 4095];
}

template <int KIND>
void run(const char *name, int per_instr_group) {
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
    unsigned long long *d;
    hipMalloc(&d, blocks * 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)ITER * 16 * per_instr_group;
    printf("%-34s waves/SIMD %d: wave-cycles/instr median %.2f (min %.2f)  -> SIMD cycles/instr %.2f   [%.3f ms, %.2f ns/instr/SIMD]\n", name, wps,
           h[h.size() / 2] / n, h[0] / n, h[h.size() / 2] / n / wps, ms, ms * 1e6 / n / wps);
    hipFree(d);
  }
}

int main() {
  run<0>("v_add_u32 independent", 8);
  run<1>("v_add_u32 dependent", 8);
  run<2>("v_bfe_u32 independent", 8);
  run<3>("v_lshlrev_b64 (2 chains)", 8);
  run<4>("v_lshl_add_u64 (dependent pair)", 8);
  run<5>("v_cmp+v_cndmask (vcc) pairs", 8);
  run<6>("v_ffbh_u32 independent", 8);
  run<11>("v_cndmask_b32 sgpr-cond", 8);
  run<12>("v_add_u32_sdwa", 8);
  run<7>("ds_bpermute_b32 x8 + wait", 8);
  run<8>("ds_read_u8 x8 + wait", 8);
  run<9>("ds_write_b8 x8 + wait", 8);
  run<10>("read_u8->bpermute->read_u8 chain", 3);
  return 0;
}
