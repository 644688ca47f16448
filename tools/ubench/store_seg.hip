// store_seg.hip -- how should the decode kernel's output leave?  The kernel's shape and pace are imitated: 16 384 waves of
// 64 rows (pitch 4096), four resident workgroups of four waves per CU; every "round" a wave loads 64 bytes of each of its
// rows (4 lanes x 16 B per row), idles for about the time 64 bins take to decode, and stores 64 bytes per row:
//   mode 0: a 64-byte segment per row every round                                  (4 store instructions per round)
//   mode 1: a 128-byte segment per row every second round, one instruction = 8 rows x 128 B   (8 per two rounds)
//   mode 2: every second round, the two 64-byte halves of each 128-byte line by two instructions back to back
//   mode 3: as mode 0 without the stores (the floor)
//   hipcc --offload-arch=gfx950 -O3 -o store_seg tools/ubench/store_seg.hip && ./store_seg
#include <hip/hip_runtime.h>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ void __launch_bounds__(256, 4) k(const uint8_t *in, uint8_t *out, uint32_t pitch, int sleeps, unsigned *sink) {
  extern __shared__ uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const uint8_t *inw = in + s0 * pitch;
  uint8_t *outw = out + s0 * pitch;
  unsigned acc = 0;
  uint4 keep[4] = {};
  auto st = [&](uint8_t *p, uint4 v) {
    u32x4 w = {v.x, v.y, v.z, v.w};
    if (NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = w;
  };
  for (uint32_t off = 0; off < pitch; off += 64) {
    uint4 r[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      const uint32_t row = k2 * 16 + (lane >> 2);
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(inw + row * pitch + off + (lane & 3) * 16));
      r[k2] = make_uint4(v.x, v.y, v.z, v.w);
    }
    for (int i = 0; i < sleeps; i++) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) acc += r[k2].x ^ r[k2].w;
    if (MODE == 0) {
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++) st(outw + (k2 * 16 + (lane >> 2)) * pitch + off + (lane & 3) * 16, r[k2]);
    } else if (MODE == 1 || MODE == 2) {
      if ((off & 64) == 0) {
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) keep[k2] = r[k2];
      } else if (MODE == 2) {
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
          uint8_t *p = outw + (k2 * 16 + (lane >> 2)) * pitch + (off - 64) + (lane & 3) * 16;
          st(p, keep[k2]);
          st(p + 64, r[k2]);
        }
      } else {
        // one instruction = 8 rows x 128 bytes: lane l writes piece l % 8 of row 8 k + l / 8 (data: whatever this lane holds)
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++)
          st(outw + (k2 * 8 + (lane >> 3)) * pitch + (off - 64) + (lane & 7) * 16, k2 < 4 ? keep[k2] : r[k2 - 4]);
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc + smem[lane];
}

// LSEG: bytes per row and load instruction group (64: 4 lanes per row, every round; 128: 8 lanes per row, every second round)
// NIN: input arrays (1: ctx, decoder; 2: bins + ctx, encoder); STORE: 128-byte output segments every second round
// TOUCH: every TOUCH-th round lane l also reads one dword of each of the next TOUCH - 1 lines of row l (a plain load whose
// result is dropped): the memory controller then sees TOUCH adjacent 128-byte lines of a row at a time, the demand loads of
// the following rounds find theirs in L2 / the memory-side cache
template <int LSEG, int NIN, bool STORE, int TOUCH = 0, bool NTLD = true>
__global__ void __launch_bounds__(256, 4) k2(const uint8_t *in, const uint8_t *in2, uint8_t *out, uint32_t pitch, int sleeps, unsigned *sink) {
  extern __shared__ uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  uint8_t *outw = out + s0 * pitch;
  unsigned acc = 0;
  constexpr int LPR = LSEG / 16, RPI = 64 / LPR, NI = 64 / RPI;  // lanes per row, rows per instruction, instructions per array
  for (uint32_t off = 0; off < pitch; off += LSEG) {
    uint4 r[NIN][NI];
#pragma unroll
    for (int a = 0; a < NIN; a++) {
      const uint8_t *inw = (a ? in2 : in) + s0 * pitch;
#pragma unroll
      for (int k2i = 0; k2i < NI; k2i++) {
        const uint32_t row = k2i * RPI + lane / LPR;
        const u32x4 *pp = reinterpret_cast<const u32x4 *>(inw + row * pitch + off + (lane % LPR) * 16);
        const u32x4 v = NTLD ? __builtin_nontemporal_load(pp) : *pp;
        r[a][k2i] = make_uint4(v.x, v.y, v.z, v.w);
      }
      if (TOUCH > 1 && (off / LSEG) % TOUCH == 0) {
#pragma unroll
        for (int j = 1; j < TOUCH; j++)
          if (off + j * LSEG < pitch) acc += *reinterpret_cast<const volatile uint32_t *>(inw + lane * pitch + off + j * LSEG) & 0x100u;
      }
    }
    for (int i = 0; i < sleeps * (LSEG / 64); i++) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int a = 0; a < NIN; a++)
#pragma unroll
      for (int k2i = 0; k2i < NI; k2i++) acc += r[a][k2i].x ^ r[a][k2i].w;
    if (STORE) {
#pragma unroll
      for (int rep = 0; rep < LSEG / 128 + (LSEG < 128 ? ((off & 64) ? 1 : 0) : 0); rep++)
#pragma unroll
        for (int k2i = 0; k2i < 8; k2i++) {
          u32x4 w = {acc, acc, acc, acc};
          __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(outw + (k2i * 8 + (lane >> 3)) * pitch + (off & ~127u) + (lane & 7) * 16));
        }
    }
  }
  if (acc == 0x12345678u) *sink = acc + smem[lane];
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
  const size_t lds = 36864;  // four workgroups per CU, like the coder kernels
  auto time = [&](const char *name, auto kern, int sleeps) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3((unsigned)(S / 256)), dim3(256), lds, 0, in, out, pitch, sleeps, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r) best = ms < best ? ms : best;
    }
    printf("%-58s sleeps %d: %.3f ms\n", name, sleeps, best);
  };
  uint8_t *in2;
  hipMalloc(&in2, bytes);
  hipMemset(in2, 2, bytes);
  auto time2 = [&](const char *name, auto kern, int sleeps) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3((unsigned)(S / 256)), dim3(256), lds, 0, in, in2, out, pitch, sleeps, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r) best = ms < best ? ms : best;
    }
    printf("%-58s sleeps %d: %.3f ms\n", name, sleeps, best);
  };
  for (int sleeps : {0, 3}) {
    time2("decoder-like: 1 x 64 B loads + 128 B stores", k2<64, 1, true>, sleeps);
    time2("decoder-like: 1 x 128 B loads + 128 B stores", k2<128, 1, true>, sleeps);
    time2("encoder-like: 2 x 64 B loads", k2<64, 2, false>, sleeps);
    time2("encoder-like: 2 x 128 B loads", k2<128, 2, false>, sleeps);
    time2("decoder-like: 128 B loads nt + touch 1 ahead + stores", k2<128, 1, true, 2>, sleeps);
    time2("decoder-like: 128 B loads plain + touch 1 ahead + stores", k2<128, 1, true, 2, false>, sleeps);
    time2("decoder-like: 128 B loads nt + touch 3 ahead + stores", k2<128, 1, true, 4>, sleeps);
    time2("decoder-like: 128 B loads plain + stores", k2<128, 1, true, 0, false>, sleeps);
    time2("encoder-like: 2 x 64 B loads plain", k2<64, 2, false, 0, false>, sleeps);
    time2("encoder-like: 2 x 64 B loads plain + touch 3 ahead (256 B)", k2<64, 2, false, 4, false>, sleeps);
    time2("encoder-like: 2 x 64 B loads plain + touch 7 ahead (512 B)", k2<64, 2, false, 8, false>, sleeps);
    time2("encoder-like: 2 x 128 B loads nt + touch 1 ahead", k2<128, 2, false, 2>, sleeps);
    time2("1 x 128 B loads nt + touch 1 ahead", k2<128, 1, false, 2>, sleeps);
    time2("1 x 128 B loads nt + touch 3 ahead", k2<128, 1, false, 4>, sleeps);
    time2("1 x 64 B loads", k2<64, 1, false>, sleeps);
    time2("1 x 128 B loads", k2<128, 1, false>, sleeps);
  }
  for (int sleeps : {0, 2, 3, 4}) {
    time("mode 3: loads only", k<3, true>, sleeps);
    time("mode 0: 64 B per row every round, nt", k<0, true>, sleeps);
    time("mode 0: 64 B per row every round", k<0, false>, sleeps);
    time("mode 1: 128 B per row every 2nd round, nt", k<1, true>, sleeps);
    time("mode 1: 128 B per row every 2nd round", k<1, false>, sleeps);
    time("mode 2: 2 x 64 B back to back every 2nd round, nt", k<2, true>, sleeps);
    time("mode 2: 2 x 64 B back to back every 2nd round", k<2, false>, sleeps);
  }
  return 0;
}
