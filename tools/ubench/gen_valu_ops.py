#!/usr/bin/env python3
"""Generates tools/ubench/valu_ops.hip: the issue cost of one instruction kind per kernel, eight independent streams per
wave, at 2 and 8 waves per SIMD (one wave alone never issues faster than every ~2.2 ns, which hides the differences)."""
OPS = [
    ("v_mov_b32", "v_mov_b32 {d}, {y}"),
    ("v_and_b32 e32", "v_and_b32 {d}, {d}, {y}"),
    ("v_and_b32 literal", "v_and_b32 {d}, 0x3f3f00, {d}"),
    ("v_xor_b32 e32", "v_xor_b32 {d}, {d}, {y}"),
    ("v_lshlrev_b32 imm", "v_lshlrev_b32 {d}, 3, {d}"),
    ("v_lshlrev_b32 vreg", "v_lshlrev_b32 {d}, {y}, {d}"),
    ("v_sub_u32 e32", "v_sub_u32 {d}, {d}, {y}"),
    ("v_subrev_u32 imm", "v_subrev_u32 {d}, 23, {d}"),
    ("v_cndmask_b32 e32 vcc", "v_cndmask_b32 {d}, {d}, {y}, vcc"),
    ("v_cndmask_b32 e64 sgpr", "v_cndmask_b32_e64 {d}, {d}, {y}, s[20:21]"),
    ("v_cndmask_b32 e64 vcc", "v_cndmask_b32_e64 {d}, {d}, {y}, vcc"),
    ("v_cmp_lt_u32 e32", "v_cmp_lt_u32 vcc, {d}, {y}"),
    ("v_cmp + v_cndmask e32 (pair)", "v_cmp_lt_u32 vcc, {d}, {y}\\n v_cndmask_b32 {d}, {d}, {y}, vcc"),
    ("v_cmp e64 + v_cndmask e64 (pair)", "v_cmp_lt_u32_e64 s[20:21], {d}, {y}\\n v_cndmask_b32_e64 {d}, {d}, {y}, s[20:21]"),
    ("v_sub_co + 2 v_cndmask e32 (triple)", "v_sub_co_u32_e32 {d}, vcc, {d}, {y}\\n v_cndmask_b32 {d}, {d}, {y}, vcc\\n v_cndmask_b32 {d}, {y}, {d}, vcc"),
    ("v_cmp_eq_u32 e64 sgpr", "v_cmp_eq_u32_e64 s[20:21], {d}, {y}"),
    ("v_sub_co_u32 e32", "v_sub_co_u32_e32 {d}, vcc, {d}, {y}"),
    ("v_addc_co_u32 e32", "v_addc_co_u32_e32 {d}, vcc, 0, {d}, vcc"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {y}, {y}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {d}, {y}, {y}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 2, {y}"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {d}, {y}, {y}"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {y}, {d}, {y}"),
    ("v_bfe_u32 imm", "v_bfe_u32 {d}, {d}, 8, 8"),
    ("v_bfe_i32 imm", "v_bfe_i32 {d}, {d}, 1, 1"),
    ("v_perm_b32", "v_perm_b32 {d}, {d}, {y}, {y}"),
    ("v_perm_b32 (x, 0, sel)", "v_perm_b32 {d}, {d}, 0, {y}"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {d}, {y}, 2"),
    ("v_ffbh_u32", "v_ffbh_u32 {d}, {d}"),
    ("v_and_b32 sdwa src byte", "v_and_b32_sdwa {d}, {d}, {y} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_add_u32 sdwa src word", "v_add_u32_sdwa {d}, {d}, {y} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("v_lshrrev_b32 sdwa dst byte", "v_lshrrev_b32_sdwa {d}, {y}, {d} dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_cndmask_b32 sdwa", "v_cndmask_b32_sdwa {d}, {d}, {y}, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"),
    ("v_lshlrev_b64", "v_lshlrev_b64 {D}, {y}, {D}"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {D}, {D}, 0, {D}"),
    ("s_and_b32", "s_and_b32 s20, s20, s21"),
    ("s_nop 0", "s_nop 0"),
    ("s_waitcnt lgkmcnt(0)", "s_waitcnt lgkmcnt(0)"),
]
# second list (round 4): candidates for replacing half-rate instructions of the coder steps -- 24-bit multiplies as shifts by
# constants, min / max, packed 16-bit forms, DPP moves, float conversions as byte extract / normaliser
OPS2 = [
    ("v_add_u32 e32 (reference)", "v_add_u32 {d}, {d}, {y}"),
    ("v_lshlrev_b32 imm (reference)", "v_lshlrev_b32 {d}, 3, {d}"),
    ("v_mul_u32_u24 imm", "v_mul_u32_u24 {d}, 4, {d}"),
    ("v_mul_u32_u24 vreg", "v_mul_u32_u24 {d}, {y}, {d}"),
    ("v_mul_i32_i24 imm", "v_mul_i32_i24 {d}, 4, {d}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, 4, {y}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {y}"),
    ("v_min_u32", "v_min_u32 {d}, {d}, {y}"),
    ("v_max_u32", "v_max_u32 {d}, {d}, {y}"),
    ("v_max_i32", "v_max_i32 {d}, {d}, {y}"),
    ("v_med3_u32", "v_med3_u32 {d}, {d}, {y}, {y}"),
    ("v_not_b32", "v_not_b32 {d}, {d}"),
    ("v_or_b32 e32", "v_or_b32 {d}, {d}, {y}"),
    ("v_or3_b32", "v_or3_b32 {d}, {d}, {y}, {y}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {d}, 2, {y}"),
    ("v_xad_u32", "v_xad_u32 {d}, {d}, {y}, {y}"),
    ("v_ashrrev_i32 imm", "v_ashrrev_i32 {d}, 3, {d}"),
    ("v_lshrrev_b32 imm", "v_lshrrev_b32 {d}, 3, {d}"),
    ("v_lshlrev_b16 imm", "v_lshlrev_b16 {d}, 3, {d}"),
    ("v_add_u16", "v_add_u16 {d}, {d}, {y}"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, {y}"),
    ("v_pk_sub_u16", "v_pk_sub_u16 {d}, {d}, {y}"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {d}, {y}, {d}"),
    ("v_pk_max_u16", "v_pk_max_u16 {d}, {d}, {y}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {d}, {y}, {y}"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 dpp quad_perm", "v_add_u32_dpp {d}, {d}, {y} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 {d}, {d}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {d}"),
    ("v_frexp_exp_i32_f32", "v_frexp_exp_i32_f32 {d}, {d}"),
    ("v_bfm_b32", "v_bfm_b32 {d}, {d}, {y}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {d}, {y}"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 {d}, {d}, {y}"),
    ("v_mov_b32 sdwa sext byte", "v_mov_b32_sdwa {d}, sext({d}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2"),
    ("v_and_b32 sdwa sext byte", "v_and_b32_sdwa {d}, {y}, sext({d}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"),
    ("v_sub_u32 sdwa src byte", "v_sub_u32_sdwa {d}, {d}, {y} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, {d}"),
    ("v_add_co_u32 e32", "v_add_co_u32_e32 {d}, vcc, {d}, {y}"),
    ("v_subb_co_u32 e32", "v_subb_co_u32_e32 {d}, vcc, {d}, {y}, vcc"),
    ("v_cmp_ne_u32 sdwa vcc", "v_cmp_ne_u32_sdwa vcc, {y}, {d} src0_sel:DWORD src1_sel:BYTE_2"),
    ("v_alignbyte_b32", "v_alignbyte_b32 {d}, {d}, {y}, {y}"),
    ("v_sad_u32", "v_sad_u32 {d}, {d}, {y}, {y}"),
    ("v_lshlrev_b32 + v_add (2 ops)", "v_lshlrev_b32 {d}, 2, {d}\\n v_add_u32 {d}, {d}, {y}"),
    ("s_lshl_b32", "s_lshl_b32 s20, s20, 1"),
    ("s_bfe_u32", "s_bfe_u32 s20, s20, 0x80008"),
    ("s_cmp_lt_u32", "s_cmp_lt_u32 s20, s21"),
    ("s_cselect_b32", "s_cselect_b32 s20, s20, s21"),
    ("s_mov_b32", "s_mov_b32 s20, s21"),
]
import sys
OUT = "valu_ops.hip"
if len(sys.argv) > 1 and sys.argv[1] == "2":
    OPS, OUT = OPS2, "valu_ops2.hip"
HEAD = r'''// valu_ops.hip -- generated by tools/ubench/gen_valu_ops.py: per-instruction issue cost on gfx950.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define ITER 1024
#define REP16(X) X X X X X X X X X X X X X X X X
'''
KERNEL = r'''
__global__ void __launch_bounds__(256) k%d(unsigned *out, unsigned seed) {
  unsigned x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 ^ 0x55, x5 = x0 + 9, x6 = x0 * 11, x7 = ~x0;
  unsigned long long w0 = x0, w1 = x1, w2 = x2, w3 = x3, w4 = x4, w5 = x5, w6 = x6, w7 = x7;
  unsigned y = (seed | 5) & 15;
  for (int i = 0; i < ITER; i++) {
    REP16(asm volatile("%s" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(y) : "vcc", "s20", "s21", "scc");)
  }
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (unsigned)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7) == 0x12345) out[0] = 1;
}
'''
TAIL = r'''
typedef void (*kfn)(unsigned *, unsigned);
static void run(kfn f, const char *name) {
  printf("%-30s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;
    unsigned *d;
    (void)hipMalloc(&d, 64);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 3u);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)ITER * 16 * 8;
    printf("  %d w/SIMD: %.2f ns/instr/SIMD", wps, ms * 1e6 / n / wps);
    (void)hipFree(d);
  }
  printf("\n");
}
int main() {
@CALLS@  return 0;
}
'''
src = HEAD
calls = ""
for i, (name, tpl) in enumerate(OPS):
    body = "\\n ".join(tpl.format(d="%%%d" % k, D="%%%d" % (8 + k), y="%16") for k in range(8)) + "\\n"
    src += KERNEL % (i, body)
    calls += '  run(k%d, "%s");\n' % (i, name)
src += TAIL.replace('@CALLS@', calls)
open(__file__.replace("gen_valu_ops.py", OUT), "w").write(src)
