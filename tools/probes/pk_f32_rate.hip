// Probe: issue cost of the instruction classes the back end (k_idct_color) is made of, on MI355X (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/pk_f32_rate tools/probes/pk_f32_rate.hip && /tmp/pk_f32_rate
// Every kernel runs ITERS x 32 copies of one instruction on 8 independent registers per wave; 256-thread workgroups
// (one wave per SIMD), W workgroups per CU => W waves per SIMD.  Reported: shader cycles (s_memtime) per wave-instruction
// and SIMD at W = 1, 2, 4, 8 -- the W = 8 column is the throughput cost that matters for an issue-bound kernel.
// Questions this answers: do v_pk_mul_f32 / v_pk_add_f32 issue at the rate of v_mul_f32 / v_add_f32
// (two flops per lane for the price of one) or at half of it?  What do DPP operands, v_readlane, LDS reads and the
// VGPR-index mode (s_set_gpr_idx_*) cost next to a plain fp32 multiply?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITERS 1024
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R4(s) s s s s
// eight independent destinations %0..%7; %8, %9 = VGPR inputs, %10 = SGPR input
#define BODY8(op, tail) R4(op " %0, " tail "\n\t" op " %1, " tail "\n\t" op " %2, " tail "\n\t" op " %3, " tail "\n\t" \
                           op " %4, " tail "\n\t" op " %5, " tail "\n\t" op " %6, " tail "\n\t" op " %7, " tail "\n\t")
#define BODY8SELF(op, mid, tail) R4(op " %0, " mid "%0" tail "\n\t" op " %1, " mid "%1" tail "\n\t" op " %2, " mid "%2" tail "\n\t" op " %3, " mid "%3" tail "\n\t" \
                                    op " %4, " mid "%4" tail "\n\t" op " %5, " mid "%5" tail "\n\t" op " %6, " mid "%6" tail "\n\t" op " %7, " mid "%7" tail "\n\t")

struct Stamp { unsigned long long t0, t1, r0, r1; };

#define RATE_KERNEL32(name, body)                                                                                          \
__global__ void __launch_bounds__(256) name(Stamp* st, float* sink, float x, float y, unsigned s)                         \
{                                                                                                                          \
    const bool half = (s >> 31) != 0;                                                                                      \
    __shared__ float lds[4096 + 64];                                                                                       \
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) lds[i] = (float)i * x;                                              \
    __syncthreads();                                                                                                       \
    float a0 = x + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = y, c = x * 0.5f; unsigned su = __builtin_amdgcn_readfirstlane(s); f32x2 x0 = { x, y }, x1 = x0, x2 = x0, x3 = x0;   \
    unsigned la = (threadIdx.x & 63) * 4 + (unsigned)(size_t)lds;                                                          \
    unsigned la8 = (threadIdx.x & 63) * 8 + (unsigned)(size_t)lds;                                                         \
    asm volatile("" : "+v"(b), "+v"(c), "+v"(la), "+v"(la8));                                                                          \
    const unsigned long long r0 = wall_clock64(), t0 = clock64();                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(r0), "s"(t0));                                                              \
    if (half) asm volatile("s_mov_b64 exec, 0xffffffff");                                                                   \
    for (int i = 0; i < ITERS; i++)                                                                                        \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                          : "v"(b), "v"(c), "s"(su), "v"(la), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [la8] "v"(la8) : "memory", "m0", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"); \
    if (half) asm volatile("s_mov_b64 exec, -1");                                                                           \
    const unsigned long long t1 = clock64(), r1 = wall_clock64();                                                          \
    if ((threadIdx.x & 63) == 0) { Stamp q; q.t0 = t0; q.t1 = t1; q.r0 = r0; q.r1 = r1; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = q; } \
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                          \
}
#define RATE_KERNEL64(name, body)                                                                                          \
__global__ void __launch_bounds__(256) name(Stamp* st, float* sink, float x, float y, unsigned s)                         \
{                                                                                                                          \
    f32x2 a0 = { x + threadIdx.x, x }, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    f32x2 b = { y, y }, c = { x * 0.5f, x }; unsigned su = __builtin_amdgcn_readfirstlane(s);                              \
    asm volatile("" : "+v"(b), "+v"(c));                                                                                    \
    const unsigned long long r0 = wall_clock64(), t0 = clock64();                                                          \
    for (int i = 0; i < ITERS; i++)                                                                                        \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                          : "v"(b), "v"(c), "s"(su) : "memory");                                                             \
    const unsigned long long t1 = clock64(), r1 = wall_clock64();                                                          \
    if ((threadIdx.x & 63) == 0) { Stamp q; q.t0 = t0; q.t1 = t1; q.r0 = r0; q.r1 = r1; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = q; } \
    const f32x2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; sink[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;               \
}

#define DPPB " row_newbcast:3 row_mask:0xf bank_mask:0xf"
RATE_KERNEL32(k_add_f32,      BODY8SELF("v_add_f32", "%8, ", ""))
RATE_KERNEL32(k_mul_f32,      BODY8SELF("v_mul_f32", "%8, ", ""))
RATE_KERNEL32(k_fma_f32,      BODY8SELF("v_fma_f32", "%8, %9, ", ""))
RATE_KERNEL32(k_fma_one,      BODY8SELF("v_fma_f32", "%8, 1.0, ", ""))
RATE_KERNEL32(k_mul_sgpr,     BODY8SELF("v_mul_f32", "%10, ", ""))
RATE_KERNEL32(k_mul_dpp,      BODY8SELF("v_mul_f32_dpp", "%8, ", DPPB))
RATE_KERNEL32(k_addu_dpp,     BODY8SELF("v_add_u32_dpp", "%8, ", DPPB))
RATE_KERNEL32(k_mov_dpp,      BODY8("v_mov_b32_dpp", "%8" DPPB))
RATE_KERNEL32(k_add_u32,      BODY8SELF("v_add_u32", "%8, ", ""))
RATE_KERNEL32(k_lshl_or,      BODY8SELF("v_lshl_or_b32", "%8, 8, ", ""))
RATE_KERNEL32(k_alignbit,     BODY8SELF("v_alignbit_b32", "%8, %9, ", ""))
RATE_KERNEL32(k_cvt_f32_i32,  BODY8("v_cvt_f32_i32", "%8"))
RATE_KERNEL32(k_cvt_i32_f32,  BODY8("v_cvt_i32_f32", "%8"))
RATE_KERNEL32(k_med3_f32,     BODY8SELF("v_med3_f32", "%8, %9, ", ""))
RATE_KERNEL32(k_pk_add_i16,   BODY8SELF("v_pk_add_i16", "%8, ", ""))
RATE_KERNEL32(k_pk_max_i16,   BODY8SELF("v_pk_max_i16", "%8, ", ""))
RATE_KERNEL32(k_and_b32,      BODY8SELF("v_and_b32", "%8, ", ""))
RATE_KERNEL32(k_bfe_u32,      BODY8("v_bfe_u32", "%8, 3, 5"))
RATE_KERNEL32(k_cndmask,      BODY8SELF("v_cndmask_b32", "%8, ", ", vcc"))
RATE_KERNEL32(k_perm,         BODY8SELF("v_perm_b32", "%8, %9, ", ""))
RATE_KERNEL32(k_cvt_pk_u8,    BODY8SELF("v_cvt_pk_u8_f32", "%8, 1, ", ""))
RATE_KERNEL32(k_cndmask_sgpr, BODY8SELF("v_cndmask_b32_e64", "%8, ", ", s[40:41]"))
RATE_KERNEL32(k_cmp_vcc,      R4("v_cmp_lt_u32 vcc, %8, %0\n\tv_cmp_lt_u32 vcc, %8, %1\n\tv_cmp_lt_u32 vcc, %8, %2\n\tv_cmp_lt_u32 vcc, %8, %3\n\tv_cmp_lt_u32 vcc, %8, %4\n\tv_cmp_lt_u32 vcc, %8, %5\n\tv_cmp_lt_u32 vcc, %8, %6\n\tv_cmp_lt_u32 vcc, %8, %7\n\t"))
RATE_KERNEL32(k_cmp_sgpr,     R4("v_cmp_lt_u32_e64 s[40:41], %8, %0\n\tv_cmp_lt_u32_e64 s[42:43], %8, %1\n\tv_cmp_lt_u32_e64 s[44:45], %8, %2\n\tv_cmp_lt_u32_e64 s[46:47], %8, %3\n\tv_cmp_lt_u32_e64 s[40:41], %8, %4\n\tv_cmp_lt_u32_e64 s[42:43], %8, %5\n\tv_cmp_lt_u32_e64 s[44:45], %8, %6\n\tv_cmp_lt_u32_e64 s[46:47], %8, %7\n\t"))
RATE_KERNEL32(k_cmp_cnd,      R4("v_cmp_lt_u32 vcc, %8, %0\n\tv_cndmask_b32 %0, %8, %0, vcc\n\tv_cmp_lt_u32 vcc, %8, %1\n\tv_cndmask_b32 %1, %8, %1, vcc\n\tv_cmp_lt_u32 vcc, %8, %2\n\tv_cndmask_b32 %2, %8, %2, vcc\n\tv_cmp_lt_u32 vcc, %8, %3\n\tv_cndmask_b32 %3, %8, %3, vcc\n\t"))
RATE_KERNEL32(k_min_u32,      BODY8SELF("v_min_u32", "%8, ", ""))
RATE_KERNEL32(k_max_i32,      BODY8SELF("v_max_i32", "%8, ", ""))
RATE_KERNEL32(k_min_f32,      BODY8SELF("v_min_f32", "%8, ", ""))
RATE_KERNEL32(k_max_f32,      BODY8SELF("v_max_f32", "%8, ", ""))
RATE_KERNEL32(k_floor_f32,    BODY8("v_floor_f32", "%8"))
RATE_KERNEL32(k_trunc_f32,    BODY8("v_trunc_f32", "%8"))
RATE_KERNEL32(k_cvt_u32_f32,  BODY8("v_cvt_u32_f32", "%8"))
RATE_KERNEL32(k_mov_b32,      BODY8("v_mov_b32", "%8"))
RATE_KERNEL32(k_lshlrev,      BODY8SELF("v_lshlrev_b32", "3, ", ""))
RATE_KERNEL32(k_lshrrev,      BODY8SELF("v_lshrrev_b32", "3, ", ""))
RATE_KERNEL32(k_lshlrev_v,    BODY8SELF("v_lshlrev_b32", "%8, ", ""))
RATE_KERNEL32(k_ashrrev,      BODY8SELF("v_ashrrev_i32", "3, ", ""))
RATE_KERNEL32(k_or_b32,       BODY8SELF("v_or_b32", "%8, ", ""))
RATE_KERNEL32(k_xor_b32,      BODY8SELF("v_xor_b32", "%8, ", ""))
RATE_KERNEL32(k_sub_u32,      BODY8SELF("v_sub_u32", "%8, ", ""))
RATE_KERNEL32(k_add_lit,      BODY8SELF("v_add_u32", "0x12345, ", ""))
RATE_KERNEL32(k_add_inl,      BODY8SELF("v_add_u32", "17, ", ""))
RATE_KERNEL32(k_mul_inl,      BODY8SELF("v_mul_f32", "0.5, ", ""))
RATE_KERNEL32(k_mul_lit,      BODY8SELF("v_mul_f32", "0x3fb374bc, ", ""))
RATE_KERNEL32(k_mul_u24,      BODY8SELF("v_mul_u32_u24", "%8, ", ""))
RATE_KERNEL32(k_mad_u24,      BODY8SELF("v_mad_u32_u24", "%8, %9, ", ""))
RATE_KERNEL32(k_mul_lo,       BODY8SELF("v_mul_lo_u32", "%8, ", ""))
RATE_KERNEL32(k_bfi,          BODY8SELF("v_bfi_b32", "%8, %9, ", ""))
RATE_KERNEL32(k_and_or,       BODY8SELF("v_and_or_b32", "%8, %9, ", ""))
RATE_KERNEL32(k_add3,         BODY8SELF("v_add3_u32", "%8, %9, ", ""))
RATE_KERNEL32(k_lshl_add,     BODY8SELF("v_lshl_add_u32", "%8, 2, ", ""))
RATE_KERNEL32(k_bcnt,         BODY8SELF("v_bcnt_u32_b32", "%8, ", ""))
RATE_KERNEL32(k_mbcnt,        BODY8SELF("v_mbcnt_lo_u32_b32", "%8, ", ""))
RATE_KERNEL32(k_ffbh,         BODY8("v_ffbh_u32", "%8"))
RATE_KERNEL32(k_sdwa,         BODY8SELF("v_add_u32_sdwa", "%8, ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"))
RATE_KERNEL32(k_mov_dpp_shr,  BODY8("v_mov_b32_dpp", "%8 row_shr:1 row_mask:0xf bank_mask:0xf"))
RATE_KERNEL32(k_ds_read_b64,  R4("ds_read_b64 %[x0], %[la8]\n\tds_read_b64 %[x1], %[la8] offset:512\n\tds_read_b64 %[x2], %[la8] offset:1024\n\tds_read_b64 %[x3], %[la8] offset:1536\n\ts_waitcnt lgkmcnt(0)\n\t"))
RATE_KERNEL32(k_ds_write_b32, R4("ds_write_b32 %11, %0\n\tds_write_b32 %11, %1 offset:256\n\tds_write_b32 %11, %2 offset:512\n\tds_write_b32 %11, %3 offset:768\n\tds_write_b32 %11, %4 offset:1024\n\tds_write_b32 %11, %5 offset:1280\n\tds_write_b32 %11, %6 offset:1536\n\tds_write_b32 %11, %7 offset:1792\n\ts_waitcnt lgkmcnt(0)\n\t"))
RATE_KERNEL32(k_ds_write_b16, R4("ds_write_b16 %11, %0\n\tds_write_b16 %11, %1 offset:256\n\tds_write_b16 %11, %2 offset:512\n\tds_write_b16 %11, %3 offset:768\n\tds_write_b16 %11, %4 offset:1024\n\tds_write_b16 %11, %5 offset:1280\n\tds_write_b16 %11, %6 offset:1536\n\tds_write_b16 %11, %7 offset:1792\n\ts_waitcnt lgkmcnt(0)\n\t"))
RATE_KERNEL32(k_salu_bfe,     R4("s_bfe_u32 s40, %10, 0x80008\n\ts_lshl_b32 s41, s40, 8\n\ts_bfe_u32 s42, %10, 0x80010\n\ts_lshl_b32 s43, s42, 8\n\ts_bfe_u32 s44, %10, 0x80008\n\ts_lshl_b32 s45, s44, 8\n\ts_bfe_u32 s46, %10, 0x80010\n\ts_lshl_b32 s47, s46, 8\n\t"))
// two VALU + two SALU per group: do the scalar instructions hide behind the vector ones?
RATE_KERNEL32(k_valu_salu,    R4("v_mul_f32 %0, %8, %0\n\ts_bfe_u32 s40, %10, 0x80008\n\tv_add_f32 %1, %8, %1\n\ts_lshl_b32 s41, s40, 8\n\tv_mul_f32 %2, %8, %2\n\ts_bfe_u32 s42, %10, 0x80010\n\tv_add_f32 %3, %8, %3\n\ts_lshl_b32 s43, s42, 8\n\t"))
// candidate term: M0 row offset (SALU) + ds_read_addtid_b32 + multiply with the coefficient as DPP operand + add; 4 terms, reads issued ahead
RATE_KERNEL32(k_term_addtid,  R4("s_lshl_b32 m0, %10, 8\n\ts_nop 0\n\tds_read_addtid_b32 %4\n\ts_and_b32 m0, %10, 0xff00\n\ts_nop 0\n\tds_read_addtid_b32 %5\n\ts_bfe_u32 s40, %10, 0x80010\n\ts_lshl_b32 m0, s40, 8\n\ts_nop 0\n\tds_read_addtid_b32 %6\n\ts_lshr_b32 s40, %10, 24\n\ts_lshl_b32 m0, s40, 8\n\ts_nop 0\n\tds_read_addtid_b32 %7\n\t"
                                 "s_waitcnt lgkmcnt(3)\n\tv_mul_f32_dpp %4, %8, %4" DPPB "\n\tv_add_f32 %0, %0, %4\n\ts_waitcnt lgkmcnt(2)\n\tv_mul_f32_dpp %5, %8, %5" DPPB "\n\tv_add_f32 %0, %0, %5\n\t"
                                 "s_waitcnt lgkmcnt(1)\n\tv_mul_f32_dpp %6, %8, %6" DPPB "\n\tv_add_f32 %0, %0, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32_dpp %7, %8, %7" DPPB "\n\tv_add_f32 %0, %0, %7\n\t"))
// production term, four reads in flight (as k_idct_color issues them)
RATE_KERNEL32(k_term_dpp4,    R4("v_add_u32_dpp %4, %9, %11" DPPB "\n\tv_add_u32_dpp %5, %9, %11" DPPB "\n\tv_add_u32_dpp %6, %9, %11" DPPB "\n\tv_add_u32_dpp %7, %9, %11" DPPB "\n\t"
                                 "ds_read_b32 %4, %4\n\tds_read_b32 %5, %5\n\tds_read_b32 %6, %6\n\tds_read_b32 %7, %7\n\t"
                                 "s_waitcnt lgkmcnt(3)\n\tv_mul_f32_dpp %4, %8, %4" DPPB "\n\tv_add_f32 %0, %0, %4\n\ts_waitcnt lgkmcnt(2)\n\tv_mul_f32_dpp %5, %8, %5" DPPB "\n\tv_add_f32 %0, %0, %5\n\t"
                                 "s_waitcnt lgkmcnt(1)\n\tv_mul_f32_dpp %6, %8, %6" DPPB "\n\tv_add_f32 %0, %0, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32_dpp %7, %8, %7" DPPB "\n\tv_add_f32 %0, %0, %7\n\t"))
RATE_KERNEL32(k_readlane,     R4("v_readlane_b32 s40, %0, 5\n\tv_readlane_b32 s41, %1, 6\n\tv_readlane_b32 s42, %2, 7\n\tv_readlane_b32 s43, %3, 8\n\t"
                                 "v_readlane_b32 s44, %4, 9\n\tv_readlane_b32 s45, %5, 10\n\tv_readlane_b32 s46, %6, 11\n\tv_readlane_b32 s47, %7, 12\n\t"))
RATE_KERNEL32(k_salu,         R4("s_lshr_b32 s40, %10, 8\n\ts_lshr_b32 s41, %10, 8\n\ts_lshr_b32 s42, %10, 8\n\ts_lshr_b32 s43, %10, 8\n\t"
                                 "s_lshr_b32 s44, %10, 8\n\ts_lshr_b32 s45, %10, 8\n\ts_lshr_b32 s46, %10, 8\n\ts_lshr_b32 s47, %10, 8\n\t"))
// one SALU between two VALU (do they co-issue from the same wave / from different waves?)
RATE_KERNEL32(k_mul_salu,     R4("v_mul_f32 %0, %8, %0\n\ts_lshr_b32 s40, %10, 8\n\tv_mul_f32 %1, %8, %1\n\ts_lshr_b32 s41, %10, 8\n\tv_mul_f32 %2, %8, %2\n\ts_lshr_b32 s42, %10, 8\n\t"
                                 "v_mul_f32 %3, %8, %3\n\ts_lshr_b32 s43, %10, 8\n\tv_mul_f32 %4, %8, %4\n\ts_lshr_b32 s44, %10, 8\n\tv_mul_f32 %5, %8, %5\n\ts_lshr_b32 s45, %10, 8\n\t"
                                 "v_mul_f32 %6, %8, %6\n\ts_lshr_b32 s46, %10, 8\n\tv_mul_f32 %7, %8, %7\n\ts_lshr_b32 s47, %10, 8\n\t"))
// the reference's term: dependent multiply -> add chains (4 chains, two instructions each)
RATE_KERNEL32(k_muladd_chain, R4("v_mul_f32 %4, %8, %9\n\tv_add_f32 %0, %0, %4\n\tv_mul_f32 %5, %8, %9\n\tv_add_f32 %1, %1, %5\n\t"
                                 "v_mul_f32 %6, %8, %9\n\tv_add_f32 %2, %2, %6\n\tv_mul_f32 %7, %8, %9\n\tv_add_f32 %3, %3, %7\n\t"))
// LDS reads: 8 per wait
RATE_KERNEL32(k_ds_read_b32,  R4("ds_read_b32 %0, %11\n\tds_read_b32 %1, %11 offset:256\n\tds_read_b32 %2, %11 offset:512\n\tds_read_b32 %3, %11 offset:768\n\t"
                                 "ds_read_b32 %4, %11 offset:1024\n\tds_read_b32 %5, %11 offset:1280\n\tds_read_b32 %6, %11 offset:1536\n\tds_read_b32 %7, %11 offset:1792\n\ts_waitcnt lgkmcnt(0)\n\t"))
RATE_KERNEL32(k_ds_addtid,    R4("s_mov_b32 m0, %10\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\tds_read_addtid_b32 %1 offset:256\n\tds_read_addtid_b32 %2 offset:512\n\tds_read_addtid_b32 %3 offset:768\n\t"
                                 "ds_read_addtid_b32 %4 offset:1024\n\tds_read_addtid_b32 %5 offset:1280\n\tds_read_addtid_b32 %6 offset:1536\n\tds_read_addtid_b32 %7 offset:1792\n\ts_waitcnt lgkmcnt(0)\n\t"))
// the production term: address add (DPP) + LDS read + multiply (DPP) + add, 4 terms
RATE_KERNEL32(k_term_dpp,     R4("v_add_u32_dpp %4, %9, %11" DPPB "\n\tds_read_b32 %5, %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32_dpp %5, %8, %5" DPPB "\n\tv_add_f32 %0, %0, %5\n\t"))

RATE_KERNEL64(k_pk_add_f32,   BODY8SELF("v_pk_add_f32", "%8, ", ""))
RATE_KERNEL64(k_pk_mul_f32,   BODY8SELF("v_pk_mul_f32", "%8, ", ""))
RATE_KERNEL64(k_pk_fma_f32,   BODY8SELF("v_pk_fma_f32", "%8, %9, ", ""))
// 64-bit DPP move (gfx90a+: row_newbcast only): does it broadcast two dwords for the price of one v_mov_b32_dpp?  And ds_read_b128.
RATE_KERNEL64(k_mov_b64_dpp,  BODY8("v_mov_b64_dpp", "%8" DPPB))
RATE_KERNEL64(k_pk_mul_opsel, BODY8SELF("v_pk_mul_f32", "%8, ", " op_sel_hi:[1,0]"))

// ---- VGPR-index mode: 64 registers v[64:127] hold a table, the row comes from an SGPR through M0 -------------------
// (a) function: acc = sum over a pseudo-random row sequence of c * T[row], against the same sum formed from LDS
// (b) rate of { s_set_gpr_idx_idx ; v_mul_f32 t, s, v[64 + M0] ; v_fma_f32 acc, t, 1.0, acc }
#define CLOB64 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
               "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56))) k_gpridx(Stamp* st, float* sink, float x, float y, unsigned s, int dpp_form, unsigned* bad)
{
    __shared__ float lds[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)((i * 2654435761u >> 9) & 0xFFFF) * (1.0f / 4096.0f) - 7.0f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    for (unsigned r = 0; r < 64; r++) {                       // T[r] <- lds[r][lane], written through the destination index
        const float v = lds[r * 64 + lane]; const unsigned rs = __builtin_amdgcn_readfirstlane(r);
        asm volatile("s_set_gpr_idx_on %1, 8\n\tv_mov_b32 v64, %0\n\ts_set_gpr_idx_off" :: "v"(v), "s"(rs) : CLOB64, "m0");
    }
    // (a) function
    float acc = 0.f, ref = 0.f; unsigned z = s | 1u; float cc = x;
    for (int i = 0; i < 512; i++) {
        z = z * 1664525u + 1013904223u; const unsigned row = __builtin_amdgcn_readfirstlane((z >> 10) & 63u);
        const float cf = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)(int)((z >> 16) % 2001u) - 1000.0f)));
        float t;
        if (!dpp_form) asm volatile("s_nop 4\n\ts_set_gpr_idx_on %2, 2\n\tv_mul_f32 %1, %3, v64\n\ts_set_gpr_idx_off\n\tv_fma_f32 %0, %1, 1.0, %0" : "+v"(acc), "=&v"(t) : "s"(row), "s"(cf) : CLOB64, "m0");
        else { float cv = cf; asm volatile("" : "+v"(cv));
               asm volatile("s_nop 4\n\ts_set_gpr_idx_on %2, 2\n\tv_mul_f32_dpp %1, %3, v64 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\ts_set_gpr_idx_off\n\tv_fma_f32 %0, %1, 1.0, %0" : "+v"(acc), "=&v"(t) : "s"(row), "v"(cv) : CLOB64, "m0"); }
        ref = __fadd_rn(ref, __fmul_rn(cf, lds[row * 64 + lane]));
    }
    if (__float_as_uint(acc) != __float_as_uint(ref)) atomicAdd(bad, 1u);
    // (b) rate
    float a0 = acc, a1 = x; unsigned su = __builtin_amdgcn_readfirstlane(s & 63u), sv = __builtin_amdgcn_readfirstlane((s >> 3) & 63u);
    float cs = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(y))); float cv = y; asm volatile("" : "+v"(cv));
    const unsigned long long r0 = wall_clock64(), t0 = clock64();
    for (int i = 0; i < ITERS; i++) {
        float t0_, t1_, t2_, t3_;
#define TERM_S(T, A, IDX) "s_set_gpr_idx_idx " IDX "\n\tv_mul_f32 " T ", %8, v64\n\tv_fma_f32 " A ", " T ", 1.0, " A "\n\t"
#define TERM_D(T, A, IDX) "s_set_gpr_idx_idx " IDX "\n\tv_mul_f32_dpp " T ", %9, v64 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f32 " A ", " T ", 1.0, " A "\n\t"
        if (!dpp_form)
            asm volatile("s_set_gpr_idx_on %6, 2\n\t" R4(TERM_S("%2", "%0", "%6") TERM_S("%3", "%1", "%7") TERM_S("%4", "%0", "%7") TERM_S("%5", "%1", "%6")) "s_set_gpr_idx_off"
                         : "+v"(a0), "+v"(a1), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "s"(su), "s"(sv), "s"(cs), "v"(cv) : CLOB64, "m0");
        else
            asm volatile("s_set_gpr_idx_on %6, 2\n\t" R4(TERM_D("%2", "%0", "%6") TERM_D("%3", "%1", "%7") TERM_D("%4", "%0", "%7") TERM_D("%5", "%1", "%6")) "s_set_gpr_idx_off"
                         : "+v"(a0), "+v"(a1), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "s"(su), "s"(sv), "s"(cs), "v"(cv) : CLOB64, "m0");
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (lane == 0) { Stamp q; q.t0 = t0; q.t1 = t1; q.r0 = r0; q.r1 = r1; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = q; }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1;
}

typedef void (*kern_t)(Stamp*, float*, float, float, unsigned);
struct Entry { const char* name; kern_t k; int per_iter; const char* note; };

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    Stamp* d_st; float* d_sink; unsigned* d_bad;
    const int max_wg = cus * 8;
    CHECK(hipMalloc(&d_st, (size_t)max_wg * 4 * sizeof(Stamp))); CHECK(hipMalloc(&d_sink, (size_t)max_wg * 256 * 4)); CHECK(hipMalloc(&d_bad, 4));
    std::vector<Stamp> h((size_t)max_wg * 4);
    const Entry tab[] = {
        { "v_add_f32", k_add_f32, 32, "" }, { "v_mul_f32", k_mul_f32, 32, "" }, { "v_fma_f32", k_fma_f32, 32, "" }, { "v_fma_f32 x,1.0,acc", k_fma_one, 32, "" },
        { "v_mul_f32 sgpr src", k_mul_sgpr, 32, "" }, { "v_mul_f32 inline const", k_mul_inl, 32, "" }, { "v_mul_f32 literal", k_mul_lit, 32, "" },
        { "v_min_f32", k_min_f32, 32, "" }, { "v_max_f32", k_max_f32, 32, "" }, { "v_floor_f32", k_floor_f32, 32, "" }, { "v_trunc_f32", k_trunc_f32, 32, "" },
        { "v_mov_b64_dpp row_newbcast", k_mov_b64_dpp, 32, "two dwords per lane" }, { "v_pk_mul_f32 op_sel_hi:[1,0]", k_pk_mul_opsel, 32, "both products with src1.lo" }, { "v_pk_add_f32", k_pk_add_f32, 32, "2 flops/lane" }, { "v_pk_mul_f32", k_pk_mul_f32, 32, "2 flops/lane" }, { "v_pk_fma_f32", k_pk_fma_f32, 32, "4 flops/lane" },
        { "v_mul_f32_dpp row_newbcast", k_mul_dpp, 32, "" }, { "v_add_u32_dpp row_newbcast", k_addu_dpp, 32, "" }, { "v_mov_b32_dpp row_newbcast", k_mov_dpp, 32, "" },
        { "v_mov_b32_dpp row_shr:1", k_mov_dpp_shr, 32, "" }, { "v_add_u32_sdwa", k_sdwa, 32, "" },
        { "v_mov_b32", k_mov_b32, 32, "" }, { "v_add_u32", k_add_u32, 32, "" }, { "v_add_u32 inline const", k_add_inl, 32, "" }, { "v_add_u32 literal", k_add_lit, 32, "" }, { "v_sub_u32", k_sub_u32, 32, "" },
        { "v_and_b32", k_and_b32, 32, "" }, { "v_or_b32", k_or_b32, 32, "" }, { "v_xor_b32", k_xor_b32, 32, "" },
        { "v_lshlrev_b32 const", k_lshlrev, 32, "" }, { "v_lshlrev_b32 vgpr", k_lshlrev_v, 32, "" }, { "v_lshrrev_b32", k_lshrrev, 32, "" }, { "v_ashrrev_i32", k_ashrrev, 32, "" },
        { "v_min_u32", k_min_u32, 32, "" }, { "v_max_i32", k_max_i32, 32, "" },
        { "v_mul_u32_u24", k_mul_u24, 32, "" }, { "v_mad_u32_u24", k_mad_u24, 32, "" }, { "v_mul_lo_u32", k_mul_lo, 32, "" },
        { "v_lshl_or_b32", k_lshl_or, 32, "" }, { "v_lshl_add_u32", k_lshl_add, 32, "" }, { "v_add3_u32", k_add3, 32, "" }, { "v_and_or_b32", k_and_or, 32, "" }, { "v_bfi_b32", k_bfi, 32, "" },
        { "v_alignbit_b32", k_alignbit, 32, "" }, { "v_bfe_u32", k_bfe_u32, 32, "" }, { "v_perm_b32", k_perm, 32, "" }, { "v_bcnt_u32_b32", k_bcnt, 32, "" }, { "v_mbcnt_lo_u32_b32", k_mbcnt, 32, "" }, { "v_ffbh_u32", k_ffbh, 32, "" },
        { "v_cndmask_b32 (vcc)", k_cndmask, 32, "" }, { "v_cndmask_b32_e64 (sgpr pair)", k_cndmask_sgpr, 32, "" }, { "v_cmp_lt_u32 -> vcc", k_cmp_vcc, 32, "" }, { "v_cmp_lt_u32_e64 -> sgpr pair", k_cmp_sgpr, 32, "" },
        { "v_cmp + v_cndmask pairs", k_cmp_cnd, 16, "per pair" },
        { "v_cvt_f32_i32", k_cvt_f32_i32, 32, "" }, { "v_cvt_i32_f32", k_cvt_i32_f32, 32, "" }, { "v_cvt_u32_f32", k_cvt_u32_f32, 32, "" }, { "v_med3_f32", k_med3_f32, 32, "" }, { "v_cvt_pk_u8_f32", k_cvt_pk_u8, 32, "" },
        { "v_pk_add_i16", k_pk_add_i16, 32, "" }, { "v_pk_max_i16", k_pk_max_i16, 32, "" },
        { "v_readlane_b32", k_readlane, 32, "" }, { "s_lshr_b32 (SALU)", k_salu, 32, "" }, { "s_bfe_u32 + s_lshl_b32 (SALU)", k_salu_bfe, 32, "" },
        { "v_mul_f32 + s_lshr_b32 pairs", k_mul_salu, 32, "per pair" }, { "2 VALU + 2 SALU groups", k_valu_salu, 8, "per group of four" },
        { "mul->add dependent pairs", k_muladd_chain, 16, "per mul+add pair, 4 chains" },
        { "ds_read_b32", k_ds_read_b32, 32, "8 per wait" }, { "ds_read_b64", k_ds_read_b64, 16, "4 per wait" }, { "ds_read_addtid_b32", k_ds_addtid, 32, "8 per wait + s_mov m0" },
        { "ds_write_b32", k_ds_write_b32, 32, "8 per wait" }, { "ds_write_b16", k_ds_write_b16, 32, "8 per wait" },
        { "term: add_dpp+ds_read+mul_dpp+add", k_term_dpp, 4, "per term, waits for its read" },
        { "term x4 (production, 4 reads ahead)", k_term_dpp4, 16, "per term" },
        { "term x4 (M0 + ds_read_addtid)", k_term_addtid, 16, "per term" },
    };
    printf("%-36s %8s %8s %8s %8s   (shader cycles per wave-instruction and SIMD from the kernel's wall time; W = waves per SIMD; [per-wave cycles per instruction at W = 8])\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    hipEvent_t ea, eb; CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
    auto run = [&](const char* name, auto launch, int per_iter, const char* note) {
        printf("%-36s", name);
        double ghz = 0, own = 0;
        for (int W = 1; W <= 8; W *= 2) {
            const int wgs = cus * W;
            launch(wgs); CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(ea)); launch(wgs); CHECK(hipEventRecord(eb)); CHECK(hipEventSynchronize(eb));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, ea, eb));
            CHECK(hipMemcpy(h.data(), d_st, (size_t)wgs * 4 * sizeof(Stamp), hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; unsigned long long tmin = ~0ull, tmax = 0;
            for (int i = 0; i < wgs * 4; i++) { cyc += (double)(h[i].t1 - h[i].t0); rt += (double)(h[i].r1 - h[i].r0); tmin = std::min(tmin, h[i].r0); tmax = std::max(tmax, h[i].r1); }
            cyc /= wgs * 4; rt /= wgs * 4;
            ghz = cyc / (rt * 10.0);          // s_memrealtime ticks at 100 MHz
            const double span_s = (double)(tmax - tmin) * 1e-8;          // first wave start -> last wave end (s_memrealtime)
            printf(" %8.2f", span_s * ghz * 1e9 * (cus * 4.0) / ((double)wgs * 4 * ITERS * per_iter));
            own = cyc / ((double)ITERS * per_iter);
            (void)ms;
        }
        printf("   [%.2f] %s  [%.2f GHz]\n", own, note, ghz);
    };
    for (const Entry& e : tab) run(e.name, [&](int wgs) { hipLaunchKernelGGL(e.k, dim3(wgs), dim3(256), 0, 0, d_st, d_sink, 1.0009765625f, 0.99951171875f, 0x03020100u); }, e.per_iter, e.note);
    printf("---- the same with EXEC = lanes 0..31 only (does a half-empty wave issue in half the time?)\n");
    for (const Entry& e : tab) {
        const char* pick[] = { "v_add_f32", "v_mul_f32_dpp row_newbcast", "v_add_u32_dpp row_newbcast", "v_cvt_f32_i32", "v_med3_f32", "ds_read_b32", "ds_read_b64", "term x4 (production, 4 reads ahead)" };
        bool on = false; for (const char* p : pick) on = on || !strcmp(p, e.name);
        if (on) run(e.name, [&](int wgs) { hipLaunchKernelGGL(e.k, dim3(wgs), dim3(256), 0, 0, d_st, d_sink, 1.0009765625f, 0.99951171875f, 0x83020100u); }, e.per_iter, "EXEC = low half");
    }
    for (int form = 0; form < 2; form++) {
        CHECK(hipMemset(d_bad, 0, 4));
        run(form ? "gpr-idx term (coef via DPP)" : "gpr-idx term (coef from SGPR)",
            [&](int wgs) { hipLaunchKernelGGL(k_gpridx, dim3(wgs), dim3(256), 0, 0, d_st, d_sink, 1.0009765625f, 0.99951171875f, 0x1234567u, form, d_bad); }, 16, "per term: s_set_gpr_idx_idx + v_mul(v[64+M0]) + v_fma");
        unsigned bad = 0; CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("    function check of the indexed multiply (%s): %u lanes differ from the LDS-table sum\n", form ? "DPP form" : "SGPR form", bad);
    }
    return 0;
}
