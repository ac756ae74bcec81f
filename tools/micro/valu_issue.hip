// Microbenchmark: what does ONE wave64 vector instruction cost a gfx950 SIMD, per instruction class and as a function of the
#pragma clang diagnostic ignored "-Wunused-value"
// waves resident per SIMD?  bench.py's `issue_frac` prices the blend kernels' SQ_INSTS_VALU against this constant; the guide
// (MI355X_MICROARCH.md "Per-instruction cycle constants") says 2 cycles for v_fma_f32, round 4 assumed 4.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue > profiles/r05_valu_issue.txt
// Every kernel runs `iters` times a block of N instructions written in inline assembly (eight independent chains, so no
// dependent-issue stall is measured), one workgroup of 256 threads (a wave per SIMD) x `w` workgroups per CU, and reports
//   cyc/inst = shader cycles (s_memtime) one wave spent in the loop / instructions it issued            (what a wave sees)
//   cyc/inst/SIMD = that / waves per SIMD                                                             (what the SIMD sustains)
//   G wave-inst/s = instructions of all waves / wall time of the launch (hip events)                   (what the chip sustains)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <map>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Op {
    FMA, FMA_DEP, PK_FMA, MUL_ADD_MIX, CMP_SGPR, CMP_VCC, CNDMASK, DPP_ADD, RCP, EXP, MOV, FMA_HALF_EXEC, FMA_QUARTER_EXEC,
    FMA_SALU, LDS_B128_BCAST, LDS_B128_2ADDR, LDS_B128_4ADDR, LDS_B128_LANE, LDS_B32_BCAST, MFMA_32, MFMA_16, MFMA_FMA, READLANE,
    PERMLANE_SWAP, N_OPS
};
static const char* kNames[N_OPS] = {
    "v_fma_f32 (8 chains)", "v_fma_f32 (1 dependent chain)", "v_pk_fma_f32", "v_mul_f32 + v_add_f32 alternating",
    "v_cmp_lt_f32 -> sgpr pair", "v_cmp_lt_f32 -> vcc", "v_cndmask_b32 (sgpr mask)", "v_add_f32 dpp row_shr:1", "v_rcp_f32",
    "v_exp_f32", "v_mov_b32", "v_fma_f32, exec = low 32 lanes", "v_fma_f32, exec = low 16 lanes",
    "v_fma_f32 + s_and_b64 1:1", "ds_read_b128 one address", "ds_read_b128 two addresses (lane>>5)",
    "ds_read_b128 four addresses (lane>>4)", "ds_read_b128 address per lane", "ds_read_b32 one address",
    "v_mfma_f32_32x32x2_f32 (2 accumulators)", "v_mfma_f32_16x16x4_f32 (4 accumulators)", "mfma 32x32x2 + 16 v_fma_f32",
    "v_readlane_b32", "v_permlane32_swap"};
// instructions per asm block (what cyc/inst divides by)
static const int kPerBlock[N_OPS] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 8, 8, 8, 8, 8, 2, 4, 17, 16, 8};

template <int OP>
__global__ __launch_bounds__(256) void k_issue(int iters, unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    const v2f pm = {m, m}, pc = {c, c};
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    v4f r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
    v4f q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    unsigned u0 = threadIdx.x, u1 = u0 + 1;
    unsigned addr = 0;
    if (OP == LDS_B128_2ADDR) addr = (lane >> 5) * 96;
    if (OP == LDS_B128_4ADDR) addr = (lane >> 4) * 96;
    if (OP == LDS_B128_LANE) addr = lane * 16;
    addr += (threadIdx.x >> 6) * 4096;      // a region per wave
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; it++) {
        if constexpr (OP == FMA) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (OP == FMA_DEP) {
            asm volatile(
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                : "+v"(a0) : "v"(m), "v"(c));
        } else if constexpr (OP == PK_FMA) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
        } else if constexpr (OP == MUL_ADD_MIX) {
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (OP == CMP_SGPR) {
            asm volatile(
                "v_cmp_lt_f32 %0, %8, %9\n v_cmp_lt_f32 %1, %9, %10\n v_cmp_lt_f32 %2, %10, %11\n v_cmp_lt_f32 %3, %11, %8\n"
                "v_cmp_lt_f32 %4, %8, %10\n v_cmp_lt_f32 %5, %9, %11\n v_cmp_lt_f32 %6, %10, %8\n v_cmp_lt_f32 %7, %11, %9\n"
                "v_cmp_gt_f32 %0, %8, %9\n v_cmp_gt_f32 %1, %9, %10\n v_cmp_gt_f32 %2, %10, %11\n v_cmp_gt_f32 %3, %11, %8\n"
                "v_cmp_gt_f32 %4, %8, %10\n v_cmp_gt_f32 %5, %9, %11\n v_cmp_gt_f32 %6, %10, %8\n v_cmp_gt_f32 %7, %11, %9\n"
                : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if constexpr (OP == CMP_VCC) {
            asm volatile(
                "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n"
                "v_cmp_lt_f32 vcc, %0, %2\n v_cmp_lt_f32 vcc, %1, %3\n v_cmp_lt_f32 vcc, %2, %0\n v_cmp_lt_f32 vcc, %3, %1\n"
                "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n"
                "v_cmp_gt_f32 vcc, %0, %2\n v_cmp_gt_f32 vcc, %1, %3\n v_cmp_gt_f32 vcc, %2, %0\n v_cmp_gt_f32 vcc, %3, %1\n"
                : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if constexpr (OP == CNDMASK) {
            s0 = 0x5555aaaa5555aaaaull;
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                "v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(s0));
        } else if constexpr (OP == DPP_ADD) {
            asm volatile(
                "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (OP == RCP || OP == EXP || OP == MOV) {
#define ISSUE8(INS) INS " %0, %8\n " INS " %1, %9\n " INS " %2, %10\n " INS " %3, %11\n " INS " %4, %8\n " INS " %5, %9\n " INS " %6, %10\n " INS " %7, %11\n"
            if constexpr (OP == RCP)
                asm volatile(ISSUE8("v_rcp_f32") ISSUE8("v_rcp_f32")
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(m), "v"(c), "v"(p0.x), "v"(p0.y));
            else if constexpr (OP == EXP)
                asm volatile(ISSUE8("v_exp_f32") ISSUE8("v_exp_f32")
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(m), "v"(c), "v"(p0.x), "v"(p0.y));
            else
                asm volatile(ISSUE8("v_mov_b32") ISSUE8("v_mov_b32")
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(m), "v"(c), "v"(p0.x), "v"(p0.y));
        } else if constexpr (OP == FMA_HALF_EXEC || OP == FMA_QUARTER_EXEC) {
#define FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            if constexpr (OP == FMA_HALF_EXEC)
                asm volatile("s_mov_b64 exec, 0xffffffff\n" FMA8 FMA8 FMA8 FMA8 "s_mov_b64 exec, -1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            else
                asm volatile("s_mov_b64 exec, 0xffff\n" FMA8 FMA8 FMA8 FMA8 "s_mov_b64 exec, -1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (OP == FMA_SALU) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %1, %1, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %2, %2, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %3, %3, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %4, %4, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %5, %5, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %6, %6, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %7, %7, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %0, %0, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %1, %1, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %2, %2, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %3, %3, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %4, %4, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %5, %5, %8, %9\n s_or_b64 %11, %10, %11\n"
                "v_fma_f32 %6, %6, %8, %9\n s_and_b64 %10, %10, %11\n v_fma_f32 %7, %7, %8, %9\n s_or_b64 %11, %10, %11\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(s0), "s"(s1) : "scc");
        } else if constexpr (OP == LDS_B128_BCAST || OP == LDS_B128_2ADDR || OP == LDS_B128_4ADDR || OP == LDS_B128_LANE) {
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:1024\n ds_read_b128 %7, %8 offset:1040\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        } else if constexpr (OP == LDS_B32_BCAST) {
            asm volatile(
                "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:16\n ds_read_b32 %2, %8 offset:32\n ds_read_b32 %3, %8 offset:48\n"
                "ds_read_b32 %4, %8 offset:64\n ds_read_b32 %5, %8 offset:80\n ds_read_b32 %6, %8 offset:1024\n ds_read_b32 %7, %8 offset:1040\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory");
        } else if constexpr (OP == MFMA_32) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, accB, 0, 0, 0);
        } else if constexpr (OP == MFMA_16) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q3, 0, 0, 0);
        } else if constexpr (OP == MFMA_FMA) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(m, c, accA, 0, 0, 0);
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (OP == READLANE) {
            unsigned t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_;
            asm volatile(
                "v_readlane_b32 %0, %8, 1\n v_readlane_b32 %1, %8, 2\n v_readlane_b32 %2, %8, 3\n v_readlane_b32 %3, %8, 4\n"
                "v_readlane_b32 %4, %8, 5\n v_readlane_b32 %5, %8, 6\n v_readlane_b32 %6, %8, 7\n v_readlane_b32 %7, %8, 8\n"
                "v_readlane_b32 %0, %8, 9\n v_readlane_b32 %1, %8, 10\n v_readlane_b32 %2, %8, 11\n v_readlane_b32 %3, %8, 12\n"
                "v_readlane_b32 %4, %8, 13\n v_readlane_b32 %5, %8, 14\n v_readlane_b32 %6, %8, 15\n v_readlane_b32 %7, %8, 16\n"
                : "=s"(t0_), "=s"(t1_), "=s"(t2_), "=s"(t3_), "=s"(t4_), "=s"(t5_), "=s"(t6_), "=s"(t7_) : "v"(u0));
            u1 += t0_ ^ t7_;
        } else if constexpr (OP == PERMLANE_SWAP) {
            asm volatile(
                "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        const size_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[4 * wid] = t1 - t0;
        cyc[4 * wid + 1] = ((unsigned long long)(xcc_id & 0xf) << 16) | (hw_id & 0xfff0u);     // SIMD = (xcc, se, sh, cu, pipe, simd)
        cyc[4 * wid + 2] = w0;
        cyc[4 * wid + 3] = w1;
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + r0.x + r1.y + r2.z + r3.w +
              r4.x + r5.y + r6.z + r7.w + accA[0] + accB[5] + q0.x + q1.y + q2.z + q3.w + (float)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) + (float)u1;
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int OP>
static void run(int iters, unsigned long long* d_cyc, float* d_sink, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-42s", kNames[OP]);
    for (int w : {1, 2, 4, 8}) {
        const int grid = cus * w;
        k_issue<OP><<<grid, 256>>>(iters / 8, d_cyc, d_sink);      // warm
        hipEventRecord(e0);
        k_issue<OP><<<grid, 256>>>(iters, d_cyc, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid * 16);
        hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        const double inst = (double)iters * kPerBlock[OP];
        // waves per SIMD as placed (key without the pipe bits 6-7), waves of the SIMDs that hold the most common count, and
        // whether those waves really ran at the same time (largest start < smallest end)
        std::map<unsigned long long, std::vector<int>> by_simd;
        for (int i = 0; i < grid * 4; i++) by_simd[h[16 * (i / 4) + 4 * (i % 4) + 1] & ~0xc0ull].push_back(i);
        std::map<int, int> hist;
        for (auto& kv : by_simd) hist[(int)kv.second.size()]++;
        int mode = 0, best = 0;
        for (auto& kv : hist) if (kv.second > best) { best = kv.second; mode = kv.first; }
        std::vector<double> cw; double clock_sum = 0; int overlapped = 0, groups = 0;
        for (auto& kv : by_simd) {
            if ((int)kv.second.size() != mode) continue;
            unsigned long long s_max = 0, e_min = ~0ull;
            for (int i : kv.second) {
                const unsigned long long* r = &h[16 * (i / 4) + 4 * (i % 4)];
                cw.push_back((double)r[0] / inst);
                clock_sum += (double)r[0] / ((double)(r[3] - r[2]) * 10.0);      // cycles per ns
                s_max = std::max(s_max, r[2]); e_min = std::min(e_min, r[3]);
            }
            groups++; overlapped += s_max < e_min;
        }
        std::sort(cw.begin(), cw.end());
        const double cyc_wave = cw[cw.size() / 2];
        const double ginst = inst * grid * 4 / (ms * 1e-3) * 1e-9;
        printf(" | w=%d: %zu SIMDs, %d%% with %d waves (%d%% concurrent), %6.2f cyc/inst/wave = %5.2f /SIMD @%.2f GHz, chip %6.1f G/s", w, by_simd.size(),
               100 * best / (int)by_simd.size(), mode, groups ? 100 * overlapped / groups : 0, cyc_wave, cyc_wave / mode, clock_sum / cw.size(), ginst);
    }
    printf("\n");
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz; %d iterations; cyc = s_memtime ticks of the median wave; G/s = wave-instructions of all waves / wall time\n",
           prop.name, cus, prop.clockRate, iters);
    unsigned long long* d_cyc; float* d_sink;
    hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 8 * 16);
    hipMalloc(&d_sink, 4096);
    // how fast does s_memtime tick?  (a 100 MHz constant clock or the shader clock)
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_issue<FMA><<<cus, 256>>>(2000, d_cyc, d_sink);
        hipEventRecord(e0);
        k_issue<FMA><<<cus, 256>>>(200000, d_cyc, d_sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0[4]; hipMemcpy(c0, d_cyc, 32, hipMemcpyDeviceToHost);
        printf("s_memtime: %.1f ticks/us of hip-event time over a %.3f ms kernel, %.1f ticks per us of s_memrealtime (100 MHz)\n", (double)c0[0] / (ms * 1e3), ms,
               (double)c0[0] / ((double)(c0[3] - c0[2]) * 0.01));
    }
    run<FMA>(iters, d_cyc, d_sink, cus);
    run<FMA_DEP>(iters, d_cyc, d_sink, cus);
    run<PK_FMA>(iters, d_cyc, d_sink, cus);
    run<MUL_ADD_MIX>(iters, d_cyc, d_sink, cus);
    run<CMP_SGPR>(iters, d_cyc, d_sink, cus);
    run<CMP_VCC>(iters, d_cyc, d_sink, cus);
    run<CNDMASK>(iters, d_cyc, d_sink, cus);
    run<DPP_ADD>(iters, d_cyc, d_sink, cus);
    run<RCP>(iters, d_cyc, d_sink, cus);
    run<EXP>(iters, d_cyc, d_sink, cus);
    run<MOV>(iters, d_cyc, d_sink, cus);
    run<FMA_HALF_EXEC>(iters, d_cyc, d_sink, cus);
    run<FMA_QUARTER_EXEC>(iters, d_cyc, d_sink, cus);
    run<FMA_SALU>(iters, d_cyc, d_sink, cus);
    run<LDS_B128_BCAST>(iters, d_cyc, d_sink, cus);
    run<LDS_B128_2ADDR>(iters, d_cyc, d_sink, cus);
    run<LDS_B128_4ADDR>(iters, d_cyc, d_sink, cus);
    run<LDS_B128_LANE>(iters, d_cyc, d_sink, cus);
    run<LDS_B32_BCAST>(iters, d_cyc, d_sink, cus);
    run<MFMA_32>(iters, d_cyc, d_sink, cus);
    run<MFMA_16>(iters, d_cyc, d_sink, cus);
    run<MFMA_FMA>(iters, d_cyc, d_sink, cus);
    run<READLANE>(iters, d_cyc, d_sink, cus);
    run<PERMLANE_SWAP>(iters, d_cyc, d_sink, cus);
    return 0;
}
