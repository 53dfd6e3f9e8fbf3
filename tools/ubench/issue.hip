// Tuning probe: what ONE wave on a SIMD pays per instruction of each class on gfx950 (issue slots, taken / not-taken branches, the
// exec-mask skip pattern of divergent ifs, SGPR spill traffic).  Every case is REP copies of a pattern inside an outer loop of 64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define STR2(x) #x
#define STR(x) STR2(x)
#define REP 256
#define CASE(M, BODY)                                                                                     \
    if (MODE == M) {                                                                                      \
        asm volatile(".rept " STR(REP) "\n" BODY "\n.endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1) : "s"(zero), "v"(d0), "v"(d1) : "vcc", "scc", "s40", "s41", "s42", "s43", "memory"); \
    }
template <int MODE> __global__ void k(int *out, unsigned long long *cyc, int zero_in, double da, double db) {
    int v0 = threadIdx.x, v1 = threadIdx.x + 1, v2 = 3, v3 = 4;
    int s0 = zero_in + 1, s1 = zero_in + 2;
    const int zero = __builtin_amdgcn_readfirstlane(zero_in);
    double d0 = da + threadIdx.x, d1 = db;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
        CASE(0, "v_mov_b32 %0, %1")                                                            // independent VALU
        CASE(1, "s_mov_b32 %4, 0x12345678")                                                    // SALU stream
        CASE(2, "s_mov_b32 %4, 0x12345678\n v_mov_b32 %0, %1")                                 // SALU + VALU alternating (2 instr)
        CASE(3, "s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 1f\n v_mov_b32 %0, %1\n1:")               // taken uniform branch over 1 instr (2 + branch)
        CASE(4, "s_cmp_eq_u32 %6, 1\n s_cbranch_scc1 1f\n v_mov_b32 %0, %1\n1:")               // not-taken uniform branch (3 instr)
        CASE(5, "v_cmp_eq_u32 vcc, 1, %1\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1f\n v_mov_b32 %0, %1\n1:\n s_or_b64 exec, exec, s[40:41]")  // divergent if, mostly... lane 0 only: not skipped
        CASE(6, "v_cmp_eq_u32 vcc, 1000, %1\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1f\n v_mov_b32 %0, %1\n1:\n s_or_b64 exec, exec, s[40:41]")  // divergent if nobody takes: skipped (taken branch)
        CASE(7, "v_cmp_eq_u32 vcc, 1, %1\n s_nop 1\n v_cndmask_b32 %0, %2, %3, vcc")           // compare + select (3 instr incl. nop)
        CASE(8, "v_writelane_b32 %2, %4, 3\n v_readlane_b32 %5, %2, 3")                        // spill + reload of one SGPR (2 instr)
        CASE(9, "v_fma_f64 %7, %7, %8, %8")                                                    // dependent f64 fma
        CASE(10, "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n s_mov_b32 %4, 1\n s_mov_b32 %5, 2")    // 2 VALU + 2 SALU
        CASE(11, "s_nop 0")
        CASE(12, "v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")        // DPP mov
        CASE(13, "s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 1f\n .rept 8\n v_mov_b32 %0, %1\n .endr\n1:") // taken branch over 8 instrs
        if (MODE >= 32 && MODE <= 35) { /* half / quarter-empty waves: does the SIMD skip the empty 16-lane passes? */
            const unsigned long long keep = MODE == 32 || MODE == 34 ? 0xffffffffULL : 0xffffULL;
            const unsigned long long ex = __builtin_amdgcn_read_exec();
            asm volatile("s_mov_b64 exec, %0" ::"s"(keep));
            if (MODE <= 33) asm volatile(".rept " STR(REP) "\n v_mov_b32 %0, %1\n.endr" : "+v"(v0) : "v"(v1));
            else asm volatile(".rept " STR(REP) "\n v_fma_f64 %0, %0, %1, %1\n.endr" : "+v"(d0) : "v"(d1));
            asm volatile("s_mov_b64 exec, %0" ::"s"(ex));
        }
        CASE(14, "v_mul_lo_u32 %0, %1, %2")
        CASE(15, "v_mul_hi_u32 %0, %1, %2")
        CASE(16, "v_mad_u64_u32 %7, vcc, %1, %2, %8")
        CASE(17, "v_mul_f64 %7, %8, %8")
        CASE(18, "v_rsq_f64 %7, %8")
        CASE(19, "v_rcp_f64 %7, %8")
        CASE(20, "v_cvt_f64_u32 %7, %1")
        CASE(21, "v_ldexp_f64 %7, %8, %1")
        CASE(22, "v_div_scale_f64 %7, vcc, %8, %8, %7")
        CASE(23, "v_div_fixup_f64 %7, %8, %8, %7")
        CASE(24, "v_permlane32_swap_b32 %0, %1")
        CASE(25, "v_cmp_lt_f64 vcc, %7, %8\n s_nop 1\n v_cndmask_b32 %0, %2, %3, vcc")
        CASE(26, "v_max_f64 %7, %7, %8")
        CASE(27, "v_floor_f64 %7, %8")
        CASE(28, "v_cvt_i32_f64 %0, %8")
        CASE(29, "v_sqrt_f64 %7, %8")
        CASE(30, "v_add_u32 %0, %1, %2")
        CASE(31, "v_lshlrev_b64 %7, 3, %8")
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + (int)d0;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 4); hipMallocManaged(&cyc, 40 * 8);
#define RUN(M) k<M><<<1, 64>>>(out, cyc, 0, 1.0000001, 0.9999999); k<M><<<1, 64>>>(out, cyc, 0, 1.0000001, 0.9999999);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29) RUN(30) RUN(31) RUN(32) RUN(33) RUN(34) RUN(35)
    hipDeviceSynchronize();
    const char *nm[] = {"v_mov", "s_mov", "s_mov+v_mov", "uniform branch taken (cmp,br | skipped 1)", "uniform branch not taken (cmp,br,v_mov)", "divergent if entered (5 instr)",
                        "divergent if skipped (4 instr + taken br)", "v_cmp+nop+cndmask", "writelane+readlane", "fma64 dep", "2 v_mov + 2 s_mov", "s_nop 0", "v_mov dpp", "uniform branch taken over 8", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_f64", "v_rsq_f64", "v_rcp_f64", "v_cvt_f64_u32", "v_ldexp_f64", "v_div_scale_f64", "v_div_fixup_f64", "v_permlane32_swap", "v_cmp_f64+nop+cndmask", "v_max_f64", "v_floor_f64", "v_cvt_i32_f64", "v_sqrt_f64", "v_add_u32", "v_lshlrev_b64", "v_mov, lanes 0-31 only", "v_mov, lanes 0-15 only", "fma64 dep, lanes 0-31 only", "fma64 dep, lanes 0-15 only"};
    for (int m = 0; m < 36; m++) printf("%-45s %8.2f cycles / pattern\n", nm[m], (double)cyc[m] / (64.0 * REP));
    return 0;
}
