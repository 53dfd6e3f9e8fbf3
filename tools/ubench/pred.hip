// Tuning probe (round 6): what predicate logic costs ONE wave on a SIMD of gfx950 — a lane mask written by a VALU compare and consumed by the scalar unit
// (s_and_b64 / s_or_b64 chains: how the compiler spells `a & b & c` on comparison results), against VALU-only spellings of the same test.
// Every case is REP copies of a pattern inside an outer loop of 64; cycles per pattern from s_memtime around the loop (one wave, one workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define STR2(x) #x
#define STR(x) STR2(x)
#define REP 128
#define CASE(M, BODY)                                                                                     \
    if (MODE == M) {                                                                                      \
        asm volatile(".rept " STR(REP) "\n" BODY "\n.endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+v"(d0), "+v"(d2) : "s"(zero), "v"(d1) : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory"); \
    }
template <int MODE> __global__ void k(int *out, unsigned long long *cyc, int zero_in, double da, double db) {
    int v0 = threadIdx.x, v1 = threadIdx.x + 1, v2 = 3, v3 = 4;
    int s0 = zero_in + 1, s1 = zero_in + 2;
    const int zero = __builtin_amdgcn_readfirstlane(zero_in);
    double d0 = da + threadIdx.x, d1 = db, d2 = da * 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
        CASE(0, "v_cmp_le_f64 s[40:41], %6, %9")                                                                  // compare alone (mask to an SGPR pair)
        CASE(1, "v_cmp_le_f64 s[40:41], %6, %9\n s_and_b64 s[42:43], s[40:41], s[42:43]")                          // compare -> dependent scalar AND
        CASE(2, "v_cmp_le_f64 s[40:41], %6, %9\n v_cmp_le_f64 s[44:45], %7, %9\n s_and_b64 s[42:43], s[40:41], s[44:45]") // two compares, one AND
        CASE(3, "v_cmp_le_f64 s[40:41], %6, %9\n v_cmp_le_f64 s[44:45], %7, %9\n v_cmp_lt_f64 s[46:47], %7, %6\n s_and_b64 s[42:43], s[40:41], s[44:45]\n s_and_b64 s[42:43], s[42:43], s[46:47]") // three compares, two ANDs
        CASE(4, "v_max_f64 %6, |%6|, |%7|\n v_cmp_le_f64 s[40:41], %6, %9")                                        // the two-sided box test as max + one compare
        CASE(5, "v_cmp_le_f64 vcc, %6, %9\n v_cndmask_b32 %0, %2, %3, vcc")                                        // compare -> select (VALU reads the mask the VALU wrote)
        CASE(6, "v_cmp_le_f64 s[40:41], %6, %9\n s_and_b64 vcc, s[40:41], s[42:43]\n v_cndmask_b32 %0, %2, %3, vcc")   // compare -> scalar AND -> select
        CASE(7, "s_and_b64 vcc, s[40:41], s[42:43]\n v_cndmask_b32 %0, %2, %3, vcc")                               // scalar AND -> select
        CASE(8, "v_cmp_le_f64 s[40:41], %6, %9\n v_cmp_le_f64 s[44:45], %7, %9\n v_cndmask_b32 %0, 0, 1, s[40:41]\n v_cndmask_b32 %1, 0, 1, s[44:45]\n v_and_b32 %0, %0, %1") // predicates as 0 / 1 integers in VGPRs
        CASE(9, "v_cmp_le_f64 vcc, %6, %9\n v_cmpx_le_f64 exec, %7, %9")                                           // (not used) 
        CASE(10, "v_cmp_le_f64 s[40:41], %6, %9\n s_nop 0\n s_nop 0\n s_and_b64 s[42:43], s[40:41], s[42:43]")     // does padding change the dependent AND's cost?
        CASE(11, "v_cmp_le_f64 s[40:41], %6, %9\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n s_and_b64 s[42:43], s[40:41], s[42:43]") // ... or independent VALU work in between?
        CASE(12, "v_readlane_b32 %4, %2, 3\n s_add_u32 %5, %4, %5")                                                // readlane -> scalar use
        CASE(13, "v_cmp_class_f64 s[40:41], %6, %2")
        CASE(14, "v_cmp_le_f64 s[40:41], |%6|, %9")                                                                // source modifier: VOP3 encoding
        CASE(15, "s_mov_b32 s40, 0x12345678\n s_mov_b32 s41, 0x40240000\n v_cmp_le_f64 s[42:43], %6, s[40:41]")    // literal materialisation + compare against it
        CASE(16, "s_mov_b32 s40, 0x12345678\n s_mov_b32 s41, 0x40240000\n v_mul_f64 %6, %9, s[40:41]")                                          // literal -> f64 multiply (3)
        CASE(17, "s_mov_b32 s40, 0x12345678\n s_mov_b32 s41, 0x40240000\n v_mul_f64 %6, %9, s[40:41]\n s_mov_b32 s44, 0x12345678\n s_mov_b32 s45, 0x40240000\n v_mul_f64 %7, %9, s[44:45]") // two register pairs in turn (6): RAW only
        CASE(18, "v_mul_f64 %6, %9, s[40:41]")                                                                                                   // loop-invariant SGPR operand (1)
        CASE(19, "s_mov_b32 s40, 0x12345678\n s_mov_b32 s41, 0x40240000\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n v_mul_f64 %6, %9, s[40:41]") // four VALU between write and read (7)
        CASE(20, "v_mul_f64 %6, %9, s[40:41]\n s_mov_b32 s40, 0x12345678\n s_mov_b32 s41, 0x40240000")                                          // WAR: the scalar write follows the vector read (3)
        CASE(21, "v_mul_f64 %6, %9, s[40:41]\n s_mov_b32 s44, 0x12345678\n s_mov_b32 s45, 0x40240000")                                          // no dependency at all (3)
        CASE(22, "v_cmp_le_f64 s[40:41], %6, %9\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n s_and_b64 s[42:43], s[40:41], s[42:43]") // four VALU between compare and AND (6)
        CASE(23, "v_cmp_le_f64 s[40:41], %6, %9\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %2\n s_and_b64 s[42:43], s[40:41], s[42:43]") // six (8)
        CASE(24, "v_cmp_le_f64 vcc, %6, %9\n s_cbranch_vccz 1f\n v_mov_b32 %2, %3\n1:")                                                         // compare -> branch on vcc
        CASE(25, "v_cmp_le_f64 s[40:41], %6, %9\n s_cmp_lg_u64 s[40:41], 0\n s_cbranch_scc0 1f\n v_mov_b32 %2, %3\n1:")                        // q_any: compare -> s_cmp -> branch
        CASE(26, "s_mov_b32 s40, 1\n v_add_u32 %0, s40, %1")                                                                                     // scalar -> 32-bit vector operand (2)
        CASE(27, "v_cmp_le_f64 s[40:41], %6, %9\n v_cmp_le_f64 s[44:45], %7, %9\n v_cndmask_b32 %0, 0, 1, s[40:41]\n v_cndmask_b32 %0, 0, %0, s[44:45]") // a & b as a chain of selects (4)
        CASE(28, "v_cmp_le_f64 vcc, %6, %9\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc")                                     // one mask, two selects (a 64-bit select) (3)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + (int)d0 + (int)d2;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 4); hipMallocManaged(&cyc, 64 * 8);
#define RUN(M) k<M><<<1, 64>>>(out, cyc, 0, 1.0000001, 0.9999999); k<M><<<1, 64>>>(out, cyc, 0, 1.0000001, 0.9999999);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28)
    hipDeviceSynchronize();
    const char *nm[] = {"v_cmp_f64 -> sgpr", "v_cmp -> s_and (2)", "2 v_cmp -> s_and (3)", "3 v_cmp -> 2 s_and (5)", "v_max_f64 |a| |b| + v_cmp (2)", "v_cmp vcc -> v_cndmask (2)",
                        "v_cmp -> s_and vcc -> v_cndmask (3)", "s_and vcc -> v_cndmask (2)", "2 v_cmp + 2 cndmask 0/1 + v_and (5)", "-", "v_cmp, 2 s_nop, s_and (4)", "v_cmp, 2 v_mov, s_and (4)",
                        "v_readlane -> s_add (2)", "v_cmp_class_f64", "v_cmp |a| (VOP3)", "2 s_mov literal + v_cmp (3)",
                        "2 s_mov literal -> v_mul_f64 (3)", "same, two register pairs in turn (6)", "v_mul_f64 with a resident SGPR pair (1)", "2 s_mov, 4 v_mov, v_mul_f64 (7)",
                        "WAR: v_mul_f64 then 2 s_mov of its pair (3)", "v_mul_f64 + 2 unrelated s_mov (3)", "v_cmp, 4 v_mov, s_and (6)", "v_cmp, 6 v_mov, s_and (8)",
                        "v_cmp vcc -> s_cbranch_vccz (2-3)", "v_cmp -> s_cmp_lg_u64 -> s_cbranch (3-4)", "s_mov -> v_add_u32 operand (2)", "a & b as two selects (4)", "one mask, two selects (3)"};
    for (int m = 0; m < 29; m++) if (m != 9) printf("%-45s %8.2f cycles / pattern\n", nm[m], (double)cyc[m] / (64.0 * REP));
    return 0;
}
