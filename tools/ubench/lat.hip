// Tuning probe: issue/latency of dependent vs independent VALU chains with ONE wave on a SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 512
template <int MODE> __global__ void k(double *out, unsigned long long *cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 5;
    float f0 = (float)a, f1 = (float)b;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int i = 0; i < REP / 64; i++) {
            if (MODE == 0) { x0 = __builtin_fma(x0, b, a); }                                    // dependent fma f64 (1 chain)
            if (MODE == 1) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); }      // 2 chains
            if (MODE == 2) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
            if (MODE == 3) { x0 = x0 * b; }                                                     // dependent mul
            if (MODE == 4) { x0 = x0 + b; }                                                     // dependent add
            if (MODE == 5) { f0 = __builtin_fmaf(f0, f1, f1); }                                 // dependent fma f32
            if (MODE == 6) { i0 = i0 * 3 + i1; }                                                // dependent int mad
            if (MODE == 7) { x0 = 1.0 / x0; }                                                   // full f64 division
            if (MODE == 8) { x0 = __builtin_sqrt(x0); }                                         // f64 sqrt
            if (MODE == 9) { x0 = x0 > b ? x0 * b : x0 + a; }                                   // compare+select chain
            if (MODE == 10) { i0 = __builtin_amdgcn_mov_dpp(i0, 0x39, 0xf, 0xf, true) + 1; }    // dpp mov + add
            if (MODE == 11) { x0 = 1.0 / x0; x1 = 1.0 / x1; }                                   // 2 independent divisions
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + f0 + i0;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 8); hipMallocManaged(&cyc, 16 * 8);
#define RUN(M) k<M><<<1, 64>>>(out, cyc, 1.0000001, 0.9999999); k<M><<<1, 64>>>(out, cyc, 1.0000001, 0.9999999);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    hipDeviceSynchronize();
    const char *nm[] = {"fma64 x1 dep", "fma64 x2 chains", "fma64 x4 chains", "mul64 dep", "add64 dep", "fma32 dep", "int mad dep", "div64 dep", "sqrt64 dep", "cmp+sel64 dep", "dpp+add dep", "div64 x2"};
    for (int m = 0; m < 12; m++) printf("%-18s %8.2f cycles / iteration\n", nm[m], (double)cyc[m] / REP);
    return 0;
}
