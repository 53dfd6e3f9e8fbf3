// micro-benchmark: v_mfma_f32_16x16x32_f16 back to back — how many cycles per MFMA does a SIMD sustain with ONE wave or TWO waves on it, when consecutive
// MFMAs of a wave alternate between NACC accumulators (dependency distance NACC)?  hh_k_policy_w16's shared layer runs NACC = 2 with two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_dep.hip -o tools/ubench/mfma_dep && tools/ubench/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define A1 "v_mfma_f32_16x16x32_f16 a[0:3], v[10:13], v[14:17], a[0:3]\n"
#define A2 A1 "v_mfma_f32_16x16x32_f16 a[4:7], v[10:13], v[18:21], a[4:7]\n"
#define A4 A2 "v_mfma_f32_16x16x32_f16 a[8:11], v[10:13], v[14:17], a[8:11]\nv_mfma_f32_16x16x32_f16 a[12:15], v[10:13], v[18:21], a[12:15]\n"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "memory"
template <int NACC>
__global__ __launch_bounds__(512, 1) void k(unsigned long long *cyc, int iters) {
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (NACC == 1) asm volatile(REP16(REP4(A1)) ::: CLOB);       // 64 MFMAs
        else if constexpr (NACC == 2) asm volatile(REP16(A2 A2) ::: CLOB);     // 64
        else asm volatile(REP16(A4) ::: CLOB);                                  // 64
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned long long *d, h;
    hipMalloc(&d, 8);
    const int iters = 2000;
    for (int wgs : {256, 8})
    for (int waves : {4, 8})
        for (int nacc : {1, 2, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(64 * waves), 0, 0, d, iters);
                else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(64 * waves), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(64 * waves), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            const double per = (double)h / (iters * 64.0);
            printf("%3d workgroups, %d wave(s) per SIMD, %d accumulator(s) in turn: %6.1f cycles per MFMA of a wave = %5.1f cycles of the SIMD per MFMA (16.0 = the dense rate)\n", wgs, waves / 4, nacc, per,
                   per / (waves / 4));
            printf("    wall clock: %.1f us for %d MFMAs per wave -> %.2f ns per MFMA of a wave, %.3f PFLOP/s over the workgroups\n", ms * 1e3, iters * 64, ms * 1e6 / (iters * 64.0),
                   (double)wgs * waves * iters * 64.0 * 16384.0 / (ms * 1e-3) * 1e-15);
        }
    return 0;
}
