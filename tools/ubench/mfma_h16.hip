// micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 in the access pattern of hh_k_policy_h's L2 loop
// (4 accumulator tiles x 3 products per block, one wave per SIMD / two waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NT>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    h8 a, al, b[NT], bl[NT];
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); al[i] = (_Float16)(0.0001f * i); }
    for (int t = 0; t < NT; t++) for (int i = 0; i < 8; i++) { b[t][i] = (_Float16)(0.002f * (t + i)); bl[t][i] = (_Float16)(0.0002f * (t - i)); }
    f16v acc[NT];
    for (int t = 0; t < NT; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[t], acc[t], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NT; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4 * 256 * 2048); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int grid : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("NT=4 grid %4d (%d waves/SIMD): %.1f ticks per MFMA\n", grid, grid / 256, (double)h / (iters * 12.0));
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("NT=1 grid %4d (%d waves/SIMD): %.1f ticks per MFMA (same accumulator back to back)\n", grid, grid / 256, (double)h / (iters * 3.0));
    }
    return 0;
}
