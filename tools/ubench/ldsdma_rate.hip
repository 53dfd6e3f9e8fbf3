// micro-benchmark: how fast does a CU pull L2-resident bytes into LDS, and how — global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave-instruction) against
// global_load_dwordx4 into registers (+ ds_write_b128) — with every CU doing it at once?  Each workgroup streams the same 1 MB buffer (a policy network's
// shared layer: L2 hits after the first pass) NPASS times, WV waves each taking every WV-th 1 KB piece; a vmcnt(0) every 8 pieces per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsdma_rate.hip -o tools/ubench/ldsdma_rate && tools/ubench/ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) unsigned char glb_u8;
#define PIECES 1024

template <int MODE> // 0: LDS-DMA, 1: global -> registers -> ds_write_b128, 2: global -> registers only
__global__ __launch_bounds__(512, 1) void k(const unsigned char *src, int npass, float *sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wv = blockDim.x >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int pass = 0; pass < npass; pass++)
        for (int p0 = wave * 8; p0 < PIECES; p0 += wv * 8) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 8; u++)
                    __builtin_amdgcn_global_load_lds((glb_u8 *)(src + (size_t)(p0 + u) * 1024 + lane * 16), (lds_u8 *)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)(p0 + u) * 1024 + lane * 16);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if constexpr (MODE == 1) *reinterpret_cast<float4 *>(lds + (wave * 8 + u) * 1024 + lane * 16) = v[u];
                    else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
            }
        }
    if (MODE == 2 && acc.x == 12345.0f) sink[0] = acc.x + acc.y + acc.z + acc.w;
    if (MODE != 2 && npass < 0) sink[0] = lds[threadIdx.x];
}

int main() {
    unsigned char *src; float *sink;
    hipMalloc(&src, PIECES * 1024); hipMemset(src, 0, PIECES * 1024); hipMalloc(&sink, 64);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount, npass = 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wv : {4, 8})
        for (int wgs : {1, 2}) {
            if (wv == 8 && wgs == 2) continue;
            for (int mode = 0; mode < 3; mode++) {
                const int lds = wgs == 2 ? 72 * 1024 : 128 * 1024; // 2 x 72 KB: two workgroups per CU | 128 KB: one
                auto run = [&](int np) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(ncu * wgs), dim3(64 * wv), lds, 0, src, np, sink);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(ncu * wgs), dim3(64 * wv), lds, 0, src, np, sink);
                    else hipLaunchKernelGGL(k<2>, dim3(ncu * wgs), dim3(64 * wv), lds, 0, src, np, sink);
                };
                hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                run(2);
                hipEventRecord(e0); run(npass); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)ncu * wgs * npass * PIECES * 1024;
                printf("%d waves x %d workgroup(s) per CU, %-34s %7.1f us per 1 MB pass and workgroup, %6.1f GB/s per CU, %5.2f TB/s chip\n", wv, wgs,
                       mode == 0 ? "global_load_lds_dwordx4:" : (mode == 1 ? "global_load_dwordx4 + ds_write_b128:" : "global_load_dwordx4 only:"),
                       ms * 1e3 / npass, bytes / ncu / (ms * 1e-3) * 1e-9, bytes / (ms * 1e-3) * 1e-12);
            }
        }
    return 0;
}
