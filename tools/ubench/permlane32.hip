// semantics probe: v_permlane32_swap_b32 on gfx950 (which half receives what)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);   // (vdst_old = x, src0_old = y)
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r[0] lanes 0,1,31,32,33,63: %d %d %d %d %d %d\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("r[1] lanes 0,1,31,32,33,63: %d %d %d %d %d %d\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
