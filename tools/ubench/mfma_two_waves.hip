// micro-benchmark: do the MFMAs of one wave and the vector instructions of ANOTHER wave on the same SIMD overlap?
// One 512-thread workgroup per CU = two waves per SIMD (wave w and wave w + 4 share SIMD w & 3).  Role masks: bit 0 = waves 0..3 run an MFMA
// stream (two accumulators alternating), bit 1 = waves 4..7 run a stream of independent v_fma_f32 (or ds_read_b128).  Prints cycles of each
// role alone and together, for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_two_waves.hip -o tools/ubench/mfma_two_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define M32 "v_mfma_f32_32x32x16_f16 a[0:15], v[10:13], v[14:17], a[0:15]\nv_mfma_f32_32x32x16_f16 a[16:31], v[10:13], v[18:21], a[16:31]\n"
#define M16 "v_mfma_f32_16x16x32_f16 a[0:3], v[10:13], v[14:17], a[0:3]\nv_mfma_f32_16x16x32_f16 a[4:7], v[10:13], v[18:21], a[4:7]\n"
#define VF "v_fma_f32 v30, v40, v41, v30\nv_fma_f32 v31, v42, v43, v31\nv_fma_f32 v32, v40, v43, v32\nv_fma_f32 v33, v42, v41, v33\n"
#define DS "ds_read_b128 v[48:51], v52\nds_read_b128 v[54:57], v52 offset:1024\n"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v30", "v31", "v32", "v33", "v40", "v41", "v42", "v43", "v48", "v49", \
             "v50", "v51", "v52", "v54", "v55", "v56", "v57", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16",  \
             "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "memory"

template <int MF, int FILL>
__global__ __launch_bounds__(512, 1) void k(unsigned long long *cyc, int iters, int roles) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    if (!((roles >> (mf ? 0 : 1)) & 1)) return;
    asm volatile("v_mov_b32 v52, 0" ::: "v52");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (mf) {
            if constexpr (MF == 32) asm volatile(REP16(M32) ::: CLOB);   // 32 MFMAs
            else asm volatile(REP16(M16) ::: CLOB);
        } else {
            if constexpr (FILL == 0) asm volatile(REP16(VF) ::: CLOB);   // 64 VALU
            else asm volatile(REP16(DS) "s_waitcnt lgkmcnt(0)\n" ::: CLOB);   // 32 ds_read_b128
        }
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}
template <int MF, int FILL>
static void run(const char *name, unsigned long long *cyc) {
    const int iters = 1000;
    unsigned long long h[2];
    double r[3][2];
    for (int roles = 1; roles <= 3; roles++) {
        hipMemset(cyc, 0, 16);
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<MF, FILL>), dim3(256), dim3(512), 0, 0, cyc, iters, roles);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        r[roles - 1][0] = (double)h[0] / iters;
        r[roles - 1][1] = (double)h[1] / iters;
    }
    printf("%-44s MFMA wave alone %7.0f | filler wave alone %7.0f | together: MFMA wave %7.0f, filler wave %7.0f   cycles per trip (32 MFMAs; 64 v_fma or 32 ds_read_b128)\n", name,
           r[0][0], r[1][1], r[2][0], r[2][1]);
}
int main() {
    unsigned long long *cyc;
    hipMalloc(&cyc, 16);
    run<32, 0>("32x32x16 beside another wave's v_fma_f32", cyc);
    run<16, 0>("16x16x32 beside another wave's v_fma_f32", cyc);
    run<32, 1>("32x32x16 beside another wave's ds_read_b128", cyc);
    run<16, 1>("16x16x32 beside another wave's ds_read_b128", cyc);
    return 0;
}
