// micro-benchmark: the shared-layer step of hh_k_policy_w16 on its own — four ds_read_b128 weight fragments (1 KB each, conflict-free) requested a step ahead,
// six v_mfma_f32_16x16x32_f16 on two accumulators in turn — with one and with two waves per SIMD, with and without the reads: does the LDS read path keep up
// with the matrix pipe, and what do the chunk barrier and the LDS-DMA requests add?  (dense rate: 6 MFMAs x 16 cycles = 96 cycles per step and wave)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds_step.hip -o tools/ubench/mfma_lds_step && tools/ubench/mfma_lds_step
#include <hip/hip_runtime.h>
#include <stdio.h>
#define RD(o) "ds_read_b128 v[30:33], v60 offset:" #o "\nds_read_b128 v[34:37], v60 offset:" #o "+1024\nds_read_b128 v[38:41], v60 offset:" #o "+2048\nds_read_b128 v[42:45], v60 offset:" #o "+3072\n"
#define RD2(o) "ds_read_b128 v[46:49], v60 offset:" #o "\nds_read_b128 v[50:53], v60 offset:" #o "+1024\nds_read_b128 v[54:57], v60 offset:" #o "+2048\nds_read_b128 v[62:65], v60 offset:" #o "+3072\n"
#define MF(a, b, c, d) "v_mfma_f32_16x16x32_f16 a[0:3], v[" a "], v[10:13], a[0:3]\nv_mfma_f32_16x16x32_f16 a[4:7], v[" c "], v[10:13], a[4:7]\n" \
                       "v_mfma_f32_16x16x32_f16 a[0:3], v[" b "], v[10:13], a[0:3]\nv_mfma_f32_16x16x32_f16 a[4:7], v[" d "], v[10:13], a[4:7]\n" \
                       "v_mfma_f32_16x16x32_f16 a[0:3], v[" a "], v[14:17], a[0:3]\nv_mfma_f32_16x16x32_f16 a[4:7], v[" c "], v[14:17], a[4:7]\n"
// the same step with the four reads of the NEXT step placed between this step's MFMAs instead of in a burst in front of them
#define R1(r, o) "ds_read_b128 v[" r "], v60 offset:" #o "\n"
#define M1(d, a, b) "v_mfma_f32_16x16x32_f16 " d ", v[" a "], v[" b "], " d "\n"
#define STEP_IL(a0, a1, a2, a3, n0, n1, n2, n3, o)                                                                                    \
    M1("a[0:3]", a0, "10:13") R1(n0, o) M1("a[4:7]", a2, "10:13") R1(n1, o + 1024) M1("a[0:3]", a1, "10:13") R1(n2, o + 2048)         \
    M1("a[4:7]", a3, "10:13") R1(n3, o + 3072) M1("a[0:3]", a0, "14:17") M1("a[4:7]", a2, "14:17")
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46",  \
             "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v60", "v62", "v63", "v64", "v65", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "memory"
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) unsigned char glb_u8;
// MODE bit 0: the fragment reads; bit 1: a workgroup barrier (+ vmcnt(0)) every 8 steps = a chunk; bit 2: the chunk's LDS-DMA requests (32 KB per chunk and
// workgroup out of an L2-resident megabyte: 8 per wave with four waves, 4 with eight), two per step in the chunk's first steps
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(int iters, float *sink, const unsigned char *src, int data) {
    extern __shared__ __align__(16) unsigned char lds[];
    // operand data: zeros, or (iters < 0 never; `data` != 0) pseudo-random fp16 values in (-1, 1) in the LDS fragments and in the B registers: the matrix pipe's
    // power, and with it the clock the chip sustains, depends on the bits it multiplies
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 97u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned lo = 0x3000u | (h & 0x8BFFu), hi = 0x3000u | ((h >> 16) & 0x8BFFu); // sign + exponent 12..14 + random mantissa
        reinterpret_cast<unsigned *>(lds)[i] = data ? (lo | (hi << 16)) : 0u;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6, npw = 32 / nw;
    {
        unsigned r = data ? reinterpret_cast<unsigned *>(lds)[threadIdx.x] : 0u;
        asm volatile("v_mov_b32 v10, %0\nv_mov_b32 v11, %0\nv_mov_b32 v12, %0\nv_mov_b32 v13, %0\nv_mov_b32 v14, %0\nv_mov_b32 v15, %0\nv_mov_b32 v16, %0\nv_mov_b32 v17, %0\n"
                     "v_mov_b32 v30, %0\nv_mov_b32 v31, %0\nv_mov_b32 v34, %0\nv_mov_b32 v38, %0\nv_mov_b32 v42, %0\nv_mov_b32 v46, %0\nv_mov_b32 v50, %0\nv_mov_b32 v54, %0\nv_mov_b32 v62, %0\n"
                     ::"v"(r) : CLOB);
    }
    asm volatile("v_lshlrev_b32 v60, 4, %0" ::"v"(threadIdx.x & 63) : "v60");
    for (int it = 0; it < iters; it++) { // two steps per iteration: the reads of one step go out before the MFMAs of the other
        if constexpr ((MODE & 9) == 9) asm volatile("s_waitcnt lgkmcnt(0)\n" STEP_IL("30:33", "34:37", "38:41", "42:45", "46:49", "50:53", "54:57", "62:65", 4096)
                                                       "s_waitcnt lgkmcnt(0)\n" STEP_IL("46:49", "50:53", "54:57", "62:65", "30:33", "34:37", "38:41", "42:45", 0) ::: CLOB);
        else if constexpr (MODE & 1) asm volatile(RD2(4096) "s_waitcnt lgkmcnt(4)\n" MF("30:33", "34:37", "38:41", "42:45") RD(0) "s_waitcnt lgkmcnt(4)\n" MF("46:49", "50:53", "54:57", "62:65") ::: CLOB);
        else asm volatile(MF("30:33", "34:37", "38:41", "42:45") MF("46:49", "50:53", "54:57", "62:65") ::: CLOB);
        if constexpr (MODE & 4) {
            const int st2 = it & 3; // iteration = 2 steps; a chunk = 4 iterations
            if (st2 * 2 < npw) {     // two requests per iteration while the wave has pieces left
                const unsigned char *g = src + (size_t)(((it >> 2) & 31) * 32 + wave * npw + st2 * 2) * 1024 + lane * 16;
                unsigned char *d = lds + 32768 + (wave * npw + st2 * 2) * 1024;
                __builtin_amdgcn_global_load_lds((glb_u8 *)g, (lds_u8 *)d, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_u8 *)g, (lds_u8 *)d, 16, 1024, 0);
            }
        }
        if constexpr (MODE & 2) { if ((it & 3) == 3) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_barrier" ::: "memory"); } }
    }
    if (iters < 0) sink[0] = lds[threadIdx.x];
}
int main() {
    float *sink; hipMalloc(&sink, 64);
    unsigned char *src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[8] = {"6 MFMAs only", "+ 4 ds_read_b128", "+ barrier per chunk", "+ reads + barrier", "+ LDS-DMA requests", "+ reads + LDS-DMA", "+ barrier + LDS-DMA", "+ reads + barrier + LDS-DMA (the kernel's step)"};
    for (int data : {0, 1})
    for (int waves : {4, 8})
        for (int mode : {0, 1, 9, 7, 15}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
#define LAUNCH(M) case M: hipFuncSetAttribute(reinterpret_cast<const void *>(k<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(k<M>, dim3(256), dim3(64 * waves), 65536, 0, iters, sink, src, data); break;
                switch (mode) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(9) LAUNCH(15) }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double ns_step = ms * 1e6 / (2.0 * iters);
            printf("%s operands, %d wave(s) per SIMD, %-48s %6.1f ns per step of a wave = %5.1f ns of the SIMD per step; %.2f PFLOP/s\n", data ? "random fp16" : "all-zero   ", waves / 4, mode == 9 ? "+ 4 ds_read_b128 BETWEEN the MFMAs" : (mode == 15 ? "+ reads between + barrier + LDS-DMA" : names[mode]), ns_step, ns_step / (waves / 4),
                   256.0 * waves * 2.0 * iters * 6 * 16384.0 / (ms * 1e-3) * 1e-15);
        }
    return 0;
}
