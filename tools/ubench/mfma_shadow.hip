// micro-benchmark: how many (and which) instructions of the SAME wave hide in the shadow of v_mfma_f32_32x32x16_f16, one wave per SIMD.
// Raw instruction streams in inline asm: per loop trip 8 MFMAs alternating between two accumulators (the shape of hh_k_policy_w's
// shared-layer loop), K filler instructions of one kind after each MFMA.  Prints cycles per MFMA for K = 0..8.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shadow.hip -o tools/ubench/mfma_shadow && tools/ubench/mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)

#define MF0 "v_mfma_f32_32x32x16_f16 a[0:15], v[10:13], v[14:17], a[0:15]\n"
#define MF1 "v_mfma_f32_32x32x16_f16 a[16:31], v[10:13], v[18:21], a[16:31]\n"
// fillers write v[30..37] / read v[40..47] (never MFMA operands)
#define F_FMA "v_fma_f32 v30, v40, v41, v30\n"
#define F_FMA2 "v_fma_f32 v31, v42, v43, v31\n"
#define F_EXP "v_exp_f32 v32, v44\n"
#define F_RCP "v_rcp_f32 v33, v45\n"
#define F_PK "v_pk_fma_f32 v[34:35], v[40:41], v[42:43], v[34:35]\n"
#define F_CVT "v_cvt_pk_f16_f32 v36, v40, v41\n"
#define F_ACCRD "v_accvgpr_read_b32 v37, a40\n"
#define F_ACCRD_HOT "v_accvgpr_read_b32 v37, a64\n"
#define F_SALU "s_add_u32 s20, s20, 1\n"
#define F_DSRD "ds_read_b128 v[48:51], v52\n"
#define F_NOP "s_nop 0\n"
#define F_MIX "v_fma_mixlo_f16 v36, v40, v41, v42\n"
#define F_SDWA "v_cvt_f32_f16_sdwa v36, v40 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define F_ACCWR "v_accvgpr_write_b32 a40, v40\n"
#define F_GLDS "s_mov_b32 m0, s21\nglobal_load_lds_dwordx4 v[54:55], off\n"

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define BODY(F)                                                                                                                              \
    asm volatile(REP4(MF0 F MF1 F) ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v30", "v31", "v32", "v33", \
                 "v34", "v35", "v36", "v37", "v40", "v41", "v42", "v43", "v44", "v45", "v48", "v49", "v50", "v51", "v52", "s20", "a0", "a1", "a2", "a3",  \
                 "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23",   \
                 "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a40", "a64", "memory")
#define FILL(X)                                                                \
    if constexpr (K == 0) BODY("");                                            \
    else if constexpr (K == 1) BODY(X);                                        \
    else if constexpr (K == 2) BODY(X X);                                      \
    else if constexpr (K == 3) BODY(X X X);                                    \
    else if constexpr (K == 4) BODY(X X X X);                                  \
    else if constexpr (K == 5) BODY(X X X X X);                                \
    else if constexpr (K == 6) BODY(X X X X X X);                              \
    else if constexpr (K == 8) BODY(X X X X X X X X);                          \
    else BODY(X X X X X X X X X X X X)
        if constexpr (KIND == 0) { FILL(F_FMA); }
        else if constexpr (KIND == 1) { FILL(F_EXP); }
        else if constexpr (KIND == 2) { FILL(F_PK); }
        else if constexpr (KIND == 3) { FILL(F_CVT); }
        else if constexpr (KIND == 4) { FILL(F_ACCRD); }
        else if constexpr (KIND == 5) { FILL(F_SALU); }
        else if constexpr (KIND == 6) { FILL(F_DSRD); }
        else if constexpr (KIND == 7) { FILL(F_NOP); }
        else if constexpr (KIND == 8) { FILL(F_FMA F_FMA2); }
        else if constexpr (KIND == 9) { FILL(F_EXP F_RCP); }
        else if constexpr (KIND == 10) { FILL(F_MIX); }
        else if constexpr (KIND == 11) { FILL(F_SDWA); }
        else if constexpr (KIND == 12) { FILL(F_ACCWR); }
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x * 7) & 4095];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int K>
static double run(float *out, unsigned long long *cyc) {
    const int iters = 2000;
    unsigned long long h;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / (iters * 8.0);
}
template <int KIND>
static void row(const char *name, float *out, unsigned long long *cyc) {
    printf("%-34s K=0 %5.1f | 1 %5.1f | 2 %5.1f | 3 %5.1f | 4 %5.1f | 5 %5.1f | 6 %5.1f | 8 %5.1f | 12 %5.1f   cycles per MFMA\n", name, run<KIND, 0>(out, cyc),
           run<KIND, 1>(out, cyc), run<KIND, 2>(out, cyc), run<KIND, 3>(out, cyc), run<KIND, 4>(out, cyc), run<KIND, 5>(out, cyc), run<KIND, 6>(out, cyc),
           run<KIND, 8>(out, cyc), run<KIND, 12>(out, cyc));
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4 * 256 * 256);
    hipMalloc(&cyc, 8);
    row<0>("v_fma_f32 (one chain)", out, cyc);
    row<8>("v_fma_f32 x2 (two chains) per K", out, cyc);
    row<1>("v_exp_f32", out, cyc);
    row<9>("v_exp_f32 + v_rcp_f32 per K", out, cyc);
    row<2>("v_pk_fma_f32", out, cyc);
    row<3>("v_cvt_pk_f16_f32", out, cyc);
    row<4>("v_accvgpr_read_b32 (idle agpr)", out, cyc);
    row<5>("s_add_u32", out, cyc);
    row<6>("ds_read_b128", out, cyc);
    row<7>("s_nop 0", out, cyc);
    row<10>("v_fma_mixlo_f16", out, cyc);
    row<11>("v_cvt_f32_f16_sdwa", out, cyc);
    row<12>("v_accvgpr_write_b32", out, cyc);
    return 0;
}
