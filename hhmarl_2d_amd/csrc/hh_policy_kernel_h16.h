/*
 * hh_policy_kernel_h16.h — the policy forward of hh_policy_kernel.h at the fp16 MFMA rate WITHOUT giving up fp32 accuracy.
 *
 * gfx950 has no tf32-like mode: fp32-in MFMA runs at the vector rate (157 TFLOP/s), fp16-in MFMA 16x faster.  Every operand x
 * (weights: on the host, once; activations: in the epilogue that produces them) is split into two halves
 *     hi = fp16(x),  lo = fp16(x - hi)          |x - hi - lo| <= max(2^-22 |x|, 2^-25)   (fp16 subnormals are kept: the
 *                                                 residual of a value below 2^-3 lands there, spaced 2^-24)
 * and a product is accumulated as  hi_a hi_b + hi_a lo_b + lo_a hi_b  in fp32 by three v_mfma_f32_32x32x16_f16 (the dropped
 * lo_a lo_b term is below 2^-22 |a b|): 3 instructions of 32 cycles cover 16 k-steps where the fp32 MFMA needs 8 of 64 — 5.3x
 * fewer matrix-pipe cycles, with an error (measured against a float64 forward, tools/policy_error.py: max 1.0e-6 / mean 1.9e-7 on the
 * logits; the fp32 MFMA kernel 7.5e-7 / 1.3e-7; PyTorch's own fp32 CPU forward 8e-7 / 1.0e-7) far inside the 1e-5 the parity tests allow.  Same tiling as the fp32 kernel: a 256-thread workgroup = 32 rows of one
 * network, each wave 128 of the 512 columns (4 MFMA tiles), activations in one LDS tile (hi and lo planes, 32 KB each) that the
 * next layer's output overwrites in place, two workgroups per CU.  Operand layout: [k/16][(k/8)&1][col or row][8 halves] — lane l of
 * a 32x32x16 MFMA holds A[l&31][8 (l>>5) .. +7] / B[8 (l>>5) .. +7][l&31], so one 16-byte access per lane is one fragment.
 * Weight fragments are requested two 16-k blocks (2 x 12 MFMAs = 768 matrix-pipe cycles) ahead of their use.
 */
#ifndef HH_POLICY_KERNEL_H16_H
#define HH_POLICY_KERNEL_H16_H

typedef _Float16 hh_h8 __attribute__((ext_vector_type(8)));

struct HhpNetH {
    const float4 *w1h, *w1l;   /* [2][2][512] fragments (8 halves each) */
    const float4 *wovh, *wovl; /* [7][2][128] */
    const float4 *wsh, *wsl;   /* [32][2][512] */
    const float4 *wah, *wal;   /* [32][2][32] */
};
struct HhpBankH {
    HhpNetH net[HH_POLICY_MAX_NETS];
};

/* index (in halves) of element (k, col) of a [K x J] operand */
__host__ __device__ inline size_t hhp_hidx(int k, int col, int J) { return ((size_t)((k >> 4) * 2 + ((k >> 3) & 1)) * J + col) * 8 + (k & 7); }
/* the same for the 32-row LDS activation planes, row slot XOR-swizzled by the plane like hhp_aidx */
__device__ __forceinline__ int hhp_haidx(int k, int row) {
    const int plane = (k >> 4) * 2 + ((k >> 3) & 1);
    return (plane * 32 + (row ^ (plane & 7))) * 8 + (k & 7);
}
__device__ __forceinline__ void hhp_split_store(_Float16 *__restrict__ hi, _Float16 *__restrict__ lo, int idx, float v) {
    const _Float16 h = (_Float16)v;
    hi[idx] = h;
    lo[idx] = (_Float16)(v - (float)h);
}
__device__ __forceinline__ hh_h8 hhp_as_h8(const float4 &v) {
    union { float4 f; hh_h8 h; } u;
    u.f = v;
    return u.h;
}

/* NT tiles over KB 16-k blocks: acc += Ahi Bhi + Ahi Blo + Alo Bhi.  A fragments from the LDS planes, B fragments from global.
 * Three register sets of weight fragments rotate BY NAME (the block loop is unrolled by three): block kb computes from set kb % 3
 * while the loads of block kb + 2 land in set (kb + 2) % 3.  Rotating by register moves instead would read the set that was just
 * requested and make the wave wait for it — a prefetch distance of one block (384 cycles) where an L2 round trip under load is
 * longer (measured: 37 % of the wave cycles waiting). */
template <int NT>
struct HhpBSet {
    float4 h[NT], l[NT];
};
template <int NT>
__device__ __forceinline__ void hhp_gemm_h(const float4 *__restrict__ a_hi, const float4 *__restrict__ a_lo, int kb0, int KB,
                                           const float4 *__restrict__ b_hi, const float4 *__restrict__ b_lo, int bkb0, int J, int j0, int lane,
                                           hh_f32x16 (&acc)[NT]) {
    const int h = lane >> 5, i = lane & 31;
    const size_t bstep = (size_t)2 * J;
    const float4 *bph = b_hi + (size_t)(bkb0 * 2 + h) * J + j0 + i;
    const float4 *bpl = b_lo + (size_t)(bkb0 * 2 + h) * J + j0 + i;
    HhpBSet<NT> S[3];
#pragma unroll
    for (int t = 0; t < NT; t++) { S[0].h[t] = bph[t * 32]; S[0].l[t] = bpl[t * 32]; S[1].h[t] = S[0].h[t]; S[1].l[t] = S[0].l[t]; S[2].h[t] = S[0].h[t]; S[2].l[t] = S[0].l[t]; }
    if (KB > 1) {
#pragma unroll
        for (int t = 0; t < NT; t++) { S[1].h[t] = bph[bstep + t * 32]; S[1].l[t] = bpl[bstep + t * 32]; }
    }
    int p0 = kb0 * 2 + h;
    float4 ah = a_hi[p0 * 32 + (i ^ (p0 & 7))], al = a_lo[p0 * 32 + (i ^ (p0 & 7))];
#pragma nounroll
    for (int base = 0; base < KB; base += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int kb = base + u;
            if (kb < KB) { /* wave-uniform */
                HhpBSet<NT> &cur = S[u], &far = S[(u + 2) % 3];
                if (kb + 2 < KB) {
#pragma unroll
                    for (int t = 0; t < NT; t++) { far.h[t] = bph[(size_t)(kb + 2) * bstep + t * 32]; far.l[t] = bpl[(size_t)(kb + 2) * bstep + t * 32]; }
                }
                float4 ahn = ah, aln = al;
                if (kb + 1 < KB) {
                    const int p = (kb0 + kb + 1) * 2 + h;
                    ahn = a_hi[p * 32 + (i ^ (p & 7))];
                    aln = a_lo[p * 32 + (i ^ (p & 7))];
                }
                __builtin_amdgcn_sched_barrier(0);
                const hh_h8 fa_h = hhp_as_h8(ah), fa_l = hhp_as_h8(al);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h, hhp_as_h8(cur.h[t]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h, hhp_as_h8(cur.l[t]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_l, hhp_as_h8(cur.h[t]), acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ah = ahn; al = aln;
            }
        }
    }
}

/* LDS (bytes): Zh 32 KB | Zl 32 KB | Xh 2 KB | Xl 2 KB | L3 partials 16 KB alias Zh | logits 4 KB | rows, norm partials */
#define HHPH_OFF_ZL 32768
#define HHPH_OFF_XH 65536
#define HHPH_OFF_XL 67584
#define HHPH_OFF_LG 69632
#define HHPH_OFF_ROWS 73728
#define HHPH_OFF_NP 73856
#define HHPH_LDS_BYTES (73856 + 512)

__global__ __launch_bounds__(256, 2) void hh_k_policy_h(HhpBank bank, HhpBankH bankh, int n_nets, const float *__restrict__ obs, int obs_stride,
                                                        const int *__restrict__ counts, const int *__restrict__ lists, int max_rows,
                                                        int8_t *__restrict__ actions, float *__restrict__ logits_out) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    _Float16 *Zh = reinterpret_cast<_Float16 *>(ldsb);                  /* [32][2][32][8] hi plane of the activation tile */
    _Float16 *Zl = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_ZL);    /* lo plane */
    _Float16 *Xh = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_XH);    /* [2][2][32][8] observation tile */
    _Float16 *Xl = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_XL);
    float *Pz = reinterpret_cast<float *>(ldsb);                        /* L3 split-K partials [4][32][32] (after S is dead) */
    float *Lg = reinterpret_cast<float *>(ldsb + HHPH_OFF_LG);          /* [32][32] logits */
    int *rows = reinterpret_cast<int *>(ldsb + HHPH_OFF_ROWS);          /* [32] */
    float *npart = reinterpret_cast<float *>(ldsb + HHPH_OFF_NP);       /* [4][32] */

    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? counts[n] : 0;
    int net = -1, tile = blockIdx.x, cnt = 0;
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) {
        const int nt = (cn[n] + HHP_ROWS - 1) / HHP_ROWS;
        if (net < 0) {
            if (tile < nt) { net = n; cnt = cn[n]; }
            else tile -= nt;
        }
    }
    if (net < 0) return;
    const HhpNet N = bank.net[net];
    const HhpNetH H = bankh.net[net];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 31;

    if (tid < HHP_ROWS) {
        const int q = tile * HHP_ROWS + tid;
        rows[tid] = q < cnt ? lists[(size_t)net * max_rows + q] : -1;
    }
    __syncthreads();
    for (int e = tid; e < HHP_ROWS * HHP_XK; e += 256) {
        const int i = e >> 5, c = e & 31, r = rows[i];
        hhp_split_store(Xh, Xl, hhp_haidx(c, i), (r >= 0 && c < N.obs_dim) ? obs[(size_t)r * obs_stride + c] : 0.0f);
    }
    __syncthreads();

    /* ---- L1 ---- */
    {
        hh_f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = hhp_zero16();
        hhp_gemm_h<4>(reinterpret_cast<const float4 *>(Xh), reinterpret_cast<const float4 *>(Xl), 0, HHP_XK / 16, H.w1h, H.w1l, 0, HHP_H, wave * 128, lane, acc);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = wave * 128 + t * 32 + ci;
            const float bj = N.b1[j];
#pragma unroll
            for (int r = 0; r < 16; r++) hhp_split_store(Zh, Zl, hhp_haidx(j, hhp_crow(r, lane)), hhp_tanh(acc[t][r] + bj));
        }
    }
    __syncthreads();

    /* ---- fight nets: x <- normalize(x + Wov x + bov) on columns 400..499 (K = 112: blocks 25..31 of the tile) ---- */
    if (N.has_att) {
        hh_f32x16 acc[1];
        acc[0] = hhp_zero16();
        hhp_gemm_h<1>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), 25, 7, H.wovh, H.wovl, 0, HHP_ATT_J, wave * 32, lane, acc);
        const int j = wave * 32 + ci;
        const float bj = N.bov[j];
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ix = hhp_haidx(400 + (j < 100 ? j : 0), hhp_crow(r, lane));
            const float x = (float)Zh[ix] + (float)Zl[ix];
            y[r] = j < 100 ? x + (acc[0][r] + bj) : 0.0f;
            float s = y[r] * y[r];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            if (ci == 0) npart[wave * 32 + hhp_crow(r, lane)] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = hhp_crow(r, lane);
            const float nn = ((npart[row] + npart[32 + row]) + npart[64 + row]) + npart[96 + row];
            const float den = fmaxf(sqrtf(nn), 1e-12f);
            if (j < 100) hhp_split_store(Zh, Zl, hhp_haidx(400 + j, row), y[r] / den);
        }
        __syncthreads();
    }

    /* ---- L2 ---- */
    {
        hh_f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = hhp_zero16();
        hhp_gemm_h<4>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), 0, HHP_H / 16, H.wsh, H.wsl, 0, HHP_H, wave * 128, lane, acc);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float bj = N.bs[wave * 128 + t * 32 + ci];
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = hhp_tanh(acc[t][r] + bj);
        }
        __syncthreads(); /* Z is dead */
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = wave * 128 + t * 32 + ci;
#pragma unroll
            for (int r = 0; r < 16; r++) hhp_split_store(Zh, Zl, hhp_haidx(j, hhp_crow(r, lane)), acc[t][r]);
        }
    }
    __syncthreads();

    /* ---- L3: split-K, wave w contracts its own columns (blocks 8 w .. 8 w + 7) ---- */
    hh_f32x16 lacc[1];
    lacc[0] = hhp_zero16();
    hhp_gemm_h<1>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), wave * 8, 8, H.wah, H.wal, wave * 8, HHP_OUT, 0, lane, lacc);
    __syncthreads(); /* every wave is done reading S: the hi plane now takes the four partials (16 KB) */
#pragma unroll
    for (int r = 0; r < 16; r++) Pz[wave * 1024 + hhp_crow(r, lane) * 32 + ci] = lacc[0][r];
    __syncthreads();
    for (int e = tid; e < HHP_ROWS * HHP_OUT; e += 256) {
        const int i = e >> 5, c = e & 31;
        const float v = (((Pz[e] + Pz[1024 + e]) + Pz[2048 + e]) + Pz[3072 + e]) + N.ba[c];
        Lg[e] = v;
        if (logits_out && rows[i] >= 0) logits_out[(size_t)rows[i] * HH_POLICY_LOGITS + c] = c < N.n_out ? v : 0.0f;
    }
    __syncthreads();
    if (tid < HHP_ROWS && rows[tid] >= 0) {
        const float *lg = Lg + tid * 32;
        int a[4] = {0, 0, 0, 0};
        const int seg0[5] = {0, 13, 22, 24, 26};
        const int ncomp = N.n_out == 26 ? 4 : 3;
        for (int k = 0; k < ncomp; k++) {
            int best = seg0[k];
            for (int c = seg0[k] + 1; c < seg0[k + 1]; c++) if (lg[c] > lg[best]) best = c;
            a[k] = best - seg0[k];
        }
        reinterpret_cast<int *>(actions)[rows[tid]] = (a[0] & 0xff) | ((a[1] & 0xff) << 8) | ((a[2] & 0xff) << 16) | ((a[3] & 0xff) << 24);
    }
}

/* ---- host side: fp32 -> (hi, lo) fp16, round to nearest even, subnormals kept ---- */
static inline uint16_t hhp_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u)); /* overflow / inf / nan */
    if (x < 0x38800000u) { /* subnormal half (or zero): value = m * 2^-24 */
        const float a = fabsf(f) * 16777216.0f; /* exact scaling by 2^24 */
        uint32_t m = (uint32_t)a;
        const float frac = a - (float)m;
        if (frac > 0.5f || (frac == 0.5f && (m & 1u))) m++;
        return (uint16_t)(sign | m);
    }
    uint32_t mant = x & 0x7fffffu, exp = (x >> 23) - 112u; /* rebias 127 -> 15 */
    uint32_t h = (exp << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++; /* may carry into the exponent: still the right encoding */
    return (uint16_t)(sign | h);
}
static inline float hhp_h2f(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float v;
    if (e == 0) v = (float)m * (1.0f / 16777216.0f);
    else if (e == 31) v = m ? NAN : INFINITY;
    else { uint32_t x = ((e + 112u) << 23) | (m << 13); memcpy(&v, &x, 4); }
    uint32_t b;
    memcpy(&b, &v, 4);
    b |= sign;
    memcpy(&v, &b, 4);
    return v;
}
/* element (k, col) of the fp32 operand M[k][col] (K x J, zero padded) -> the two fragment planes */
static inline void hhp_split_put(std::vector<uint16_t> &hi, std::vector<uint16_t> &lo, size_t off, int k, int col, int J, float v) {
    const uint16_t h = hhp_f2h(v);
    hi[off + hhp_hidx(k, col, J)] = h;
    lo[off + hhp_hidx(k, col, J)] = hhp_f2h(v - hhp_h2f(h));
}

#endif /* HH_POLICY_KERNEL_H16_H */
