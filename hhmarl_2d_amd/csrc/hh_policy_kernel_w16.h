/*
 * hh_policy_kernel_w16.h — weights through LDS, activations in registers, SIXTEEN rows per wave and TWO workgroups per CU.
 *
 * hh_k_policy_w (hh_policy_kernel_w.h) showed what that form is bounded by at one wave per SIMD: a wave's own epilogue (tanh, fp16 split:
 * ~13 issue slots per element) does not hide behind its own MFMAs — the matrix pipe is busy 46 % of a tile's 107 k cycles, the phases
 * that are not the shared layer (observation gather, first layer, attention block, decode: a third of the tile) leave it idle, and nobody
 * else is there to use it.  This form halves a wave's rows so that TWO waves fit a SIMD (256 registers each), and gives them to two
 * INDEPENDENT workgroups (64 rows = four waves each, 69 KB of LDS each): the waves that share a SIMD belong to different tiles in different
 * phases, so one's epilogues, barriers and LDS round trips run under the other's MFMAs without any hand scheduling.
 *   - v_mfma_f32_16x16x32_f16 (weights = A: 16 columns x 32 k, activations = B: 32 k x 16 rows, C^T: lane (row = l & 15, g = l >> 4) holds
 *     columns 4 g .. 4 g + 3).  Two adjacent C^T tiles ARE one B fragment of the next layer (k order hhw16_korder): the 512-wide hidden row
 *     of 16 rows is 16 fragments x (hi, lo) = 128 registers per lane.
 *   - the same linear stream of 1 KB fragments per network as hh_k_policy_w (other packing), copied by LDS-DMA in chunks of <= 32 pieces
 *     into a double buffer one chunk ahead; the output layer's 64 pieces are read straight from global memory (L2 hits; they would make the
 *     last chunk of a column group 40 pieces).
 * Chunks (pieces): L1 tiles 0..15 (32) | L1 tiles 16..31 (32) | ATT tiles 0..3 (32) | ATT tiles 4..6 (24) | 8 column groups x 4 K quarters
 * (32 each: 4 k-blocks x 4 tiles x (hi, lo)) | output layer 64 (not chunked).
 */
#ifndef HH_POLICY_KERNEL_W16_H
#define HH_POLICY_KERNEL_W16_H

typedef float hh_f32x4 __attribute__((ext_vector_type(4)));

#define HHX_CHUNK 32 /* pieces per LDS buffer */
#define HHX_BUF_BYTES (HHX_CHUNK * HHW_PIECE)
#define HHX_L1_PIECES 64
#define HHX_ATT_PIECES 56
#define HHX_L2_PIECES 1024
#define HHX_L3_PIECES 64
#define HHX_STREAM_PIECES (HHX_L1_PIECES + HHX_ATT_PIECES + HHX_L2_PIECES + HHX_L3_PIECES)
#define HHX_OFF_BIAS (2 * HHX_BUF_BYTES)
#define HHX_OFF_ROWS (HHX_OFF_BIAS + (512 + 512 + 128 + 32) * 4)
#define HHX_LDS_BYTES (HHX_OFF_ROWS + 64 * 4)

struct HhpBankX {
    const unsigned char *stream[HH_POLICY_MAX_NETS];
};

/* position (k group, element) of hidden column offset w (0..31) inside a 32-column k-block, in the order two adjacent C^T tiles leave
 * in a lane's registers: tile 0 (columns 0..15) -> elements 0..3 of group (w >> 2), tile 1 (16..31) -> elements 4..7 */
__host__ __device__ inline void hhw16_korder(int w, int &kg, int &e) {
    kg = (w & 15) >> 2;
    e = (w & 3) + ((w >> 4) << 2);
}

#define HHX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)

__device__ __forceinline__ hh_f32x4 hhx_bias_acc(const float *__restrict__ bias /* LDS: the tile's 16 columns */, int g) {
    const float4 b = *reinterpret_cast<const float4 *>(bias + 4 * g);
    return hh_f32x4{b.x, b.y, b.z, b.w};
}
/* pieces [first, first + n) of a chunk, this wave's contiguous quarter of them */
template <int NPW>
__device__ __forceinline__ void hhx_issue(const unsigned char *__restrict__ src, unsigned char *lbuf, int wave, int lane) {
    const unsigned char *s = src + (size_t)wave * NPW * HHW_PIECE + lane * 16;
    unsigned char *d = lbuf + wave * NPW * HHW_PIECE;
#pragma unroll
    for (int u = 0; u < NPW; u++) hhw_glds(s + (size_t)(u >> 2) * 4 * HHW_PIECE, d + (u >> 2) * 4 * HHW_PIECE, u & 3);
}
/* tanh of two adjacent C^T tiles (bias already in the accumulators) -> the (hi, lo) halves of one B fragment */
__device__ __forceinline__ void hhx_pair_to_frag(const hh_f32x4 &a0, const hh_f32x4 &a1, hh_h8 &fh, hh_h8 &fl) {
    const hh_f2 p0 = hhp_tanh2(hh_f2{a0[0], a0[1]}), p1 = hhp_tanh2(hh_f2{a0[2], a0[3]}), p2 = hhp_tanh2(hh_f2{a1[0], a1[1]}), p3 = hhp_tanh2(hh_f2{a1[2], a1[3]});
    const float v[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const _Float16 h = (_Float16)v[i];
        fh[i] = h;
        fl[i] = (_Float16)(v[i] - (float)h);
    }
}

/* one 64-row tile of one network: four waves of 16 rows */
__device__ __forceinline__ void hhx_forward_tile(const HhpNet &N, const unsigned char *__restrict__ st, const float *__restrict__ obs, int obs_stride,
                                                 const int *__restrict__ list, int tile, int cnt, int8_t *__restrict__ actions, float *__restrict__ logits_out,
                                                 unsigned char *ldsb) {
    constexpr int NTH = 256, R = 64;
    unsigned char *buf[2] = {ldsb, ldsb + HHX_BUF_BYTES};
    float *bl = reinterpret_cast<float *>(ldsb + HHX_OFF_BIAS);
    int *rows = reinterpret_cast<int *>(ldsb + HHX_OFF_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 15, g = lane >> 4;

    hhx_issue<8>(st, buf[0], wave, lane); /* chunk 0: L1 tiles 0..15 */
    for (int e = tid; e < 512; e += NTH) { bl[e] = N.b1[e]; bl[512 + e] = N.bs[e]; }
    if (tid < 128) bl[1024 + tid] = N.has_att ? N.bov[tid] : 0.0f;
    if (tid < 32) bl[1152 + tid] = N.ba[tid];
    const int q_ = tile * R + wave * 16 + ci;
    const int row = q_ < cnt ? list[q_] : -1;
    if (g == 0) rows[wave * 16 + ci] = row;
    hh_h8 xh, xl; /* the observation: lane (row, g) holds columns 8 g .. 8 g + 7 */
    {
        const int od = N.obs_dim;
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int c = 8 * g + e;
            const bool ok = row >= 0 && c < od;
            const float x = obs[ok ? (size_t)row * obs_stride + c : 0];
            xv[e] = ok ? x : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const _Float16 h = (_Float16)xv[e];
            xh[e] = h;
            xl[e] = (_Float16)(xv[e] - (float)h);
        }
    }
    const unsigned char *sp = st + (size_t)HHX_CHUNK * HHW_PIECE; /* the next chunk to request */
    hh_h8 zh[16], zl[16]; /* the hidden row: fragment kb = columns 32 kb .. 32 kb + 31 in hhw16_korder */

    /* ---- L1: 32 column tiles of 16, K = one block of 32 observation columns; two chunks of 16 tiles ---- */
#pragma unroll
    for (int c = 0; c < 2; c++) {
        __syncthreads(); /* chunk c landed; the other buffer is free */
        if (c == 0) hhx_issue<8>(sp, buf[1], wave, lane);
        else if (N.has_att) hhx_issue<8>(sp, buf[0], wave, lane);                                    /* ATT tiles 0..3 */
        else { sp += (size_t)HHX_ATT_PIECES * HHW_PIECE; hhx_issue<8>(sp, buf[0], wave, lane); }   /* escape nets: straight to shared layer (0, 0) */
        sp += (size_t)HHX_CHUNK * HHW_PIECE;
#pragma unroll
        for (int tp = 0; tp < 8; tp++) { /* pairs of tiles = one fragment of the hidden row */
            hh_f32x4 a[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int T = c * 16 + tp * 2 + u;
                a[u] = hhx_bias_acc(bl + 16 * T, g);
                const hh_h8 wh = hhw_frag(buf[c], (tp * 2 + u) * 2, lane), wl = hhw_frag(buf[c], (tp * 2 + u) * 2 + 1, lane);
                HHX_MFMA(wh, xh, a[u]);
                HHX_MFMA(wl, xh, a[u]);
                HHX_MFMA(wh, xl, a[u]);
            }
            hhx_pair_to_frag(a[0], a[1], zh[c * 8 + tp], zl[c * 8 + tp]);
        }
    }

    /* ---- fight nets: x <- normalize(x + Wov x + bov) on hidden columns 400..499.  K = fragments 12..15 (columns 384..511, the weights of
     *      384..399 are zero); output tile j (columns 400 + 16 j ..) is half (25 + j) & 1 of fragment (25 + j) >> 1 ---- */
    if (N.has_att) {
        hh_f32x4 y[7];
        float ssum = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            __syncthreads();
            if (c == 0) { hhx_issue<6>(sp, buf[1], wave, lane); sp += (size_t)(HHX_ATT_PIECES - HHX_CHUNK) * HHW_PIECE; }   /* ATT tiles 4..6 (24 pieces) */
            else { hhx_issue<8>(sp, buf[0], wave, lane); sp += (size_t)HHX_CHUNK * HHW_PIECE; }                             /* shared layer (0, 0) */
#pragma unroll
            for (int jj = 0; jj < (c == 0 ? 4 : 3); jj++) {
                const int j = c * 4 + jj;
                hh_f32x4 acc = hhx_bias_acc(bl + 1024 + 16 * j, g);
#pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    const hh_h8 wh = hhw_frag(buf[c], (jj * 4 + kb) * 2, lane), wl = hhw_frag(buf[c], (jj * 4 + kb) * 2 + 1, lane);
                    HHX_MFMA(wh, zh[12 + kb], acc);
                    HHX_MFMA(wl, zh[12 + kb], acc);
                    HHX_MFMA(wh, zl[12 + kb], acc);
                }
                const int f = (25 + j) >> 1, e0 = 4 * ((25 + j) & 1);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool ok = 16 * j + 4 * g + r < 100;
                    const float x = (float)zh[f][e0 + r] + (float)zl[f][e0 + r];
                    const float v = ok ? x + acc[r] : 0.0f;
                    y[j][r] = v;
                    ssum += v * v;
                }
            }
        }
        ssum += __shfl_xor(ssum, 16);
        ssum += __shfl_xor(ssum, 32); /* the four k groups of a row: (s0 + s1) + (s2 + s3) on every lane */
        const float inv = 1.0f / fmaxf(sqrtf(ssum), 1e-12f);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int f = (25 + j) >> 1, e0 = 4 * ((25 + j) & 1);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = y[j][r] * inv; /* columns >= 500 were zeroed above */
                const _Float16 h = (_Float16)v;
                zh[f][e0 + r] = h;
                zl[f][e0 + r] = (_Float16)(v - (float)h);
            }
        }
    }

    /* ---- L2 (shared layer): 8 groups of four column tiles x 4 K quarters; the output layer from the group's registers ---- */
    const unsigned char *l3 = st + (size_t)(HHX_L1_PIECES + HHX_ATT_PIECES + HHX_L2_PIECES) * HHW_PIECE + lane * 16;
    hh_f32x4 lacc[2] = {hh_f32x4{0.0f, 0.0f, 0.0f, 0.0f}, hh_f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma nounroll
    for (int p = 0; p < 8; p++) {
        hh_f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = hhx_bias_acc(bl + 512 + 64 * p + 16 * t, g);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            __syncthreads(); /* chunk (p, q) landed in buf[q & 1]; the other buffer is free */
            const bool more = p < 7 || q < 3;
            const unsigned char *gsrc = sp + (size_t)wave * 8 * HHW_PIECE + lane * 16; /* this wave's eight pieces of the next chunk: two per step, behind the */
            unsigned char *gdst = buf[(q + 1) & 1] + wave * 8 * HHW_PIECE;              /* MFMAs of the first four steps (an LDS-DMA request costs ~60 cycles to issue) */
            sp += (size_t)HHX_CHUNK * HHW_PIECE;
            { /* eight steps of (k-block kk, tile pair tp) = 4 fragments, 6 MFMAs; the fragments of step s + 1 are requested before the MFMAs of step s */
                const unsigned char *cb = buf[q & 1];
                hh_h8 an[4];
#pragma unroll
                for (int u = 0; u < 4; u++) an[u] = hhw_frag(cb, u, lane);
#pragma unroll
                for (int s_ = 0; s_ < 8; s_++) {
                    const int kk = s_ >> 1, tp = s_ & 1;
                    hh_h8 a[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) a[u] = an[u];
                    hhw_need4(a);
                    if (s_ + 1 < 8) {
#ifdef HHX_ABL_HALF_LDS /* tuning builds: half of the fragment reads (wrong results): is the LDS read path what two co-resident tiles fight over? */
                        an[0] = hhw_frag(cb, (s_ + 1) * 4, lane); an[1] = hhw_frag(cb, (s_ + 1) * 4 + 1, lane); an[2] = an[0]; an[3] = an[1];
#else
#pragma unroll
                        for (int u = 0; u < 4; u++) an[u] = hhw_frag(cb, (s_ + 1) * 4 + u, lane);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    HHX_MFMA(a[0], zh[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[2], zh[4 * q + kk], acc[2 * tp + 1]);
#ifndef HHX_ABL_THIRD_MFMA /* tuning builds: the hi x hi products only (wrong results): is the matrix pipe what two co-resident tiles fight over? */
                    HHX_MFMA(a[1], zh[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[3], zh[4 * q + kk], acc[2 * tp + 1]);
                    HHX_MFMA(a[0], zl[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[2], zl[4 * q + kk], acc[2 * tp + 1]);
#endif
#ifndef HHX_ABL_NO_GLDS /* tuning builds: the shared layer's chunks are never copied (wrong results): what does the LDS-DMA stream cost? */
                    if (s_ < 4 && more) hhw_issue_some(gsrc, gdst, 2 * s_, 2);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        /* tanh of the four tiles = two B fragments of the output layer (S columns 64 p .. 64 p + 63 = k-blocks 2 p, 2 p + 1) */
#pragma unroll
        for (int u = 0; u < 2; u++) {
            hh_h8 sh, sl;
            hhx_pair_to_frag(acc[2 * u], acc[2 * u + 1], sh, sl);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float4 *w = reinterpret_cast<const float4 *>(l3 + (size_t)(((2 * p + u) * 2 + t) * 2) * HHW_PIECE);
                const hh_h8 wh = hhp_as_h8(w[0]), wl = hhp_as_h8(w[HHW_PIECE / 16]);
                HHX_MFMA(wh, sh, lacc[t]);
                HHX_MFMA(wl, sh, lacc[t]);
                HHX_MFMA(wh, sl, lacc[t]);
            }
        }
    }

    /* ---- logits: lane (row, g) holds output columns 16 t + 4 g + (0..3); they meet in LDS for the decode ---- */
    __syncthreads(); /* every wave is done with the chunk buffers */
    float *Lg = reinterpret_cast<float *>(ldsb); /* [64][32] */
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float4 b = *reinterpret_cast<const float4 *>(bl + 1152 + 16 * t + 4 * g);
        *reinterpret_cast<float4 *>(Lg + (wave * 16 + ci) * 32 + 16 * t + 4 * g) = make_float4(lacc[t][0] + b.x, lacc[t][1] + b.y, lacc[t][2] + b.z, lacc[t][3] + b.w);
    }
    __syncthreads();
    if (logits_out)
        for (int e = tid; e < R * 32; e += NTH) {
            const int i = e >> 5, c = e & 31;
            if (rows[i] >= 0) logits_out[(size_t)rows[i] * HH_POLICY_LOGITS + c] = c < N.n_out ? Lg[e] : 0.0f;
        }
    { /* greedy decode (env_base.py:373-382), one thread per (row, MultiDiscrete component): first maximum of its segment */
        const int i = tid >> 2, k = tid & 3;
        const int lo = k == 0 ? 0 : (k == 1 ? 13 : (k == 2 ? 22 : 24)), hi = k == 0 ? 13 : (k == 1 ? 22 : (k == 2 ? 24 : 26));
        const float *lg = Lg + i * 32;
        int best = lo;
        if (k < (N.n_out == 26 ? 4 : 3))
            for (int c = lo + 1; c < hi; c++) if (lg[c] > lg[best]) best = c;
        int a = (best - lo) << (8 * k);
        a |= __builtin_amdgcn_mov_dpp(a, 0xB1, 0xf, 0xf, true);
        a |= __builtin_amdgcn_mov_dpp(a, 0x4E, 0xf, 0xf, true);
        if (k == 0 && rows[i] >= 0) reinterpret_cast<int *>(actions)[rows[i]] = a;
    }
}

__global__ __launch_bounds__(256, 2) void hh_k_policy_w16(HhpBank bank, HhpBankX bankx, int n_nets, const float *__restrict__ obs, int obs_stride, int *counts,
                                                          const int *__restrict__ lists, int max_rows, int8_t *__restrict__ actions, float *__restrict__ logits_out,
                                                          int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    int net, tile, cnt;
#ifdef HHX_STAGGER /* tuning builds: delay one of the two workgroups that share a CU (1: odd ids, 2: the second half of the grid) by HHX_STAGGER_SLEEPS x 8 k cycles */
    if ((HHX_STAGGER == 1 && (blockIdx.x & 1)) || (HHX_STAGGER == 2 && blockIdx.x >= gridDim.x / 2))
        for (int i = 0; i < HHX_STAGGER_SLEEPS; i++) __builtin_amdgcn_s_sleep(127);
#endif
    if (hhp_locate<64>(cn, (int)blockIdx.x, net, tile, cnt))
        hhx_forward_tile(bank.net[net], bankx.stream[net], obs, obs_stride, lists + (size_t)net * max_rows, tile, cnt, actions, logits_out, ldsb);
    hhp_consume_counts(counts, consume);
}

/* host: element (k, col) of a [K x J] operand into piece `piece_hi` (16 columns x 32 k; the lo plane is the next piece): natural k order
 * (nat: k group = (k & 31) >> 3, e = k & 7) or hhw16_korder */
static inline void hhx_put(std::vector<uint16_t> &S, size_t piece_hi, int k, int col, bool nat, float v) {
    int kg, e;
    if (nat) { kg = (k & 31) >> 3; e = k & 7; }
    else hhw16_korder(k & 31, kg, e);
    const size_t at = (size_t)((kg * 16 + (col & 15)) * 8 + e);
    const uint16_t h = hhp_f2h(v);
    S[piece_hi * (HHW_PIECE / 2) + at] = h;
    S[(piece_hi + 1) * (HHW_PIECE / 2) + at] = hhp_f2h(v - hhp_h2f(h));
}

#endif /* HH_POLICY_KERNEL_W16_H */
