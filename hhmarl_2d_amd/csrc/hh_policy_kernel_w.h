/*
 * hh_policy_kernel_w.h — the policy forward with the WEIGHTS streamed through LDS and the ACTIVATIONS resident in registers.
 *
 * hh_k_policy_h (hh_policy_kernel_h16.h) keeps a 32-row activation tile in LDS and every 32-row tile streams the network's 1.2 MB
 * of weight fragments out of the L2 through the CU's 64 B/clk vector-memory path; its phases (weight round trips, epilogues, barriers)
 * leave the matrix pipe idle two thirds of the time.  This form turns the tile inside out:
 *   - a WAVE owns 32 rows from the observation to the logits.  The transposed contraction (weights = MFMA A operand, activations = B
 *     operand) leaves a C^T tile in exactly the register layout of two B fragments of the next layer (hhp_hidx_t), so the 512-wide
 *     hidden row of every layer lives in 256 registers per lane (hi and lo fp16 planes) and never touches LDS: no activation tile,
 *     no cross-wave exchange, no barrier inside a layer, the attention block's row norm is an in-lane sum + one v_permlane32_swap;
 *   - the four waves of a workgroup (128 rows, one wave per SIMD, 512 registers) share ONE pass over the weights: the host packs every
 *     network as a linear stream of 1 KB fragments in consumption order, the waves copy it chunk by chunk into a double-buffered LDS
 *     ring with global_load_lds_dwordx4 (LDS-DMA: 1 KB per wave-instruction, no staging registers) one chunk ahead of the MFMAs that
 *     read it, and every wave reads every fragment with one conflict-free ds_read_b128.  L2 -> CU traffic per row drops 4x, the LDS
 *     read rate (4 waves x 1 KB per 3 MFMAs) stays under the LDS unit's 128 B/clk.
 * Same arithmetic as hh_k_policy_h: split-fp16 operands, hi hi + lo hi + hi lo accumulated in fp32 by v_mfma_f32_32x32x16_f16 in the
 * same k order, the same epilogues (hhp_tanh2, fp16 split), the same eight 64-column partial ranges are NOT needed any more — a wave
 * accumulates its rows' logits over all 512 hidden columns in ONE accumulator, in k order (the logits differ from hh_k_policy_h's in
 * the last bits: both are pinned to the reference's fp32 forward within 1e-5, tests/test_policy_nets.py).
 *
 * Stream of one network (pieces of 1 KB = one A fragment: lane l holds 8 halves A[col 32 t + (l & 31)][k group l >> 5]):
 *   chunk L1   : for T in 0..15, kb in 0..1      : hi, lo          64 pieces   (K = 32 observation columns, natural k order)
 *   chunk ATT  : for j in 0..3,  kb in 0..6      : hi, lo          56 pieces   (fight nets only; K = Z columns 400..511 in hidx_t order)
 *   chunk (p,h): for kk in 0..15, t in 0..1      : hi, lo          64 pieces   p = 0..7 column-tile pairs of the shared layer, h = K half
 *               + behind (p, 0) for p > 0: the output layer's 4 k-blocks of S columns 64 (p - 1) .. 64 p - 1 (pair p - 1's epilogue runs
 *                 in the shadow of pair p's first half), behind (7, 1): those of pair 7                : hi, lo    8 pieces
 */
#ifndef HH_POLICY_KERNEL_W_H
#define HH_POLICY_KERNEL_W_H

#include <type_traits>

#define HHW_PIECE 1024
#define HHW_L1_PIECES 64
#define HHW_ATT_PIECES 56
#define HHW_L2_PIECES 64
#define HHW_L3_PIECES 8
#define HHW_BUF_BYTES ((HHW_L2_PIECES + HHW_L3_PIECES) * HHW_PIECE) /* 72 KB */
#define HHW_STREAM_PIECES (HHW_L1_PIECES + HHW_ATT_PIECES + 8 * (2 * HHW_L2_PIECES + HHW_L3_PIECES))
/* LDS: two chunk buffers | biases b1 512, bs 512, bov 128, ba 32 floats | row ids 128 */
#define HHW_OFF_BIAS (2 * HHW_BUF_BYTES)
#define HHW_OFF_ROWS (HHW_OFF_BIAS + (512 + 512 + 128 + 32) * 4)
#define HHW_LDS_BYTES (HHW_OFF_ROWS + 128 * 4)

struct HhpNetW {
    const unsigned char *stream; /* HHW_STREAM_PIECES pieces (the ATT chunk is present but unused for escape nets) */
};
struct HhpBankW {
    HhpNetW net[HH_POLICY_MAX_NETS];
};

/* first piece of shared-layer chunk (p, h) relative to chunk (0, 0): (0, 0) has 64 pieces, (p, 0) for p > 0 has 64 + the 8 output-layer
 * pieces of pair p - 1, (p, 1) has 64 (+ the 8 output-layer pieces of pair 7 for p = 7) */
__host__ __device__ inline int hhw_chunk_piece(int p, int h) {
    return h ? p * (2 * HHW_L2_PIECES + HHW_L3_PIECES) + HHW_L2_PIECES : (p ? p * (2 * HHW_L2_PIECES + HHW_L3_PIECES) - HHW_L3_PIECES : 0);
}
typedef __attribute__((address_space(3))) unsigned char hhw_lds_u8;
typedef const __attribute__((address_space(1))) unsigned char hhw_glb_u8;

/* this wave's share of a chunk: pieces wave, wave + WV, ... -> LDS (LDS-DMA; completion is counted by vmcnt: the compiler waits
 * vmcnt(0) in front of the next __syncthreads()) */
/* one 1 KB piece: global (per lane) -> LDS (wave-uniform base), both addresses + o x 1 KB through the instruction's immediate offset */
__device__ __forceinline__ void hhw_glds(const unsigned char *s, unsigned char *d, int o) {
    switch (o) { /* the builtin wants a literal; o is a constant after unrolling */
    case 0: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 0, 0); break;
    case 1: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, HHW_PIECE, 0); break;
    case 2: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 2 * HHW_PIECE, 0); break;
    default: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 3 * HHW_PIECE, 0); break;
    }
}
template <int WV, int NP>
__device__ __forceinline__ void hhw_issue(const unsigned char *__restrict__ src, unsigned char *lbuf, int wave, int lane) {
    static_assert(NP % WV == 0, "a chunk is a whole number of pieces per wave");
    constexpr int NPW = NP / WV; /* wave w copies the CONTIGUOUS pieces w NPW .. (w + 1) NPW - 1: four pieces share one address / M0 setup through the instruction's immediate offset */
    const unsigned char *s = src + (size_t)wave * NPW * HHW_PIECE + lane * 16;
    unsigned char *d = lbuf + wave * NPW * HHW_PIECE;
#pragma unroll
    for (int u = 0; u < NPW; u++) /* straight-line: a rolled loop costs a taken branch (29 cycles) per 1 KB piece */
        hhw_glds(s + (size_t)(u >> 2) * 4 * HHW_PIECE, d + (u >> 2) * 4 * HHW_PIECE, u & 3);
}

__device__ __forceinline__ hh_h8 hhw_frag(const unsigned char *lbuf, int piece, int lane) {
    return hhp_as_h8(*reinterpret_cast<const float4 *>(lbuf + piece * HHW_PIECE + lane * 16));
}

/* acc(16) + bias -> f -> (hi, lo) halves: the C^T tile as the two B fragments of the next layer (k order hhp_hidx_t) */
template <class F>
__device__ __forceinline__ void hhw_tile_to_frags(const hh_f32x16 &acc, const float *__restrict__ bias /* LDS, this tile's 32 columns */, int g,
                                                  hh_h8 (&fh)[2], hh_h8 (&fl)[2], F f) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + 8 * q + 4 * g);
        const hh_f2 p0 = f(hh_f2{acc[4 * q + 0], acc[4 * q + 1]} + hh_f2{b.x, b.y}), p1 = f(hh_f2{acc[4 * q + 2], acc[4 * q + 3]} + hh_f2{b.z, b.w});
        const float v[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const _Float16 h = (_Float16)v[i];
            fh[q >> 1][4 * (q & 1) + i] = h;
            fl[q >> 1][4 * (q & 1) + i] = (_Float16)(v[i] - (float)h);
        }
    }
}

/* tuning builds (-DHHP_PROFILE): per-phase cycles of wave 0, summed in registers and flushed ONCE per tile — an atomic per phase from
 * 256 tiles that move in lockstep serialises on the counter's address and the barriers then wait for it */
#ifdef HHP_PROFILE
#define HHW_T0 unsigned long long pacc_[16] = {0}, pt_ = __builtin_readcyclecounter()
#define HHW_T(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); pacc_[k] += t_ - pt_; pt_ = t_; } while (0)
#define HHW_TFLUSH do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 16; k_++) atomicAdd(&hhp_prof[k_], pacc_[k_]); } while (0)
#else
#define HHW_T0
#define HHW_T(k)
#define HHW_TFLUSH
#endif
/* a C^T accumulator that starts at its columns' biases (LDS, the tile's 32 columns): the epilogue then has no LDS operand of its own —
 * a bias read in the shadow of the MFMAs made hipcc wait for every fragment read in flight (lgkmcnt(0)) once per k-block */
__device__ __forceinline__ hh_f32x16 hhw_bias_acc(const float *__restrict__ bias, int g) {
    hh_f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + 8 * q + 4 * g);
        a[4 * q] = b.x; a[4 * q + 1] = b.y; a[4 * q + 2] = b.z; a[4 * q + 3] = b.w;
    }
    return a;
}
/* "these fragments are needed HERE": hipcc then waits for them (s_waitcnt lgkmcnt(0): it does not count LDS returns beside LDS-DMA)
 * BEFORE the next step's reads are issued instead of after — the wait covers reads that are a whole step old, not the ones just issued */
__device__ __forceinline__ void hhw_need4(hh_h8 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void hhw_need2(hh_h8 &a, hh_h8 &b) { asm volatile("" : "+v"(a), "+v"(b)); }
#define HHW_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)

/* one K half of two column tiles of the shared layer: 16 k-blocks x (2 tiles x 3 MFMAs), the four fragments of k-block kk + 1 requested
 * before the MFMAs of k-block kk issue.  fill(kk) is independent work (this wave's LDS-DMA requests for the next chunk, a slice of the
 * previous tile pair's epilogue) placed in the same scheduling region as step kk's MFMAs: with one wave per SIMD nobody else fills the
 * matrix pipe's shadow (~5 issue slots per MFMA) */
template <int VALU_PER_MFMA, class P, class F>
__device__ __forceinline__ void hhw_l2_half(const unsigned char *buf, int lane, const hh_h8 (&zh)[32], const hh_h8 (&zl)[32], int kb0, hh_f32x16 &acc0,
                                            hh_f32x16 &acc1, P pre, F fill) {
    hh_h8 an[4];
#pragma unroll
    for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf, u, lane);
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        hh_h8 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = an[u];
        hhw_need4(a);
        const auto pv = pre(kk);
        if (kk + 1 < 16) {
#pragma unroll
            for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf, (kk + 1) * 4 + u, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        HHW_MFMA(a[0], zh[kb0 + kk], acc0); HHW_MFMA(a[2], zh[kb0 + kk], acc1);
        HHW_MFMA(a[1], zh[kb0 + kk], acc0); HHW_MFMA(a[3], zh[kb0 + kk], acc1);
        HHW_MFMA(a[0], zl[kb0 + kk], acc0); HHW_MFMA(a[2], zl[kb0 + kk], acc1);
        fill(kk, pv);
        if constexpr (VALU_PER_MFMA > 0) { /* an MFMA, then its shadow's worth of the filler's vector instructions, six times */
#pragma unroll
            for (int i = 0; i < 6; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

/* The first K half of a tile pair WITH the previous pair's epilogue in the MFMAs' shadow, scheduled by hand.  An in-order wave that meets
 * a vector instruction whose operand is not ready yet stalls — and with it the next MFMA's issue: a tanh chain (mul, exp, add, rcp, fma,
 * cvt, cvt, sub, cvt) issued back to back between two MFMAs costs its whole dependent latency (measured: +19 cycles per MFMA, nothing
 * hidden).  So FOUR values travel together, one operation of each in turn, three or four instructions per MFMA, every instruction's
 * operand at least three issue slots and one MFMA old; scheduling barriers pin the order.  Step pair kk2 (k-blocks 2 kk2, 2 kk2 + 1, twelve
 * MFMAs) carries slice kk2 = accumulator values 4 q .. 4 q + 3 of tile kk2 >> 2, q = kk2 & 3. */
template <class G>
__device__ __forceinline__ void hhw_l2_half_epi(const unsigned char *buf, int lane, const hh_h8 (&zh)[32], const hh_h8 (&zl)[32], hh_f32x16 &acc0, hh_f32x16 &acc1,
                                                const hh_f32x16 &p0, const hh_f32x16 &p1, hh_h8 (&sh)[4], hh_h8 (&sl)[4], G glds) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    constexpr float K2 = 2.885390081777926815f; /* 2 log2(e) */
    hh_h8 an[4];
#pragma unroll
    for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf, u, lane);
#define HHW_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int kk2 = 0; kk2 < 8; kk2++) {
        const int t = kk2 >> 2, q = kk2 & 3;
        const hh_f32x16 &pa = t ? p1 : p0;
        const float xa = pa[4 * q], xb = pa[4 * q + 1], xc = pa[4 * q + 2], xd = pa[4 * q + 3];
        float ta, tb, tc, td, ea, eb, ec, ed, va, vb, vc, vd;
        h2v hab, hcd;
        hh_h8 a[4];
        /* ---- k-block 2 kk2 ---- */
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = an[u];
        hhw_need4(a);
#pragma unroll
        for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf, (2 * kk2 + 1) * 4 + u, lane);
        glds(2 * kk2);
        HHW_SB;
        HHW_MFMA(a[0], zh[2 * kk2], acc0); ta = xa * K2; tb = xb * K2; tc = xc * K2; HHW_SB;
        HHW_MFMA(a[2], zh[2 * kk2], acc1); td = xd * K2; ea = __builtin_amdgcn_exp2f(ta); eb = __builtin_amdgcn_exp2f(tb); HHW_SB;
        HHW_MFMA(a[1], zh[2 * kk2], acc0); ec = __builtin_amdgcn_exp2f(tc); ed = __builtin_amdgcn_exp2f(td); ea += 1.0f; HHW_SB;
        HHW_MFMA(a[3], zh[2 * kk2], acc1); eb += 1.0f; ec += 1.0f; ed += 1.0f; HHW_SB;
        HHW_MFMA(a[0], zl[2 * kk2], acc0); ea = __builtin_amdgcn_rcpf(ea); eb = __builtin_amdgcn_rcpf(eb); ec = __builtin_amdgcn_rcpf(ec); HHW_SB;
        HHW_MFMA(a[2], zl[2 * kk2], acc1); ed = __builtin_amdgcn_rcpf(ed); va = __builtin_fmaf(ea, -2.0f, 1.0f); vb = __builtin_fmaf(eb, -2.0f, 1.0f); HHW_SB;
        /* ---- k-block 2 kk2 + 1 ---- */
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = an[u];
        hhw_need4(a);
        if (kk2 < 7) {
#pragma unroll
            for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf, (2 * kk2 + 2) * 4 + u, lane);
        }
        glds(2 * kk2 + 1);
        HHW_SB;
        HHW_MFMA(a[0], zh[2 * kk2 + 1], acc0); vc = __builtin_fmaf(ec, -2.0f, 1.0f); vd = __builtin_fmaf(ed, -2.0f, 1.0f);
        hab = h2v{(_Float16)va, (_Float16)vb}; HHW_SB;
        HHW_MFMA(a[2], zh[2 * kk2 + 1], acc1); hcd = h2v{(_Float16)vc, (_Float16)vd};
        const float fa = (float)hab.x, fb = (float)hab.y; HHW_SB;
        HHW_MFMA(a[1], zh[2 * kk2 + 1], acc0); const float fc = (float)hcd.x, fd = (float)hcd.y, da = va - fa; HHW_SB;
        HHW_MFMA(a[3], zh[2 * kk2 + 1], acc1); const float db = vb - fb, dc = vc - fc, dd = vd - fd; HHW_SB;
        HHW_MFMA(a[0], zl[2 * kk2 + 1], acc0); const h2v lab = h2v{(_Float16)da, (_Float16)db}, lcd = h2v{(_Float16)dc, (_Float16)dd}; HHW_SB;
        HHW_MFMA(a[2], zl[2 * kk2 + 1], acc1); HHW_SB;
        const int f = t * 2 + (q >> 1), e = 4 * (q & 1);
        sh[f][e] = hab.x; sh[f][e + 1] = hab.y; sh[f][e + 2] = hcd.x; sh[f][e + 3] = hcd.y;
        sl[f][e] = lab.x; sl[f][e + 1] = lab.y; sl[f][e + 2] = lcd.x; sl[f][e + 3] = lcd.y;
    }
#undef HHW_SB
}

/* pieces first .. first + n - 1 of this wave's (contiguous) share of a chunk; s / d = the lane's source byte / the LDS address of the share's first piece */
__device__ __forceinline__ void hhw_issue_some(const unsigned char *__restrict__ s, unsigned char *d, int first, int n) {
#pragma unroll
    for (int u = 0; u < n; u++) {
        const int i = first + u;
        hhw_glds(s + (size_t)(i >> 2) * 4 * HHW_PIECE, d + (i >> 2) * 4 * HHW_PIECE, i & 3);
    }
}

/* slice kk (0..15) of a tile pair's epilogue: two accumulator values of tile kk >> 3 (bias included: hhw_bias_acc) -> tanh -> (hi, lo)
 * halves of the output layer's B fragments */
__device__ __forceinline__ void hhw_epi_slice(int kk, const hh_f32x16 &p0, const hh_f32x16 &p1, hh_h8 (&sh)[4], hh_h8 (&sl)[4]) {
    const int t = kk >> 3, q = (kk >> 1) & 3, hf = kk & 1;
    const hh_f32x16 &acc = t ? p1 : p0;
    /* scalar on purpose: beside MFMAs a packed f32 instruction (v_pk_add / mul / fma_f32) costs ~13 cycles beyond its issue slot, a
     * transcendental ~2 (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the library is built with -fno-slp-vectorize so that
     * hipcc does not pair these up again */
#ifdef HHW_ABL_NO_TANH /* tuning builds: what the transcendentals cost in the MFMAs' shadow */
    const float v0 = acc[4 * q + 2 * hf] * 0.5f, v1 = acc[4 * q + 2 * hf + 1] * 0.5f;
#else
    const float v0 = hhp_tanh(acc[4 * q + 2 * hf]), v1 = hhp_tanh(acc[4 * q + 2 * hf + 1]);
#endif
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    const int f = t * 2 + (q >> 1), e = 4 * (q & 1) + 2 * hf;
    sh[f][e] = h0; sh[f][e + 1] = h1;
    sl[f][e] = (_Float16)(v0 - (float)h0); sl[f][e + 1] = (_Float16)(v1 - (float)h1);
}
/* the output layer's share of one tile pair: four k-blocks from the fragments hhw_epi_slice built, weights = pieces first .. first + 7 of buf */
__device__ __forceinline__ void hhw_l3_pair(const unsigned char *buf, int first, int lane, const hh_h8 (&sh)[4], const hh_h8 (&sl)[4], hh_f32x16 &lacc) {
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const hh_h8 wh = hhw_frag(buf, first + f * 2, lane), wl = hhw_frag(buf, first + f * 2 + 1, lane);
        HHW_MFMA(wh, sh[f], lacc);
        HHW_MFMA(wl, sh[f], lacc);
        HHW_MFMA(wh, sl[f], lacc);
    }
}

/* one tile of 32 WV rows of one network */
template <int WV>
__device__ __forceinline__ void hhw_forward_tile(const HhpNet &N, const HhpNetW &W, const float *__restrict__ obs, int obs_stride, const int *__restrict__ list,
                                                 int tile, int cnt, int8_t *__restrict__ actions, float *__restrict__ logits_out, unsigned char *ldsb) {
    constexpr int NTH = 64 * WV, R = 32 * WV;
    unsigned char *buf0 = ldsb, *buf1 = ldsb + HHW_BUF_BYTES;
    float *bl = reinterpret_cast<float *>(ldsb + HHW_OFF_BIAS);
    int *rows = reinterpret_cast<int *>(ldsb + HHW_OFF_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 31, g = lane >> 5;
    const unsigned char *st = W.stream;

    HHW_T0;
    hhw_issue<WV, HHW_L1_PIECES>(st, buf0, wave, lane); /* L1's weights are on their way while the rows are looked up */
    for (int e = tid; e < 512; e += NTH) { bl[e] = N.b1[e]; bl[512 + e] = N.bs[e]; }
    for (int e = tid; e < 128; e += NTH) bl[1024 + e] = N.has_att ? N.bov[e] : 0.0f;
    for (int e = tid; e < 32; e += NTH) bl[1152 + e] = N.ba[e];
    const int q_ = tile * R + wave * 32 + ci;
    const int row = q_ < cnt ? list[q_] : -1;
    if (g == 0) rows[wave * 32 + ci] = row;
    /* the observation as the two B fragments of L1: lane (row, g) holds columns 8 g .. 8 g + 7 and 16 + 8 g .. 16 + 8 g + 7 */
    hh_h8 xh[2], xl[2];
    {
        const int od = N.obs_dim;
        float xv[16];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = 16 * b + 8 * g + e;
                const bool ok = row >= 0 && c < od;
                const float x = obs[ok ? (size_t)row * obs_stride + c : 0];
                xv[8 * b + e] = ok ? x : 0.0f;
            }
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const _Float16 h = (_Float16)xv[8 * b + e];
                xh[b][e] = h;
                xl[b][e] = (_Float16)(xv[8 * b + e] - (float)h);
            }
    }
    __syncthreads(); /* chunk L1 landed (every wave waited for its own pieces), biases staged */
    HHW_T(0);

    /* ---- L1: 16 column tiles of the hidden row, each the sum over the two observation k-blocks ---- */
    const unsigned char *sp = st + (size_t)HHW_L1_PIECES * HHW_PIECE; /* next chunk in the stream */
    if (!N.has_att) sp += (size_t)HHW_ATT_PIECES * HHW_PIECE;           /* escape nets: straight to chunk (0, 0) */
    if (N.has_att) hhw_issue<WV, HHW_ATT_PIECES>(sp, buf1, wave, lane);
    else hhw_issue<WV, HHW_L2_PIECES>(sp, buf1, wave, lane);
    hh_h8 zh[32], zl[32]; /* the hidden row: fragment kb = columns 16 kb .. 16 kb + 15 in hidx_t order */
    /* LDS reads run one step ahead of the MFMAs that consume them (one wave per SIMD: nobody else hides an LDS round trip); the
     * scheduling barriers keep hipcc from sinking each read to its use */
    {
        hh_h8 an[4];
#pragma unroll
        for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf0, u, lane);
#pragma unroll
        for (int T = 0; T < 16; T++) {
            hh_h8 a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = an[u];
            hhw_need4(a);
            if (T + 1 < 16) {
#pragma unroll
                for (int u = 0; u < 4; u++) an[u] = hhw_frag(buf0, (T + 1) * 4 + u, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            hh_f32x16 acc = hhp_zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
                HHW_MFMA(a[2 * kb], xh[kb], acc);
                HHW_MFMA(a[2 * kb + 1], xh[kb], acc);
                HHW_MFMA(a[2 * kb], xl[kb], acc);
            }
            hh_h8 fh[2], fl[2];
            hhw_tile_to_frags(acc, bl + 32 * T, g, fh, fl, [](hh_f2 v) { return hhp_tanh2(v); });
            zh[2 * T] = fh[0]; zh[2 * T + 1] = fh[1]; zl[2 * T] = fl[0]; zl[2 * T + 1] = fl[1];
        }
    }
    HHW_T(1);
    __syncthreads(); /* next chunk landed; buf0 is free */
    HHW_T(2);

    unsigned char *bufA = buf1, *bufB = buf0; /* (p, 0) is read from bufA, (p, 1) from bufB */
    if (N.has_att) {
        /* ---- x <- normalize(x + Wov x + bov) on hidden columns 400..499: fragments 25..31, output tiles j <-> fragments 25 + 2 j, 26 + 2 j ---- */
        sp += (size_t)HHW_ATT_PIECES * HHW_PIECE;
        hhw_issue<WV, HHW_L2_PIECES>(sp, buf0, wave, lane);
        float y[4][16];
        float ssum = 0.0f;
        hh_h8 anh = hhw_frag(buf1, 0, lane), anl = hhw_frag(buf1, 1, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hh_f32x16 acc = hhp_zero16();
#pragma unroll
            for (int kb = 0; kb < 7; kb++) {
                hh_h8 ah = anh, al = anl;
                hhw_need2(ah, al);
                if (j * 7 + kb + 1 < 28) { anh = hhw_frag(buf1, (j * 7 + kb + 1) * 2, lane); anl = hhw_frag(buf1, (j * 7 + kb + 1) * 2 + 1, lane); }
                __builtin_amdgcn_sched_barrier(0);
                HHW_MFMA(ah, zh[25 + kb], acc);
                HHW_MFMA(al, zh[25 + kb], acc);
                HHW_MFMA(ah, zl[25 + kb], acc);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int jb = 32 * j + 8 * q + 4 * g; /* column inside the 100-wide block; 100 % 4 == 0 */
                const bool ok = jb < 100;
                const int f = 25 + 2 * j + (q >> 1); /* the fragment that holds these four columns (31 at most when ok) */
                const float4 b = *reinterpret_cast<const float4 *>(bl + 1024 + (ok ? jb : 0));
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int e = 4 * (q & 1) + i;
                    const float x = f < 32 ? (float)zh[f < 32 ? f : 31][e] + (float)zl[f < 32 ? f : 31][e] : 0.0f;
                    const float v = ok ? x + (acc[4 * q + i] + bb[i]) : 0.0f;
                    y[j][4 * q + i] = v;
                    ssum += v * v;
                }
            }
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(ssum), __float_as_int(ssum), false, false);
            const float other = __int_as_float(g ? sw[0] : sw[1]);
            ssum = g ? other + ssum : ssum + other; /* low half + high half on both lanes of the row */
        }
        const float den = fmaxf(sqrtf(ssum), 1e-12f); /* F.normalize divides, and so does this: `y * (1.0f / den)` was folded into an approximate reciprocal
                                                         square root by hipcc and left 0.2 % of real observation rows 2e-5 off the fp32 forward */
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int jb = 32 * j + 8 * q + 4 * g;
                const int f = 25 + 2 * j + (q >> 1);
                if (f < 32) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int e = 4 * (q & 1) + i;
                        const float v = jb < 100 ? y[j][4 * q + i] / den : 0.0f; /* columns 500..511 stay zero */
                        const _Float16 h = (_Float16)v;
                        zh[f][e] = h;
                        zl[f][e] = (_Float16)(v - (float)h);
                    }
                }
            }
        HHW_T(3);
        __syncthreads(); /* (0, 0) landed in buf0; buf1 is free */
        bufA = buf0; bufB = buf1;
        HHW_T(4);
    }

    /* ---- L2 (shared layer) two column tiles at a time over the two K halves.  Software pipeline over the tile pairs: while pair p's MFMAs
     * run, the wave requests the next chunks piece by piece and works off pair p - 1's epilogue (tanh, fp16 split) slice by slice; pair
     * p - 1's share of the output layer (12 MFMAs, weights at the end of chunk (p, 0)) follows its last slice.  Chunk (p, 0) carries the
     * output-layer pieces of pair p - 1, chunk (7, 1) those of pair 7. ---- */
    hh_f32x16 lacc = hhp_zero16();
    hh_f32x16 pv0 = hhp_zero16(), pv1 = hhp_zero16(); /* the finished accumulators of the previous pair */
    const unsigned char *gs = st + lane * 16; /* + chunk offset + the wave's share of that chunk */
    const int l2_0 = (HHW_L1_PIECES + HHW_ATT_PIECES) * HHW_PIECE;    /* byte offset of chunk (0, 0) */

    auto pair = [&](int p, auto first_tag, auto last_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        /* chunk sizes in pieces: (p, 0) = 64 (+ 8 for p > 0), (p, 1) = 64 (+ 8 for p = 7); offsets follow */
        const int off_p1 = l2_0 + hhw_chunk_piece(p, 1) * HHW_PIECE, off_n0 = l2_0 + hhw_chunk_piece(p + 1, 0) * HHW_PIECE;
        hh_f32x16 acc0 = hhw_bias_acc(bl + 512 + 64 * p, g), acc1 = hhw_bias_acc(bl + 512 + 64 * p + 32, g);
        hh_h8 sh[4], sl[4];
        constexpr int N1 = (HHW_L2_PIECES + (LAST ? HHW_L3_PIECES : 0)) / WV; /* this wave's pieces of chunk (p, 1) */
        const unsigned char *s1 = gs + off_p1 + wave * (N1 * HHW_PIECE);
        unsigned char *d1 = bufB + wave * (N1 * HHW_PIECE);
        auto glds1 = [&](int kk) { if (2 * kk < N1) hhw_issue_some(s1, d1, 2 * kk, 2 * kk + 2 <= N1 ? 2 : 1); };
        if constexpr (FIRST) hhw_l2_half<0>(bufA, lane, zh, zl, 0, acc0, acc1, [&](int) { return 0; }, [&](int kk, int) { glds1(kk); });
        else hhw_l2_half_epi(bufA, lane, zh, zl, acc0, acc1, pv0, pv1, sh, sl, glds1);
        HHW_T(6);
#ifdef HHW_ABL_NO_EPI
        if constexpr (!FIRST) { for (int f = 0; f < 4; f++) { sh[f] = zh[f]; sl[f] = zl[f]; } }
#endif
        if constexpr (!FIRST) hhw_l3_pair(bufA, HHW_L2_PIECES, lane, sh, sl, lacc);
        HHW_T(7);
        __syncthreads(); /* (p, 1) landed; bufA is free */
        HHW_T(8);
        constexpr int N0 = (HHW_L2_PIECES + HHW_L3_PIECES) / WV; /* this wave's pieces of chunk (p + 1, 0) */
        const unsigned char *s0 = gs + off_n0 + wave * (N0 * HHW_PIECE);
        unsigned char *d0 = bufA + wave * (N0 * HHW_PIECE);
        hhw_l2_half<0>(bufB, lane, zh, zl, 16, acc0, acc1, [&](int) { return 0; }, [&](int kk, int) {
            if constexpr (!LAST) { if (2 * kk < N0) hhw_issue_some(s0, d0, 2 * kk, 2 * kk + 2 <= N0 ? 2 : 1); }
        });
        HHW_T(9);
        pv0 = acc0; pv1 = acc1;
        if constexpr (LAST) { /* nobody is left to hide pair 7's epilogue behind */
#pragma unroll
            for (int kk = 0; kk < 16; kk++) hhw_epi_slice(kk, pv0, pv1, sh, sl);
            hhw_l3_pair(bufB, HHW_L2_PIECES, lane, sh, sl, lacc);
        }
        HHW_T(10);
        __syncthreads(); /* (p + 1, 0) landed; bufB is free */
        HHW_T(11);
    };
    pair(0, std::true_type{}, std::false_type{});
#pragma nounroll
    for (int p = 1; p < 7; p++) pair(p, std::false_type{}, std::false_type{});
    pair(7, std::false_type{}, std::true_type{});

    /* ---- logits: lane (row, g) holds output columns 8 q + 4 g + (0..3); they meet in LDS for the decode ---- */
    float *Lg = reinterpret_cast<float *>(ldsb); /* [R][32]; every chunk buffer is dead */
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = *reinterpret_cast<const float4 *>(bl + 1152 + 8 * q + 4 * g);
        *reinterpret_cast<float4 *>(Lg + (wave * 32 + ci) * 32 + 8 * q + 4 * g) =
            make_float4(lacc[4 * q] + b.x, lacc[4 * q + 1] + b.y, lacc[4 * q + 2] + b.z, lacc[4 * q + 3] + b.w);
    }
    __syncthreads();
    if (logits_out)
        for (int e = tid; e < R * 32; e += NTH) {
            const int i = e >> 5, c = e & 31;
            if (rows[i] >= 0) logits_out[(size_t)rows[i] * HH_POLICY_LOGITS + c] = c < N.n_out ? Lg[e] : 0.0f;
        }
    /* greedy decode (env_base.py:373-382), one thread per (row, MultiDiscrete component): first maximum of its segment */
    for (int e = tid; e < R * 4; e += NTH) {
        const int i = e >> 2, k = e & 3;
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
    HHW_T(12);
    HHW_TFLUSH;
}

template <int WV>
__global__ __launch_bounds__(64 * WV, 1) void hh_k_policy_w(HhpBank bank, HhpBankW bankw, int n_nets, const float *__restrict__ obs, int obs_stride,
                                                             int *counts, const int *__restrict__ lists, int max_rows, int8_t *__restrict__ actions,
                                                             float *__restrict__ logits_out, int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    int net, tile, cnt;
    if (hhp_locate<32 * WV>(cn, (int)blockIdx.x, net, tile, cnt))
        hhw_forward_tile<WV>(bank.net[net], bankw.net[net], obs, obs_stride, lists + (size_t)net * max_rows, tile, cnt, actions, logits_out, ldsb);
    hhp_consume_counts(counts, consume);
}

/* host: element (k, col) of a [K x J] operand -> piece-local halves index; natural k order (nat = true: k group = (k >> 3) & 1, e = k & 7)
 * or the order of a transposed C tile's registers (hhp_hidx_t) */
static inline void hhw_put(std::vector<uint16_t> &S, size_t piece_hi, int k, int col, bool nat, float v) {
    const int w = k & 15;
    const int g = nat ? (w >> 3) : ((w >> 2) & 1), e = nat ? (w & 7) : (((w >> 3) << 2) | (w & 3));
    const size_t at = (size_t)((g * 32 + (col & 31)) * 8 + e); /* lane = g * 32 + column in tile; 8 halves per lane */
    const uint16_t h = hhp_f2h(v);
    S[piece_hi * (HHW_PIECE / 2) + at] = h;
    S[(piece_hi + 1) * (HHW_PIECE / 2) + at] = hhp_f2h(v - hhp_h2f(h));
}

#endif /* HH_POLICY_KERNEL_W_H */
