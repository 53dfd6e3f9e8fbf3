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
 * Chunks (pieces): L1 tiles 0..15 (32) | L1 tiles 16..31 (32) | ATT tiles 0..3 (32) | ATT tiles 4..6 (24 + 8 of padding) | 8 column groups x 4 K quarters
 * (32 each: 4 k-blocks x 4 tiles x (hi, lo)) | output layer 64 (8 per column group, side buffer).
 */
#ifndef HH_POLICY_KERNEL_W16_H
#define HH_POLICY_KERNEL_W16_H

typedef float hh_f32x4 __attribute__((ext_vector_type(4)));

#define HHX_CHUNK 32 /* pieces per LDS buffer */
#define HHX_BUF_BYTES (HHX_CHUNK * HHW_PIECE)
#define HHX_L1_PIECES 64
#define HHX_ATT_PIECES 64 /* 7 tiles x 4 k-blocks x (hi, lo) = 56, padded to two whole chunks */
#define HHX_L2_PIECES 1024
#define HHX_L3_PIECES 64
#define HHX_STREAM_PIECES (HHX_L1_PIECES + HHX_ATT_PIECES + HHX_L2_PIECES + HHX_L3_PIECES)
/* LDS: a ring of NB chunk buffers (2: one chunk ahead, the 64-row form, two workgroups per CU | 4: three ahead, the 128-row form) | biases b1 512, bs 512,
 * bov 128, ba 32 floats | row ids (<= 128) | the head's eight pieces of one column group */
#define HHX_OFF_BIAS_NB(NB) ((NB) * HHX_BUF_BYTES)
#define HHX_OFF_ROWS_NB(NB) (HHX_OFF_BIAS_NB(NB) + (512 + 512 + 128 + 32) * 4)
#define HHX_OFF_L3_NB(NB) (HHX_OFF_ROWS_NB(NB) + 128 * 4)
#define HHX_LDS_BYTES_NB(NB) (HHX_OFF_L3_NB(NB) + 8 * HHW_PIECE)
#define HHX_OFF_BIAS HHX_OFF_BIAS_NB(2)
#define HHX_OFF_ROWS HHX_OFF_ROWS_NB(2)
#define HHX_LDS_BYTES HHX_LDS_BYTES_NB(2)

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

/* tuning builds (-DHHP_PROFILE): per-phase cycles of wave 0 of every tile, summed in registers, flushed once (tools/policy_w16_phase_profile.py) */
#ifdef HHP_PROFILE
struct HhxProf { unsigned long long acc[16], pt, rt; }; /* slot 15: the tile in s_memrealtime ticks (100 MHz): slots 0..9 against it = the shader clock the tile ran at */
#define HHX_T0(P) do { for (int k_ = 0; k_ < 16; k_++) (P).acc[k_] = 0; (P).rt = __builtin_amdgcn_s_memrealtime(); (P).pt = __builtin_readcyclecounter(); } while (0)
#ifdef HHP_PROFILE_LITE /* the two ends of the tile only: the phase stamps (an s_waitcnt lgkmcnt(0) each) perturb what they measure */
#define HHX_T(P, k) do { if ((k) == 9) { const unsigned long long t_ = __builtin_readcyclecounter(); (P).acc[k] += t_ - (P).pt; (P).pt = t_; } } while (0)
#else
#define HHX_T(P, k) do { const unsigned long long t_ = __builtin_readcyclecounter(); (P).acc[k] += t_ - (P).pt; (P).pt = t_; } while (0)
#endif
#define HHX_TFLUSH(P) do { (P).acc[15] = __builtin_amdgcn_s_memrealtime() - (P).rt; if (threadIdx.x == 0) for (int k_ = 0; k_ < 16; k_++) atomicAdd(&hhp_prof[k_], (P).acc[k_]); } while (0)
#else
struct HhxProf {};
#define HHX_T0(P)
#define HHX_T(P, k)
#define HHX_TFLUSH(P)
#endif

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
/* the fp16 split of two values in four instructions: hi pair = v_cvt_pk_f16_f32 (round to nearest even, like a scalar cast), the two exact remainders
 * v - (float)hi by v_fma_mix_f32 straight from the packed halves (no unpacking conversion, no subtraction of its own), lo pair = v_cvt_pk_f16_f32.
 * The same bits as `h = (_Float16)v; l = (_Float16)(v - (float)h)`, which hipcc spells with eight. */
typedef _Float16 hh_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hhx_split2(hh_f2 v, unsigned &hi, unsigned &lo) {
    union { hh_h2 h; unsigned u; } a, b;
    a.h = __builtin_convertvector(v, hh_h2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a.u), "v"(v.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a.u), "v"(v.y));
    b.h = __builtin_convertvector(hh_f2{r0, r1}, hh_h2);
    hi = a.u;
    lo = b.u;
}
__device__ __forceinline__ void hhx_split8(const hh_f2 (&p)[4], hh_h8 &fh, hh_h8 &fl) {
    union { unsigned u[4]; hh_h8 h; } H, L;
#pragma unroll
    for (int i = 0; i < 4; i++) hhx_split2(p[i], H.u[i], L.u[i]);
    fh = H.h;
    fl = L.h;
}
/* The hand-over of a chunk: this wave's LDS-DMA pieces of it have landed — at most NEWER requests issued after them may still be in flight (they complete
 * in order) —, its LDS reads of the chunk before are back, then everyone's.  __syncthreads() is this with NEWER = 0; a ring deeper than two buffers
 * needs the count (hipcc does not order LDS reads against LDS-DMA on its own: the explicit wait IS the ordering). */
template <int NEWER>
__device__ __forceinline__ void hhx_chunk_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEWER) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
/* 1 / max(sqrt(s), 1e-12), correctly rounded from the correctly rounded norm: F.normalize divides every element by the norm, and a product with this
 * reciprocal is within 1.5 ulp of that quotient.  The empty asm hides the norm from hipcc, which otherwise folds 1 / max(sqrt(s), eps) into ONE approximate
 * v_rsq_f32: 0.2 % of real observation rows then leave the fp32 forward by 2e-5 (found by the world-observation test of tests/test_policy_nets.py). */
__device__ __forceinline__ float hhx_recip_norm(float ssum) {
    float den = fmaxf(sqrtf(ssum), 1e-12f);
    asm volatile("" : "+v"(den));
    return 1.0f / den;
}
/* four values into elements 2 w .. 2 w + 3 of a fragment's (hi, lo) halves (w = 0 | 2: a compile-time constant where it is used) */
__device__ __forceinline__ void hhx_put4(hh_h8 &fh, hh_h8 &fl, int w, hh_f2 v01, hh_f2 v23) {
    union { unsigned u[4]; hh_h8 h; } H, L;
    H.h = fh;
    L.h = fl;
    hhx_split2(v01, H.u[w], L.u[w]);
    hhx_split2(v23, H.u[w + 1], L.u[w + 1]);
    fh = H.h;
    fl = L.h;
}
/* tanh of two adjacent C^T tiles (bias already in the accumulators) -> the (hi, lo) halves of one B fragment */
__device__ __forceinline__ void hhx_pair_to_frag(const hh_f32x4 &a0, const hh_f32x4 &a1, hh_h8 &fh, hh_h8 &fl) {
    const hh_f2 p[4] = {hhp_tanh2(hh_f2{a0[0], a0[1]}), hhp_tanh2(hh_f2{a0[2], a0[3]}), hhp_tanh2(hh_f2{a1[0], a1[1]}), hhp_tanh2(hh_f2{a1[2], a1[3]})};
    hhx_split8(p, fh, fl);
}

/* The shared layer and the head contracted from its registers: 8 groups of four column tiles x 4 K quarters = 32 chunks of 32 pieces, chunk c in ring
 * buffer (j0 + c) % NB (j0 = what the caller streamed before), requested D = NB - 1 chunks ahead: on entry chunks 0 .. D - 1 are on their way, chunk c + D is
 * requested behind the MFMAs of chunk c's first four steps.  l2s = chunk 0 in the stream.  The head's NOUT 16-column output tiles are accumulated in lacc;
 * its pieces of group p (2 k-blocks x NOUT x (hi, lo), contiguous at l3 + p x 4 NOUT pieces) travel beside the group's first chunk into a side buffer
 * l3buf of their own: read from global memory where they are used, their L2 latency stood in the open eight times a tile (tools/policy_w16_phase_profile.py:
 * 8 k of a tile's 69 k cycles).  bl = the biases in LDS. */
template <int NOUT, int WV, int NB>
__device__ __forceinline__ void hhx_l2_l3(const hh_h8 (&zh)[16], const hh_h8 (&zl)[16], unsigned char *ring, int j0, const unsigned char *l2s,
                                          const unsigned char *l3, unsigned char *l3buf, const float *__restrict__ bl, int wave, int lane, int g,
                                          hh_f32x4 (&lacc)[NOUT], HhxProf &pf) {
    static_assert(NOUT == 1 || NOUT == 2, "one piece (value head) or two (policy head) per wave and group");
    static_assert(NB == 2 || NB == 4, "one chunk ahead or three");
    constexpr int D = NB - 1;
    constexpr int NPW = HHX_CHUNK / WV;                              /* a wave's pieces of a chunk: 8 (four waves) | 4 (eight) */
    constexpr int HPW = 4 * NOUT >= WV ? 4 * NOUT / WV : 1;         /* and of a group's 4 NOUT head pieces */
    const unsigned char *gsrc = l2s + (size_t)D * HHX_BUF_BYTES + (size_t)wave * NPW * HHW_PIECE + lane * 16; /* this wave's pieces of the next chunk to request */
    /* ---- L2 (shared layer): 8 groups of four column tiles x 4 K quarters; the output layer from the group's registers ----
     * NB = 2: chunk (p, q) is handed over at its top (everyone's pieces landed, the other buffer free), its first fragments are read behind that barrier,
     *   chunk + 1 is requested in steps 0..3.
     * NB = 4: the hand-over sits INSIDE the chunk before, after its second step: there every wave makes sure its pieces of chunk + 1 have landed and meets
     *   the others, which also says that everyone is done with chunk - 1, whose buffer takes the requests of chunk + 3 in steps 2..5.  A chunk therefore
     *   starts with its data in LDS and its first four fragments already in registers (requested in the last step of the chunk before; across the group's
     *   epilogue: behind it): no barrier, no LDS round trip in front of its MFMAs — at the top they cost ~600 of a chunk's 2 200 cycles with the matrix
     *   pipe idle. */
#pragma unroll
    for (int t = 0; t < NOUT; t++) lacc[t] = hh_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    hh_h8 an[4];
    if (NB == 4) {
        hhx_chunk_barrier<2 * NPW>(); /* chunk 0 (chunks 1 and 2 may be on their way); everyone is out of the phase before */
        HHX_T(pf, 5);
#pragma unroll
        for (int u = 0; u < 4; u++) an[u] = hhw_frag(ring + (j0 & (NB - 1)) * HHX_BUF_BYTES, u, lane);
    }
#pragma nounroll
    for (int p = 0; p < 8; p++) {
        hh_f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = hhx_bias_acc(bl + 512 + 64 * p + 16 * t, g);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned char *cb = ring + ((j0 + q) & (NB - 1)) * HHX_BUF_BYTES;
            const unsigned char *cbn = ring + ((j0 + q + 1) & (NB - 1)) * HHX_BUF_BYTES;
            unsigned char *gdst = ring + ((j0 + q + D) & (NB - 1)) * HHX_BUF_BYTES + wave * NPW * HHW_PIECE; /* requests: one at a time behind a step's MFMAs (60 - 100 cycles each to issue) */
            const bool more = p < 7 || q + D < 4; /* chunk (p, q) + D exists */
            if (NB == 2) {
#ifndef HHX_ABL_NO_BARRIER /* tuning builds: the shared layer's chunks are not waited for (wrong results): what do the 32 barriers + waits cost? */
                hhx_chunk_barrier<0>();
#endif
                HHX_T(pf, 5);
#pragma unroll
                for (int u = 0; u < 4; u++) an[u] = hhw_frag(cb, u, lane);
            }
            { /* eight steps of (k-block kk, tile pair tp) = 4 fragments, 6 MFMAs; the fragments of step s + 1 are requested before the MFMAs of step s */
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
                    } else if (NB == 4 && q < 3) { /* the chunk after this one is in LDS since this chunk's hand-over */
#pragma unroll
                        for (int u = 0; u < 4; u++) an[u] = hhw_frag(cbn, u, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    HHX_MFMA(a[0], zh[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[2], zh[4 * q + kk], acc[2 * tp + 1]);
#ifndef HHX_ABL_THIRD_MFMA /* tuning builds: the hi x hi products only (wrong results): is the matrix pipe what two co-resident tiles fight over? */
                    HHX_MFMA(a[1], zh[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[3], zh[4 * q + kk], acc[2 * tp + 1]);
                    HHX_MFMA(a[0], zl[4 * q + kk], acc[2 * tp]); HHX_MFMA(a[2], zl[4 * q + kk], acc[2 * tp + 1]);
#endif
                    if (NB == 4 && s_ == 1) { /* the hand-over of chunk + 1: at most chunk + 2's requests (if it exists) stay in flight */
#ifndef HHX_ABL_NO_BARRIER
                        if (p < 7 || q < 2) hhx_chunk_barrier<NPW>();
                        else hhx_chunk_barrier<0>();
#endif
                        HHX_T(pf, 5);
                    }
#ifndef HHX_ABL_NO_GLDS /* tuning builds: the shared layer's chunks are never copied (wrong results): what does the LDS-DMA stream cost? */
                    if (NB == 2) { if (s_ < 4 && more) hhw_issue_some(gsrc, gdst, NPW / 4 * s_, NPW / 4); }
                    else { if (s_ >= 2 && s_ < 6 && more) hhw_issue_some(gsrc, gdst, NPW / 4 * (s_ - 2), NPW / 4); }
                    if (q == 0 && s_ == (NB == 2 ? 4 : 6) && wave * HPW < 4 * NOUT) { /* everyone is past the hand-over in (p, 0), so past its reads of group p - 1's head pieces;
                                                                                        * these are older than anything the hand-over before the group's end lets stay in flight */
                        const unsigned char *hs = l3 + (size_t)(p * 4 * NOUT + wave * HPW) * HHW_PIECE + lane * 16;
                        hhw_issue_some(hs, l3buf + wave * HPW * HHW_PIECE, 0, HPW);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            gsrc += (size_t)HHX_BUF_BYTES;
            HHX_T(pf, 6);
        }
        /* tanh of the four tiles = two B fragments of the output layer (S columns 64 p .. 64 p + 63 = k-blocks 2 p, 2 p + 1) */
#pragma unroll
        for (int u = 0; u < 2; u++) {
            hh_h8 sh, sl;
#ifdef HHX_ABL_SKIP_EPI /* tuning builds: no tanh / split behind a column group (wrong results) */
            sh = zh[u]; sl = zl[u];
            asm volatile("" :: "v"(acc[2 * u]), "v"(acc[2 * u + 1]));
#else
            hhx_pair_to_frag(acc[2 * u], acc[2 * u + 1], sh, sl);
#endif
            HHX_T(pf, 7);
#pragma unroll
            for (int t = 0; t < NOUT; t++) {
                const hh_h8 wh = hhw_frag(l3buf, (u * NOUT + t) * 2, lane), wl = hhw_frag(l3buf, (u * NOUT + t) * 2 + 1, lane);
                HHX_MFMA(wh, sh, lacc[t]);
                HHX_MFMA(wl, sh, lacc[t]);
                HHX_MFMA(wh, sl, lacc[t]);
            }
            HHX_T(pf, 8);
        }
        if (NB == 4 && p < 7) { /* the next group's first chunk: in LDS since the hand-over inside (p, 3); (j0 + 4) % 4 = j0 % 4 */
#pragma unroll
            for (int u = 0; u < 4; u++) an[u] = hhw_frag(ring + (j0 & (NB - 1)) * HHX_BUF_BYTES, u, lane);
        }
    }
}

/* one 64-row tile of one network: four waves of 16 rows.  SAMPLE: the PPO sampler's tail (hh_policy_sample: a Categorical draw per action
 * component, its log-probability) instead of the greedy decode */
template <bool SAMPLE, int WV>
__device__ __forceinline__ void hhx_forward_tile(const HhpNet &N, const unsigned char *__restrict__ st, const float *__restrict__ obs, int obs_stride,
                                                 const int *__restrict__ list, int tile, int cnt, int8_t *__restrict__ actions, float *__restrict__ logits_out,
                                                 unsigned char *ldsb, const HhpSampleArgs *sa = nullptr) {
    constexpr int NTH = 64 * WV, R = 16 * WV, NPW = HHX_CHUNK / WV;
    constexpr int NB = WV == 8 ? 4 : 2, D = NB - 1; /* ring buffers; chunks requested ahead */
#ifdef HHX_ABL_SKIP_P
    const int has_att = 0;
#else
    const int has_att = N.has_att;
#endif
    float *bl = reinterpret_cast<float *>(ldsb + HHX_OFF_BIAS_NB(NB));
    int *rows = reinterpret_cast<int *>(ldsb + HHX_OFF_ROWS_NB(NB));
    /* the stream in PROCESSING order: chunks 0, 1 = first layer, then (fight nets) 2, 3 = attention block, then the shared layer's 32; chunk j sits in
     * ring buffer j % NB and is requested while chunk j - D is worked on (the first D up front) */
    const int att_skip = has_att ? 0 : HHX_ATT_PIECES / HHX_CHUNK;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid)); /* the callers walk their tiles in a loop: what derives from the thread index is recomputed per tile, not held in registers across the tile before */
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 15, g = lane >> 4;

    auto request = [&](int j, int first, int n) { /* pieces first .. first + n - 1 of this wave's NPW of chunk j */
        hhw_issue_some(st + (size_t)(j < 2 ? j : j + att_skip) * HHX_BUF_BYTES + (size_t)wave * NPW * HHW_PIECE + lane * 16,
                       ldsb + (j % NB) * HHX_BUF_BYTES + wave * NPW * HHW_PIECE, first, n);
    };
    HhxProf pf;
    HHX_T0(pf);
    request(0, 0, NPW); /* chunk 0: L1 tiles 0..15 */
    for (int e = tid; e < 512; e += NTH) { bl[e] = N.b1[e]; bl[512 + e] = N.bs[e]; }
    if (tid < 128) bl[1024 + tid] = has_att ? N.bov[tid] : 0.0f;
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
    hh_h8 zh[16], zl[16]; /* the hidden row: fragment kb = columns 32 kb .. 32 kb + 31 in hhw16_korder */

    /* ---- L1: 32 column tiles of 16, K = one block of 32 observation columns; two chunks of 16 tiles.  Steps of one tile pair = one fragment of
     *      the hidden row: the pair's four weight fragments are requested a step ahead (an LDS round trip in the open per pair was a third of this
     *      phase), the next chunk's eight LDS-DMA requests of this wave go out one per step ---- */
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c == 0) hhx_chunk_barrier<0>(); /* chunk c landed (nothing else is on its way yet: hipcc waits for the observation with vmcnt(0) anyway) */
        else hhx_chunk_barrier<(D - 1) * NPW>();
        HHX_T(pf, c == 0 ? 0 : 2);
        const unsigned char *cb = ldsb + c * HHX_BUF_BYTES;
        hh_h8 an[4];
#pragma unroll
        for (int u = 0; u < 4; u++) an[u] = hhw_frag(cb, u, lane);
#pragma unroll
        for (int tp = 0; tp < 8; tp++) {
            hh_h8 a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = an[u];
            hhw_need4(a);
            if (tp + 1 < 8) {
#pragma unroll
                for (int u = 0; u < 4; u++) an[u] = hhw_frag(cb, (tp + 1) * 4 + u, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            hh_f32x4 a0 = hhx_bias_acc(bl + 16 * (c * 16 + tp * 2), g), a1 = hhx_bias_acc(bl + 16 * (c * 16 + tp * 2 + 1), g);
            HHX_MFMA(a[0], xh, a0); HHX_MFMA(a[2], xh, a1);
            HHX_MFMA(a[1], xh, a0); HHX_MFMA(a[3], xh, a1);
            HHX_MFMA(a[0], xl, a0); HHX_MFMA(a[2], xl, a1);
            if (D == 1 || c == 1) { if (tp % (8 / NPW) == 0) request(c + D, tp / (8 / NPW), 1); }
            else if (tp < 4) request(1 + tp / 2, (tp & 1) * (NPW / 2), NPW / 2); /* the deep ring fills behind chunk 0: chunks 1, 2 whole (in order: they complete in order) ... */
            else request(D, (tp - 4) * (NPW / 4), NPW / 4);                         /* ... then chunk 3 */
#ifdef HHX_ABL_SKIP_P /* tuning builds: no tanh / split behind the first layer, no attention block (wrong results): what does the shared layer take when nothing precedes it? */
            zh[c * 8 + tp] = a[0]; zl[c * 8 + tp] = a[1];
            asm volatile("" :: "v"(a0), "v"(a1));
#else
            hhx_pair_to_frag(a0, a1, zh[c * 8 + tp], zl[c * 8 + tp]);
#endif
        }
        HHX_T(pf, c == 0 ? 1 : 3);
    }

    /* ---- fight nets: x <- normalize(x + Wov x + bov) on hidden columns 400..499.  K = fragments 12..15 (columns 384..511, the weights of
     *      384..399 are zero); output tile j (columns 400 + 16 j ..) is half (25 + j) & 1 of fragment (25 + j) >> 1.  Steps of (tile pair, k-block)
     *      like the shared layer's (two accumulators in turn instead of twelve dependent MFMAs on one), fragments a step ahead ---- */
    if (has_att) {
        hh_f32x4 y[7];
        float ssum = 0.0f;
        auto fold = [&](int j, const hh_f32x4 &acc) {
            const int f = (25 + j) >> 1, e0 = 4 * ((25 + j) & 1);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool ok = 16 * j + 4 * g + r < 100;
                const float x = (float)zh[f][e0 + r] + (float)zl[f][e0 + r];
                const float v = ok ? x + acc[r] : 0.0f;
                y[j][r] = v;
                ssum += v * v;
            }
        };
#pragma unroll
        for (int c = 0; c < 2; c++) {
            hhx_chunk_barrier<(D - 1) * NPW>();
            const unsigned char *cb = ldsb + ((2 + c) % NB) * HHX_BUF_BYTES;
            /* tile pairs (0, 1) (2, 3) | (4, 5); piece of (tile jj of the chunk, k-block kb, plane) = (jj * 4 + kb) * 2 + plane */
#pragma unroll
            for (int pr = 0; pr < (c == 0 ? 2 : 1); pr++) {
                const int j0 = c * 4 + pr * 2;
                hh_f32x4 acc0 = hhx_bias_acc(bl + 1024 + 16 * j0, g), acc1 = hhx_bias_acc(bl + 1024 + 16 * (j0 + 1), g);
                hh_h8 an[4];
                an[0] = hhw_frag(cb, (pr * 2 * 4) * 2, lane); an[1] = hhw_frag(cb, (pr * 2 * 4) * 2 + 1, lane);
                an[2] = hhw_frag(cb, ((pr * 2 + 1) * 4) * 2, lane); an[3] = hhw_frag(cb, ((pr * 2 + 1) * 4) * 2 + 1, lane);
#pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    hh_h8 a[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) a[u] = an[u];
                    hhw_need4(a);
                    if (kb + 1 < 4) {
                        an[0] = hhw_frag(cb, (pr * 2 * 4 + kb + 1) * 2, lane); an[1] = hhw_frag(cb, (pr * 2 * 4 + kb + 1) * 2 + 1, lane);
                        an[2] = hhw_frag(cb, ((pr * 2 + 1) * 4 + kb + 1) * 2, lane); an[3] = hhw_frag(cb, ((pr * 2 + 1) * 4 + kb + 1) * 2 + 1, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    HHX_MFMA(a[0], zh[12 + kb], acc0); HHX_MFMA(a[2], zh[12 + kb], acc1);
                    HHX_MFMA(a[1], zh[12 + kb], acc0); HHX_MFMA(a[3], zh[12 + kb], acc1);
                    HHX_MFMA(a[0], zl[12 + kb], acc0); HHX_MFMA(a[2], zl[12 + kb], acc1);
                    if (pr * 4 + kb < NPW) request(2 + c + D, pr * 4 + kb, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                fold(j0, acc0);
                fold(j0 + 1, acc1);
            }
            if (c == 1) { /* tile 6 on its own */
                hh_f32x4 acc = hhx_bias_acc(bl + 1024 + 16 * 6, g);
                hh_h8 w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = hhw_frag(cb, 2 * 4 * 2 + u, lane);
#pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    HHX_MFMA(w[2 * kb], zh[12 + kb], acc);
                    HHX_MFMA(w[2 * kb + 1], zh[12 + kb], acc);
                    HHX_MFMA(w[2 * kb], zl[12 + kb], acc);
                    if (4 + kb < NPW) request(2 + c + D, 4 + kb, 1);
                }
                fold(6, acc);
            }
        }
        ssum += __shfl_xor(ssum, 16);
        ssum += __shfl_xor(ssum, 32); /* the four k groups of a row: (s0 + s1) + (s2 + s3) on every lane */
        const float rden = hhx_recip_norm(ssum);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int f = (25 + j) >> 1, e0 = 4 * ((25 + j) & 1);
            hhx_put4(zh[f], zl[f], e0 >> 1, hh_f2{y[j][0] * rden, y[j][1] * rden}, hh_f2{y[j][2] * rden, y[j][3] * rden}); /* columns >= 500 were zeroed above */
        }
    }

    HHX_T(pf, 4);
    /* ---- L2 (shared layer) + the output layer from its registers ---- */
    hh_f32x4 lacc[2];
    hhx_l2_l3<2, WV, NB>(zh, zl, ldsb, has_att ? 4 : 2, st + (size_t)(HHX_L1_PIECES + HHX_ATT_PIECES) * HHW_PIECE,
                         st + (size_t)(HHX_L1_PIECES + HHX_ATT_PIECES + HHX_L2_PIECES) * HHW_PIECE, ldsb + HHX_OFF_L3_NB(NB), bl, wave, lane, g, lacc, pf);

    /* ---- logits: lane (row, g) holds output columns 16 t + 4 g + (0..3); they meet in LDS for the decode ---- */
    __syncthreads(); /* every wave is done with the chunk buffers */
    float *Lg = reinterpret_cast<float *>(ldsb); /* [R][32] */
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
    if constexpr (!SAMPLE) { /* greedy decode (env_base.py:373-382), one thread per (row, MultiDiscrete component): first maximum of its segment */
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
    } else { /* TorchMultiCategorical over [13, 9, 2, 2] (| [13, 9, 2]): the tail of hh_k_policy_ppo's actor tiles (hh_policy_kernel_ppo.h), same arithmetic */
        const int i = tid >> 2, k = tid & 3, r = rows[i];
        const int lo = k == 0 ? 0 : (k == 1 ? 13 : (k == 2 ? 22 : 24)), hi = k == 0 ? 13 : (k == 1 ? 22 : (k == 2 ? 24 : 26));
        const float *lg = Lg + i * 32;
        int a = 0;
        float lp = 0.0f;
        if (k < (N.n_out == 26 ? 4 : 3)) {
            float m = lg[lo];
            int best = lo;
            for (int c = lo + 1; c < hi; c++) if (lg[c] > m) { m = lg[c]; best = c; }
            float S = 0.0f;
            for (int c = lo; c < hi; c++) S += __expf(lg[c] - m);
            a = best - lo;
            if (!sa->greedy && r >= 0) {
                double u;
                if (sa->uniforms) u = sa->uniforms[(size_t)r * 4 + k];
                else {
                    const int n = r / sa->rows_per_arena, sl_ = r - n * sa->rows_per_arena;
                    const int4 ap = sa->ar_pack[n];
                    u = hh_rng_u01(hh_rng_tick_key(hh_rng_arena_key(sa->seed, sa->arena_offset + (unsigned long long)n), (uint32_t)ap.y, (uint32_t)ap.x),
                                   (uint32_t)(sl_ + 1), HH_SITE_POLICY_SAMPLE, (uint32_t)k);
                }
                const float t = (float)u * S;
                float cum = 0.0f;
                a = hi - lo - 1;
                bool found = false;
                for (int c = lo; c < hi; c++) {
                    cum += __expf(lg[c] - m);
                    if (!found && cum > t) { a = c - lo; found = true; }
                }
            }
            lp = (lg[lo + a] - m) - logf(S);
        }
        lp += __shfl_xor(lp, 1);
        lp += __shfl_xor(lp, 2);
        int aw = a << (8 * k);
        aw |= __builtin_amdgcn_mov_dpp(aw, 0xB1, 0xf, 0xf, true);
        aw |= __builtin_amdgcn_mov_dpp(aw, 0x4E, 0xf, 0xf, true);
        if (k == 0 && r >= 0) {
            reinterpret_cast<int *>(actions)[r] = aw;
            if (sa->logp) sa->logp[r] = lp;
        }
    }
    HHX_T(pf, 9);
    HHX_TFLUSH(pf);
}

/* ---- the value branch as a 64-row tile (hh_policy_sample; see hh_policy_kernel_ppo.h for the reference lines): input row = own observation |
 * own action | the other agent's observation (row r ^ 1) | its action in THREE 32-column k-blocks; hidden columns in the critic's order
 * (attention block first: fragments 0..4 = columns 0..159); chunks: first layer 8 x (4 tiles x 3 k-blocks x (hi, lo) = 24 pieces), attention 5 x
 * (2 tiles x 5 k-blocks x 2 = 20 pieces, fight nets), then the shared layer's 32 chunks; the head is one 16-column tile whose column 0 is the value. */
#define HHXC_L1_PIECES 192
#define HHXC_ATT_PIECES 100
#define HHXC_L3_PIECES 32
#define HHXC_STREAM_PIECES (HHXC_L1_PIECES + HHXC_ATT_PIECES + HHX_L2_PIECES + HHXC_L3_PIECES)
struct HhpCritX {
    const unsigned char *stream;
    const float *b1, *bs, *bov, *ba; /* [512], [512], [160], [32] */
    int d1, a1, d2, a2, has_att, loaded;
};
struct HhpCritBankX {
    HhpCritX c[HH_POLICY_MAX_NETS];
};
#define HHXC_OFF_ROWS (HHX_OFF_BIAS + (512 + 512 + 160 + 32) * 4)
#define HHXC_OFF_L3 (HHXC_OFF_ROWS + 64 * 4)
#define HHXC_LDS_BYTES (HHXC_OFF_L3 + 8 * HHW_PIECE) /* the value head needs four pieces; both tile kinds of hh_k_policy_w16_ppo share one launch size */

template <bool ATT> /* fight nets (attention block, 13 chunks ahead of the shared layer) | escape nets (8): compile-time, so that the buffer parity is too */
__device__ __forceinline__ void hhx_critic_tile(const HhpCritX &Cw, const float *__restrict__ obs, int obs_stride, const int *__restrict__ list, int tile, int cnt,
                                                const HhpSampleArgs &sa, unsigned char *ldsb) {
    constexpr int NTH = 256, R = 64;
    unsigned char *const buf[2] = {ldsb, ldsb + HHX_BUF_BYTES};
    float *bl = reinterpret_cast<float *>(ldsb + HHX_OFF_BIAS); /* b1 512 | bs 512 | bov 160 | ba 32 */
    int *rows = reinterpret_cast<int *>(ldsb + HHXC_OFF_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 15, g = lane >> 4;
    const unsigned char *st = Cw.stream;

    hhx_issue<6>(st, buf[0], wave, lane); /* chunk 0: first-layer tiles 0..3 (24 pieces) */
    for (int e = tid; e < 512; e += NTH) { bl[e] = Cw.b1[e]; bl[512 + e] = Cw.bs[e]; }
    if (tid < 160) bl[1024 + tid] = ATT ? Cw.bov[tid] : 0.0f;
    if (tid < 32) bl[1184 + tid] = Cw.ba[tid];
    const int q_ = tile * R + wave * 16 + ci;
    const int row = q_ < cnt ? list[q_] : -1;
    if (g == 0) rows[wave * 16 + ci] = row;
    hh_h8 xh[3], xl[3]; /* central_critic_observer's row (train_hetero.py:162-181): lane (row, g) holds columns 32 b + 8 g .. + 7 of k-block b */
    {
        const int e1 = Cw.d1, e2 = e1 + Cw.a1, e3 = e2 + Cw.d2, e4 = e3 + Cw.a2;
        const float *ca = sa.crit_act;
        float xv[24];
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = 32 * b + 8 * g + e;
                const bool own = c < e2, is_obs = c < e1 || (c >= e2 && c < e3);
                const int cc = c < e1 ? c : (c < e2 ? c - e1 : (c < e3 ? c - e2 : c - e3));
                const bool ok = row >= 0 && c < e4 && (is_obs || ca != nullptr);
                const size_t rr = (size_t)(own ? row : row ^ 1);
                const float *src = is_obs ? obs + rr * obs_stride + cc : ca + rr * 4 + cc;
                const float x = *(ok ? src : obs);
                xv[8 * b + e] = ok ? x : 0.0f;
            }
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const _Float16 h = (_Float16)xv[8 * b + e];
                xh[b][e] = h;
                xl[b][e] = (_Float16)(xv[8 * b + e] - (float)h);
            }
    }
    const unsigned char *sp = st + (size_t)24 * HHW_PIECE; /* the next chunk to request */
    hh_h8 zh[16], zl[16];
    int ck = 0; /* chunk counter: chunk ck sits in buf[ck & 1] */

    /* ---- first layer: 8 chunks of 4 tiles (two fragments of the hidden row each).  Steps of (tile pair, k-block) = 4 fragments + 6 MFMAs on two accumulators
     *      in turn, the fragments a step ahead, this wave's requests of the next chunk one per step (the scheme of the actor tile: nine dependent MFMAs per tile
     *      and an LDS round trip in the open per tile before) ---- */
#pragma unroll
    for (int c = 0; c < 8; c++, ck++) {
        __syncthreads(); /* chunk ck landed; the other buffer is free */
        const int npw = c < 7 ? 6 : (ATT ? 5 : 8); /* next: first-layer chunk (24 pieces) | attention tiles 0, 1 (20) | escape nets: shared layer (0, 0) (32) */
        if (c == 7 && !ATT) sp += (size_t)HHXC_ATT_PIECES * HHW_PIECE;
        const unsigned char *gsrc = sp + (size_t)wave * npw * HHW_PIECE + lane * 16;
        unsigned char *gdst = buf[(ck + 1) & 1] + wave * npw * HHW_PIECE;
        sp += (size_t)(npw * 4) * HHW_PIECE;
        const unsigned char *cb = buf[ck & 1];
        /* piece of (tile tt of the chunk, k-block kb, plane) = (tt * 3 + kb) * 2 + plane; step st = (pair tp, kb) */
        hh_h8 an[4];
        an[0] = hhw_frag(cb, 0, lane); an[1] = hhw_frag(cb, 1, lane); an[2] = hhw_frag(cb, 6, lane); an[3] = hhw_frag(cb, 7, lane);
        hh_f32x4 a0, a1;
#pragma unroll
        for (int st_ = 0; st_ < 6; st_++) {
            const int tp = st_ / 3, kb = st_ % 3;
            if (kb == 0) { a0 = hhx_bias_acc(bl + 16 * (c * 4 + tp * 2), g); a1 = hhx_bias_acc(bl + 16 * (c * 4 + tp * 2 + 1), g); }
            hh_h8 a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = an[u];
            hhw_need4(a);
            if (st_ + 1 < 6) {
                const int tpn = (st_ + 1) / 3, kbn = (st_ + 1) % 3;
                an[0] = hhw_frag(cb, ((tpn * 2) * 3 + kbn) * 2, lane); an[1] = hhw_frag(cb, ((tpn * 2) * 3 + kbn) * 2 + 1, lane);
                an[2] = hhw_frag(cb, ((tpn * 2 + 1) * 3 + kbn) * 2, lane); an[3] = hhw_frag(cb, ((tpn * 2 + 1) * 3 + kbn) * 2 + 1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            HHX_MFMA(a[0], xh[kb], a0); HHX_MFMA(a[2], xh[kb], a1);
            HHX_MFMA(a[1], xh[kb], a0); HHX_MFMA(a[3], xh[kb], a1);
            HHX_MFMA(a[0], xl[kb], a0); HHX_MFMA(a[2], xl[kb], a1);
            if (npw == 8) hhw_issue_some(gsrc, gdst, st_ < 2 ? 2 * st_ : st_ + 2, st_ < 2 ? 2 : 1); /* 2, 2, 1, 1, 1, 1 */
            else if (st_ < npw) hhw_issue_some(gsrc, gdst, st_, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kb == 2) hhx_pair_to_frag(a0, a1, zh[c * 2 + tp], zl[c * 2 + tp]);
        }
    }

    /* ---- fight nets: y3 <- normalize(y3 + att_val(y3)) on hidden columns 0..149 = fragments 0..4; output tile j = half j & 1 of fragment j >> 1.  Five chunks
     *      of one tile pair each, steps of one k-block ---- */
    if constexpr (ATT) {
        hh_f32x4 y[10];
        float ssum = 0.0f;
#pragma unroll
        for (int c = 0; c < 5; c++, ck++) {
            __syncthreads();
            const int npw = c < 4 ? 5 : 8; /* next: attention tiles (20 pieces) | shared layer (0, 0) (32) */
            const unsigned char *gsrc = sp + (size_t)wave * npw * HHW_PIECE + lane * 16;
            unsigned char *gdst = buf[(ck + 1) & 1] + wave * npw * HHW_PIECE;
            sp += (size_t)(npw * 4) * HHW_PIECE;
            const unsigned char *cb = buf[ck & 1];
            /* piece of (tile jj of the chunk, k-block kb, plane) = (jj * 5 + kb) * 2 + plane */
            hh_f32x4 acc0 = hhx_bias_acc(bl + 1024 + 16 * (c * 2), g), acc1 = hhx_bias_acc(bl + 1024 + 16 * (c * 2 + 1), g);
            hh_h8 an[4];
            an[0] = hhw_frag(cb, 0, lane); an[1] = hhw_frag(cb, 1, lane); an[2] = hhw_frag(cb, 10, lane); an[3] = hhw_frag(cb, 11, lane);
#pragma unroll
            for (int kb = 0; kb < 5; kb++) {
                hh_h8 a[4];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] = an[u];
                hhw_need4(a);
                if (kb + 1 < 5) {
                    an[0] = hhw_frag(cb, (kb + 1) * 2, lane); an[1] = hhw_frag(cb, (kb + 1) * 2 + 1, lane);
                    an[2] = hhw_frag(cb, (5 + kb + 1) * 2, lane); an[3] = hhw_frag(cb, (5 + kb + 1) * 2 + 1, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                HHX_MFMA(a[0], zh[kb], acc0); HHX_MFMA(a[2], zh[kb], acc1);
                HHX_MFMA(a[1], zh[kb], acc0); HHX_MFMA(a[3], zh[kb], acc1);
                HHX_MFMA(a[0], zl[kb], acc0); HHX_MFMA(a[2], zl[kb], acc1);
                if (npw == 8) hhw_issue_some(gsrc, gdst, kb < 3 ? 2 * kb : kb + 3, kb < 3 ? 2 : 1); /* 2, 2, 2, 1, 1 */
                else hhw_issue_some(gsrc, gdst, kb, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int j = c * 2 + jj;
                const hh_f32x4 &acc = jj ? acc1 : acc0;
                const int f = j >> 1, e0 = 4 * (j & 1);
#pragma unroll
                for (int r = 0; r < 4; r++) { /* columns 150..159 are padding: zero first-layer outputs, zero weights, zero biases -> v = 0 */
                    const float x = (float)zh[f][e0 + r] + (float)zl[f][e0 + r];
                    const float v = x + acc[r];
                    y[j][r] = v;
                    ssum += v * v;
                }
            }
        }
        ssum += __shfl_xor(ssum, 16);
        ssum += __shfl_xor(ssum, 32);
        const float rden = hhx_recip_norm(ssum);
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int f = j >> 1, e0 = 4 * (j & 1);
            hhx_put4(zh[f], zl[f], e0 >> 1, hh_f2{y[j][0] * rden, y[j][1] * rden}, hh_f2{y[j][2] * rden, y[j][3] * rden});
        }
    }

    /* ---- the shared layer and val_out ---- */
    hh_f32x4 lacc[1];
    HhxProf pfc;
    HHX_T0(pfc);
    hhx_l2_l3<1, 4, 2>(zh, zl, ldsb, ck, st + (size_t)(HHXC_L1_PIECES + HHXC_ATT_PIECES) * HHW_PIECE,
                       st + (size_t)(HHXC_L1_PIECES + HHXC_ATT_PIECES + HHX_L2_PIECES) * HHW_PIECE, ldsb + HHXC_OFF_L3, bl, wave, lane, g, lacc, pfc);
    if (g == 0 && row >= 0) sa.vf[row] = lacc[0][0] + bl[1184]; /* output column 0 of the head's tile */
}

template <int WV> /* 4: 64-row tiles, two workgroups per CU | 8: 128-row tiles, one workgroup of eight waves per CU whose two waves per SIMD share ONE pass over the weights */
__global__ __launch_bounds__(64 * WV, 2) void hh_k_policy_w16(HhpBank bank, HhpBankX bankx, int n_nets, const float *__restrict__ obs, int obs_stride, int *counts,
                                                          const int *__restrict__ lists, int max_rows, int8_t *__restrict__ actions, float *__restrict__ logits_out,
                                                          int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    int net, tile, cnt;
    HhTl tl;
    hh_tl_begin(tl);
    /* grid-stride over the tiles: the host sizes the grid by the rows it EXPECTS to carry a network (hhp_launch_forward), not by the row buffer — a workgroup that
     * finds no tile still needs a whole CU (eight waves x 256 registers, 141 KB of LDS) to find that out, and a call whose grid covers every row slot of the buffer
     * (three times the rows a HighLevelEnv sub-step lists) ends only when all of them have had one: behind the other sub-worlds' tiles (tools/timeline.py).  A call
     * with more rows than expected is still complete: some workgroups take a second tile. */
    bool work = false;
#pragma nounroll
    for (int gt = (int)blockIdx.x; hhp_locate<16 * WV>(cn, gt, net, tile, cnt); gt += (int)gridDim.x) {
        if (work) __syncthreads(); /* the tile before is through with the logits / row ids that share the LDS */
        work = true;
        hhx_forward_tile<false, WV>(bank.net[net], bankx.stream[net], obs, obs_stride, lists + (size_t)net * max_rows, tile, cnt, actions, logits_out, ldsb);
    }
    hhp_consume_counts(counts, consume);
    if (threadIdx.x == 0) hh_tl_end(tl, work ? 2 : 3, (unsigned)(size_t)counts);
}

/* hh_policy_sample in this form: workgroup 2 t = actor tile t with the sampler's tail, workgroup 2 t + 1 = the value branch of the same 64 rows */
__global__ __launch_bounds__(256, 2) void hh_k_policy_w16_ppo(HhpBank bank, HhpBankX bankx, HhpCritBankX cbank, int n_nets, const float *__restrict__ obs,
                                                              int obs_stride, int *counts, const int *__restrict__ lists, int max_rows, HhpSampleArgs sa,
                                                              int with_critic, int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    const int gt = with_critic ? (int)blockIdx.x >> 1 : (int)blockIdx.x, crit = with_critic ? (int)blockIdx.x & 1 : 0;
    int net, tile, cnt;
    if (hhp_locate<64>(cn, gt, net, tile, cnt)) {
        const int *list = lists + (size_t)net * max_rows;
        if (crit) {
            if (cbank.c[net].has_att) hhx_critic_tile<true>(cbank.c[net], obs, obs_stride, list, tile, cnt, sa, ldsb);
            else hhx_critic_tile<false>(cbank.c[net], obs, obs_stride, list, tile, cnt, sa, ldsb);
        }
        else hhx_forward_tile<true, 4>(bank.net[net], bankx.stream[net], obs, obs_stride, list, tile, cnt, sa.actions, sa.logits_out, ldsb, &sa);
    }
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
