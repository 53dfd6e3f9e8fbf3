/*
 * hh_policy_kernel_ppo.h — the TRAINABLE policies of a PPO rollout in the fused policy kernel (C ABI: hh_policy_sample, hh_policy.h).
 *
 * What RLlib's sampler evaluates per env step for each policy of train_hetero.py:206-243: the actor (the forward of
 * hh_policy_kernel_h16.h), a Categorical draw per MultiDiscrete component with its log-probability, and the centralised value branch
 * of models/ac_models_hetero.py (Fight1 232-255 / 277-289, Fight2 344-367 / 390-402, Esc1 72-83 / 97-103, Esc2 148-159 / 173-179) on
 * central_critic_observer's rows (train_hetero.py:162-181).  The value branch has the actor's shape — input FCs (tanh) concatenated to
 * 500, the fight nets' last block through a length-1 MultiheadAttention + L2 normalisation, the SAME shared layer, a linear head — so it
 * runs as a second KIND OF TILE of one kernel: workgroup 2 t takes the actor tile t of the row lists, workgroup 2 t + 1 the critic tile
 * of the same 32 rows.  Split-fp16 contractions (three v_mfma_f32_32x32x16_f16 per 16 k-steps, fp32 accumulate), transposed C tiles and
 * the output layer contracted from registers exactly as in hhp_forward_tiles<1>; what differs per tile kind:
 *
 *                 input row (K)                                  L1 blocks            attention block          head
 *     actor       obs [<= 30]                  (K = 32)          200|200|100 or ...   columns 400..499, 100    26 | 24 logits -> draw, logp
 *     critic      own obs | own act | other obs | other act      fight: y3 150 first, columns 0..149, 150      1 value
 *                 (57 fight, 66 escape; K = 80)                  pad 10, v1 175, v2 175 (the shared layer's rows permuted to match)
 *
 * The critic's hidden columns are stored with the attention block FIRST (16-k aligned at column 0; the reference has it at 350); the
 * host permutes the shared layer's input rows of the critic's copy accordingly, so nothing is re-ordered on the device.
 */
#ifndef HH_POLICY_KERNEL_PPO_H
#define HH_POLICY_KERNEL_PPO_H

#define HHC_XK 80     /* padded critic input width (57 | 66) */
#define HHC_ATT_W 160 /* padded attention width of the value branch (150) */
#define HHC_ATT_J 256 /* ... as output columns: two 32-column tiles per wave */
#define HHC_V3_OFF 0  /* first hidden column of the y3 block */
#define HHC_V12_OFF 160

struct HhpCrit {
    const float4 *w1h, *w1l;   /* [5][2][512] fragments (8 halves each): [80 x 512] */
    const float4 *wovh, *wovl; /* [10][2][256]: [160 x 256], fight nets */
    const float4 *wsh, *wsl;   /* [32][2][512] shared layer, input rows in the critic's hidden order */
    const float4 *wah, *wal;   /* [32][2][32] val_out in output column 0, k order hhp_hidx_t */
    const float *b1, *bs, *bov, *ba; /* [512], [512], [160], [32] */
    int d1, a1, d2, a2, has_att, loaded;
};
struct HhpCritBank {
    HhpCrit c[HH_POLICY_MAX_NETS];
};

struct HhpSampleArgs {
    const double *uniforms; /* [n_rows, 4] or nullptr */
    const int4 *ar_pack;    /* the world's per-arena counters (steps, episode, ...) or nullptr */
    unsigned long long seed, arena_offset;
    int rows_per_arena;
    const float *crit_act;  /* [n_rows, 4] or nullptr */
    int greedy;
    int8_t *actions;
    float *logp, *vf, *logits_out;
};

/* LDS (bytes): Zh 32 KB | Zl 32 KB | Xh 5 KB | Xl 5 KB | rows 128 | norm partials 512 | biases (512 + 512 + 160 floats); the L3 partials
 * (36 KB) and the logits (4 KB at byte 40960) alias the activation tile once S is dead.  2 x 81152 B = 158.5 KB: two workgroups per CU. */
#define HHPP_OFF_ZL 32768
#define HHPP_OFF_XH 65536
#define HHPP_OFF_XL (65536 + 5120)
#define HHPP_OFF_ROWS (65536 + 10240)
#define HHPP_OFF_NP (HHPP_OFF_ROWS + 128)
#define HHPP_OFF_BIAS (HHPP_OFF_NP + 512)
#define HHPP_LDS_BYTES (HHPP_OFF_BIAS + (512 + 512 + 160) * 4)
#define HHPP_OFF_LG 40960

/* x <- normalize(x + Wov x + bov) on the W-wide block that starts at hidden column c0 (16-k aligned), every wave NT column tiles
 * (F.normalize(x_full + att(x_full)), ac_models_hetero.py:270 / 280): see the attention section of hhp_forward_tiles */
template <int NT, int KB>
__device__ __forceinline__ void hhp_att_block(_Float16 *__restrict__ Zh, _Float16 *__restrict__ Zl, float *__restrict__ npart,
                                              const float *__restrict__ bov, const float4 *__restrict__ wovh, const float4 *__restrict__ wovl,
                                              int J, int c0, int width, int wave, int lane) {
    constexpr int R = 32;
    const int ci = lane & 31, g = lane >> 5, row = ci;
    hh_f32x16 acc[1][NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[0][t] = hhp_zero16();
    if constexpr (KB <= 8) hhp_gemm_h_short<NT, KB, 1, R>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), c0 >> 4, 0, wovh, wovl, 0, J, wave * 32 * NT, lane, acc);
    else { /* two halves, each with its weight fragments requested up front */
        constexpr int K1 = KB / 2, K2 = KB - K1;
        hhp_gemm_h_short<NT, K1, 1, R>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), c0 >> 4, 0, wovh, wovl, 0, J, wave * 32 * NT, lane, acc);
        hhp_gemm_h_short<NT, K2, 1, R>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), (c0 >> 4) + K1, 0, wovh, wovl, K1, J, wave * 32 * NT, lane, acc);
    }
    float y[NT][16];
    float ssum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jb = wave * 32 * NT + t * 32 + 8 * q + 4 * g; /* first of this group's four columns; width % 4 == 0 */
            const bool ok = jb < width;
            const int col = c0 + (ok ? jb : 0), plane = col >> 3;
            const int off = (plane * R + (row ^ (plane & 7))) * 8 + 4 * g;
            const hh_h4 xh = *reinterpret_cast<const hh_h4 *>(Zh + off), xl = *reinterpret_cast<const hh_h4 *>(Zl + off);
            const float4 b = *reinterpret_cast<const float4 *>(bov + (ok ? jb : 0));
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x = (float)xh[i] + (float)xl[i];
                const float v = ok ? x + (acc[0][t][4 * q + i] + bb[i]) : 0.0f;
                y[t][4 * q + i] = v;
                ssum += v * v;
            }
        }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(ssum), __float_as_int(ssum), false, false);
        const float other = __int_as_float(g ? sw[0] : sw[1]);
        const float tot = g ? other + ssum : ssum + other;
        if (!g) npart[wave * R + row] = tot;
    }
    __syncthreads();
    const float nn = ((npart[row] + npart[R + row]) + npart[2 * R + row]) + npart[3 * R + row];
    const float den = fmaxf(sqrtf(nn), 1e-12f);
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jb = wave * 32 * NT + t * 32 + 8 * q + 4 * g;
            if (jb < width) {
                const int col = c0 + jb, plane = col >> 3;
                const int off = (plane * R + (row ^ (plane & 7))) * 8 + 4 * g;
                hh_h4 h, l;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float v = y[t][4 * q + i] / den;
                    h[i] = (_Float16)v;
                    l[i] = (_Float16)(v - (float)h[i]);
                }
                *reinterpret_cast<hh_h4 *>(Zh + off) = h;
                *reinterpret_cast<hh_h4 *>(Zl + off) = l;
            }
        }
    __syncthreads();
}

/* one 32-row tile of one network, actor (CRIT = false) or value branch (CRIT = true) */
template <bool CRIT>
__device__ __forceinline__ void hhp_ppo_tile(const HhpNet &N, const HhpNetH &H, const HhpCrit &Cw, const float *__restrict__ obs, int obs_stride,
                                             const int *__restrict__ list, int tile, int cnt, const HhpSampleArgs &sa, unsigned char *ldsb) {
    constexpr int R = 32, NTH = 256, NT = 4, WC = 128, NP = 2, PZS = 36;
    _Float16 *Zh = reinterpret_cast<_Float16 *>(ldsb);
    _Float16 *Zl = reinterpret_cast<_Float16 *>(ldsb + HHPP_OFF_ZL);
    _Float16 *Xh = reinterpret_cast<_Float16 *>(ldsb + HHPP_OFF_XH);
    _Float16 *Xl = reinterpret_cast<_Float16 *>(ldsb + HHPP_OFF_XL);
    float *Pz = reinterpret_cast<float *>(ldsb);
    float *Lg = reinterpret_cast<float *>(ldsb + HHPP_OFF_LG);
    int *rows = reinterpret_cast<int *>(ldsb + HHPP_OFF_ROWS);
    float *npart = reinterpret_cast<float *>(ldsb + HHPP_OFF_NP);
    float *bl = reinterpret_cast<float *>(ldsb + HHPP_OFF_BIAS);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 31, g = lane >> 5;
    const bool has_att = CRIT ? Cw.has_att != 0 : N.has_att != 0;
    {
        const float *b1 = CRIT ? Cw.b1 : N.b1, *bs = CRIT ? Cw.bs : N.bs, *bov = CRIT ? Cw.bov : N.bov;
        for (int e = tid; e < 512; e += NTH) { bl[e] = b1[e]; bl[512 + e] = bs[e]; }
        if (has_att && tid < (CRIT ? HHC_ATT_W : HHP_ATT_J)) bl[1024 + tid] = bov[tid];
    }
    const float bar = (CRIT ? Cw.ba : N.ba)[ci];
    HHP_T0;
#define HHPP_T(k) HHP_T(16 + (CRIT ? 16 : 0) + (k))
    if (tid < R) {
        const int q = tile * R + tid;
        rows[tid] = q < cnt ? list[q] : -1;
    }
    __syncthreads();
    /* the input tile: every thread's loads are REQUESTED before the first is used (clamped addresses instead of predicated loads: a
     * load per loop trip with its own wait cost 14.5 k cycles per critic tile, 16 % of it) */
    if constexpr (!CRIT) {
        const int od = N.obs_dim;
        float v[R * HHP_XK / NTH];
#pragma unroll
        for (int u = 0; u < R * HHP_XK / NTH; u++) {
            const int e = tid + u * NTH, i = e >> 5, c = e & 31, r = rows[i];
            const bool ok = r >= 0 && c < od;
            const float x = obs[ok ? (size_t)r * obs_stride + c : 0];
            v[u] = ok ? x : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < R * HHP_XK / NTH; u++) {
            const int e = tid + u * NTH;
            hhp_split_store<R>(Xh, Xl, hhp_haidx<R>(e & 31, e >> 5), v[u]);
        }
    } else {
        /* central_critic_observer's row (train_hetero.py:162-181), in the order the value branch concatenates it: own observation, own
         * action, the other agent's observation, the other agent's action; the other agent of row r is row r ^ 1 */
        const int e1 = Cw.d1, e2 = e1 + Cw.a1, e3 = e2 + Cw.d2, e4 = e3 + Cw.a2;
        const float *ca = sa.crit_act;
        float v[R * HHC_XK / NTH];
#pragma unroll
        for (int u = 0; u < R * HHC_XK / NTH; u++) {
            const int e = tid + u * NTH, i = e / HHC_XK, c = e - i * HHC_XK, r = rows[i];
            const bool own = c < e2, is_obs = c < e1 || (c >= e2 && c < e3);
            const int cc = c < e1 ? c : (c < e2 ? c - e1 : (c < e3 ? c - e2 : c - e3));
            const bool ok = r >= 0 && c < e4 && (is_obs || ca != nullptr);
            const size_t rr = (size_t)(own ? r : r ^ 1);
            const float *src = is_obs ? obs + rr * obs_stride + cc : ca + rr * 4 + cc;
            const float x = *(ok ? src : obs);
            v[u] = ok ? x : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < R * HHC_XK / NTH; u++) {
            const int e = tid + u * NTH, i = e / HHC_XK;
            hhp_split_store<R>(Xh, Xl, hhp_haidx<R>(e - i * HHC_XK, i), v[u]);
        }
    }
    __syncthreads();
    HHPP_T(0);

    /* ---- L1 ---- */
    {
        hh_f32x16 acc[1][NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[0][t] = hhp_zero16();
        if constexpr (!CRIT) hhp_gemm_h_short<NT, HHP_XK / 16, 1, R>(reinterpret_cast<const float4 *>(Xh), reinterpret_cast<const float4 *>(Xl), 0, 0, H.w1h, H.w1l, 0, HHP_H, wave * WC, lane, acc);
        else { /* K = 80 as 3 + 2 blocks with all of a part's weight fragments requested up front (96 + 64 registers): two L2 round trips */
            hhp_gemm_h_short<NT, 3, 1, R>(reinterpret_cast<const float4 *>(Xh), reinterpret_cast<const float4 *>(Xl), 0, 0, Cw.w1h, Cw.w1l, 0, HHP_H, wave * WC, lane, acc);
            hhp_gemm_h_short<NT, 2, 1, R>(reinterpret_cast<const float4 *>(Xh), reinterpret_cast<const float4 *>(Xl), 3, 0, Cw.w1h, Cw.w1l, 3, HHP_H, wave * WC, lane, acc);
        }
        HHPP_T(1);
#pragma unroll
        for (int t = 0; t < NT; t++)
            hhp_store_tile_t<R>(Zh, Zl, wave * WC + t * 32, ci, g, acc[0][t], bl, [](hh_f2 a) { return hhp_tanh2(a); });
    }
    __syncthreads();
    HHPP_T(2);

    /* ---- the attention block of the fight nets ---- */
    if (has_att) {
        if constexpr (!CRIT) hhp_att_block<1, 7>(Zh, Zl, npart, bl + 1024, H.wovh, H.wovl, HHP_ATT_J, 400, 100, wave, lane);
        else hhp_att_block<2, 10>(Zh, Zl, npart, bl + 1024, Cw.wovh, Cw.wovl, HHC_ATT_J, HHC_V3_OFF, HHC_ATT_W, wave, lane);
    }
    HHPP_T(3);

    /* ---- L2 (the shared layer) and the head straight from its registers ---- */
    {
        const float4 *wsh = CRIT ? Cw.wsh : H.wsh, *wsl = CRIT ? Cw.wsl : H.wsl, *wah = CRIT ? Cw.wah : H.wah, *wal = CRIT ? Cw.wal : H.wal;
        hh_f32x16 acc[1][NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[0][t] = hhp_zero16();
        hhp_gemm_h<NT, 1>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), 0, HHP_H / 16, wsh, wsl, 0, HHP_H, wave * WC, lane, acc);
        HHPP_T(4);
        float4 wfh[NT][2], wfl[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int idx = ((((wave * WC + t * 32) >> 4) + b) * 2 + g) * HHP_OUT + ci;
                wfh[t][b] = wah[idx]; wfl[t][b] = wal[idx];
            }
        hh_f32x16 lacc[NP];
#pragma unroll
        for (int pp = 0; pp < NP; pp++) lacc[pp] = hhp_zero16();
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 b = *reinterpret_cast<const float4 *>(bl + 512 + wave * WC + t * 32 + 8 * q + 4 * g);
                const hh_f2 p0 = hhp_tanh2(hh_f2{acc[0][t][4 * q + 0], acc[0][t][4 * q + 1]} + hh_f2{b.x, b.y});
                const hh_f2 p1 = hhp_tanh2(hh_f2{acc[0][t][4 * q + 2], acc[0][t][4 * q + 3]} + hh_f2{b.z, b.w});
                v[4 * q + 0] = p0.x; v[4 * q + 1] = p0.y; v[4 * q + 2] = p1.x; v[4 * q + 3] = p1.y;
            }
#pragma unroll
            for (int b = 0; b < 2; b++) {
                hh_h8 xh, xl;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    xh[e] = (_Float16)v[8 * b + e];
                    xl[e] = (_Float16)(v[8 * b + e] - (float)xh[e]);
                }
                lacc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfh[t][b]), xh, lacc[t >> 1], 0, 0, 0);
                lacc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfl[t][b]), xh, lacc[t >> 1], 0, 0, 0);
                lacc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfh[t][b]), xl, lacc[t >> 1], 0, 0, 0);
            }
        }
        HHPP_T(5);
        __syncthreads(); /* every wave is done reading Z: the partials go there */
        HHPP_T(6);
#pragma unroll
        for (int pp = 0; pp < NP; pp++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4 *>(Pz + (wave * NP + pp) * (R * PZS) + ci * PZS + 8 * q + 4 * g) =
                    make_float4(lacc[pp][4 * q], lacc[pp][4 * q + 1], lacc[pp][4 * q + 2], lacc[pp][4 * q + 3]);
    }
    __syncthreads();
    HHPP_T(7);
    for (int e = tid; e < R * HHP_OUT; e += NTH) {
        const int i = e >> 5, c = e & 31, pe = i * PZS + c;
        float v = Pz[pe];
#pragma unroll
        for (int w_ = 1; w_ < 8; w_++) v += Pz[w_ * (R * PZS) + pe]; /* the same eight 64-column ranges in the same order as hhp_forward_tiles */
        v += bar;
        Lg[e] = v;
        if constexpr (!CRIT) {
            if (sa.logits_out && rows[i] >= 0) sa.logits_out[(size_t)rows[i] * HH_POLICY_LOGITS + c] = c < N.n_out ? v : 0.0f;
        }
    }
    __syncthreads();
    if constexpr (CRIT) {
        if (tid < R && rows[tid] >= 0) sa.vf[rows[tid]] = Lg[tid * 32];
    } else {
        /* TorchMultiCategorical over [13, 9, 2, 2] (| [13, 9, 2]): one thread per (row, component) */
        if (tid < R * 4) {
            const int row = tid >> 2, k = tid & 3, r = rows[row];
            const int lo = k == 0 ? 0 : (k == 1 ? 13 : (k == 2 ? 22 : 24)), hi = k == 0 ? 13 : (k == 1 ? 22 : (k == 2 ? 24 : 26));
            const float *lg = Lg + row * 32;
            int a = 0;
            float lp = 0.0f;
            if (k < (N.n_out == 26 ? 4 : 3)) {
                float m = lg[lo];
                int best = lo;
                for (int c = lo + 1; c < hi; c++) if (lg[c] > m) { m = lg[c]; best = c; }
                float S = 0.0f;
                for (int c = lo; c < hi; c++) S += __expf(lg[c] - m);
                a = best - lo;
                if (!sa.greedy && r >= 0) {
                    double u;
                    if (sa.uniforms) u = sa.uniforms[(size_t)r * 4 + k];
                    else {
                        const int n = r / sa.rows_per_arena, s = r - n * sa.rows_per_arena;
                        const int4 ap = sa.ar_pack[n];
                        u = hh_rng_u01(hh_rng_tick_key(hh_rng_arena_key(sa.seed, sa.arena_offset + (unsigned long long)n), (uint32_t)ap.y, (uint32_t)ap.x),
                                       (uint32_t)(s + 1), HH_SITE_POLICY_SAMPLE, (uint32_t)k);
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
                reinterpret_cast<int *>(sa.actions)[r] = aw;
                if (sa.logp) sa.logp[r] = lp;
            }
        }
    }
    HHPP_T(8);
#undef HHPP_T
}

__global__ __launch_bounds__(256, 2) void hh_k_policy_ppo(HhpBank bank, HhpBankH bankh, HhpCritBank cbank, int n_nets, const float *__restrict__ obs,
                                                          int obs_stride, int *counts, const int *__restrict__ lists, int max_rows, HhpSampleArgs sa,
                                                          int with_critic, int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    const int gt = with_critic ? (int)blockIdx.x >> 1 : (int)blockIdx.x, crit = with_critic ? (int)blockIdx.x & 1 : 0;
    int net, tile, cnt;
    if (hhp_locate<32>(cn, gt, net, tile, cnt)) {
        const int *list = lists + (size_t)net * max_rows;
        if (crit) hhp_ppo_tile<true>(bank.net[net], bankh.net[net], cbank.c[net], obs, obs_stride, list, tile, cnt, sa, ldsb);
        else hhp_ppo_tile<false>(bank.net[net], bankh.net[net], cbank.c[net], obs, obs_stride, list, tile, cnt, sa, ldsb);
    }
    hhp_consume_counts(counts, consume);
}

#endif /* HH_POLICY_KERNEL_PPO_H */
