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
 * Round 3: every contraction runs TRANSPOSED — the weight fragment is the MFMA's A operand, the activation fragment its B operand (both
 * are "32 vectors of 8 halves", so this is only the order of the two arguments) — which leaves a lane of the C tile with one ROW and four
 * consecutive COLUMNS per register group: 8-byte packed epilogue stores, broadcast float4 biases from an LDS copy, and the output layer
 * contracted straight from the shared layer's registers (see hhp_store_tile_t and the L2 / L3 section of hhp_forward_tiles).
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
/* the output layer's weights for L3 FROM REGISTERS: the B operand there is built from the shared layer's transposed C tile, where lane
 * group g holds columns 16 b + 4 g + (0..3) and 16 b + 8 + 4 g + (0..3) of k-block b — the A fragments follow the same k order */
__host__ __device__ inline size_t hhp_hidx_t(int k, int col, int J) {
    const int kb = k >> 4, w = k & 15, g = (w >> 2) & 1, e = ((w >> 3) << 2) | (w & 3);
    return ((size_t)(kb * 2 + g) * J + col) * 8 + e;
}
/* the same for the R-row LDS activation planes (R = 32 or 64 rows per tile), row slot XOR-swizzled by the plane like hhp_aidx (the
 * XOR touches the low three bits only) */
template <int R>
__device__ __forceinline__ int hhp_haidx(int k, int row) {
    const int plane = (k >> 4) * 2 + ((k >> 3) & 1);
    return (plane * R + (row ^ (plane & 7))) * 8 + (k & 7);
}
template <int R>
__device__ __forceinline__ void hhp_split_store(_Float16 *__restrict__ hi, _Float16 *__restrict__ lo, int idx, float v) {
    const _Float16 h = (_Float16)v;
    hi[idx] = h;
    lo[idx] = (_Float16)(v - (float)h);
}
/* ---- round 3: the contractions run TRANSPOSED (weights as the A operand, activations as B), so a lane of the C tile holds ONE ROW
 * (lane & 31) and, per register group q = r >> 2, FOUR CONSECUTIVE COLUMNS 8 q + 4 (lane >> 5) + (r & 3): exactly the four halves that
 * are contiguous in the [plane][row][8] operand layout.  An epilogue therefore writes a tile with four 8-byte stores per plane instead of
 * sixteen 2-byte ones, the bias of a column comes as one broadcast float4 per group from a copy staged in LDS, and the split pairs up
 * for v_cvt_pk_f16_f32.  The products and their accumulation order are those of the untransposed form. */
typedef _Float16 hh_h4 __attribute__((ext_vector_type(4)));
template <int R, class F>
__device__ __forceinline__ void hhp_store_tile_t(_Float16 *__restrict__ hi_plane, _Float16 *__restrict__ lo_plane, int j0, int row, int g,
                                                 const hh_f32x16 &acc, const float *__restrict__ bias, F f) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int plane = (j0 >> 3) + q;
        const int off = (plane * R + (row ^ (plane & 7))) * 8 + 4 * g; /* halves; 8-byte aligned */
        float4 b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (bias) b = *reinterpret_cast<const float4 *>(bias + j0 + 8 * q + 4 * g);
        const hh_f2 p0 = f(hh_f2{acc[4 * q + 0], acc[4 * q + 1]} + hh_f2{b.x, b.y}), p1 = f(hh_f2{acc[4 * q + 2], acc[4 * q + 3]} + hh_f2{b.z, b.w});
        const float v0 = p0.x, v1 = p0.y, v2 = p1.x, v3 = p1.y;
        hh_h4 h, l;
        h[0] = (_Float16)v0; h[1] = (_Float16)v1; h[2] = (_Float16)v2; h[3] = (_Float16)v3;
        l[0] = (_Float16)(v0 - (float)h[0]); l[1] = (_Float16)(v1 - (float)h[1]); l[2] = (_Float16)(v2 - (float)h[2]); l[3] = (_Float16)(v3 - (float)h[3]);
        *reinterpret_cast<hh_h4 *>(hi_plane + off) = h;
        *reinterpret_cast<hh_h4 *>(lo_plane + off) = l;
    }
}

__device__ __forceinline__ hh_h8 hhp_as_h8(const float4 &v) {
    union { float4 f; hh_h8 h; } u;
    u.f = v;
    return u.h;
}

/* NT column tiles x RH row halves over KB 16-k blocks: acc += Ahi Bhi + Ahi Blo + Alo Bhi.  A fragments from the LDS planes, B
 * fragments from global: with RH = 2 a weight fragment serves two row halves from registers — half the weight stream per row.
 * Three register sets of weight fragments rotate BY NAME (the block loop is unrolled by three): block kb computes from set kb % 3
 * while the loads of block kb + 2 land in set (kb + 2) % 3.  Rotating by register moves instead would read the set that was just
 * requested and make the wave wait for it — a prefetch distance of one block where an L2 round trip under load is longer
 * (measured: 37 % of the wave cycles waiting). */
template <int NT>
struct HhpBSet {
    float4 h[NT], l[NT];
};
template <int NT, int RH>
__device__ __forceinline__ void hhp_gemm_h(const float4 *__restrict__ a_hi, const float4 *__restrict__ a_lo, int kb0, int KB,
                                           const float4 *__restrict__ b_hi, const float4 *__restrict__ b_lo, int bkb0, int J, int j0, int lane,
                                           hh_f32x16 (&acc)[RH][NT]) {
    constexpr int R = 32 * RH;
    const int h = lane >> 5, i = lane & 31;
    const unsigned bstep = 2u * (unsigned)J;
    unsigned boff = (unsigned)((bkb0 * 2 + h) * J + j0 + i); /* fragment index from the (wave-uniform) plane bases: one 32-bit add per block */
    HhpBSet<NT> S[3];
#pragma unroll
    for (int t = 0; t < NT; t++) { S[0].h[t] = b_hi[boff + t * 32]; S[0].l[t] = b_lo[boff + t * 32]; S[1].h[t] = S[0].h[t]; S[1].l[t] = S[0].l[t]; S[2].h[t] = S[0].h[t]; S[2].l[t] = S[0].l[t]; }
    if (KB > 1) {
#pragma unroll
        for (int t = 0; t < NT; t++) { S[1].h[t] = b_hi[boff + bstep + t * 32]; S[1].l[t] = b_lo[boff + bstep + t * 32]; }
    }
    boff += 2u * bstep; /* block kb + 2 */
    int p0 = kb0 * 2 + h;
    float4 ah[RH], al[RH];
#pragma unroll
    for (int f = 0; f < RH; f++) { ah[f] = a_hi[p0 * R + f * 32 + (i ^ (p0 & 7))]; al[f] = a_lo[p0 * R + f * 32 + (i ^ (p0 & 7))]; }
#pragma nounroll
    for (int base = 0; base < KB; base += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int kb = base + u;
            if (kb < KB) { /* wave-uniform */
                HhpBSet<NT> &cur = S[u], &far = S[(u + 2) % 3];
#ifndef HHP_ABL_NO_BLOAD /* tuning builds only: the shared-layer loop without its weight loads / without its MFMAs (what each costs) */
                if (kb + 2 < KB) {
#pragma unroll
                    for (int t = 0; t < NT; t++) { far.h[t] = b_hi[boff + t * 32]; far.l[t] = b_lo[boff + t * 32]; }
                }
#endif
                boff += bstep;
                float4 ahn[RH], aln[RH];
#pragma unroll
                for (int f = 0; f < RH; f++) { ahn[f] = ah[f]; aln[f] = al[f]; }
                if (kb + 1 < KB) {
                    const int p = (kb0 + kb + 1) * 2 + h;
#pragma unroll
                    for (int f = 0; f < RH; f++) { ahn[f] = a_hi[p * R + f * 32 + (i ^ (p & 7))]; aln[f] = a_lo[p * R + f * 32 + (i ^ (p & 7))]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef HHP_ABL_NO_MFMA
#pragma unroll
                for (int f = 0; f < RH; f++)
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(cur.h[t]), hhp_as_h8(ah[f]), acc[f][t], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < RH; f++)
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(cur.l[t]), hhp_as_h8(ah[f]), acc[f][t], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < RH; f++)
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(cur.h[t]), hhp_as_h8(al[f]), acc[f][t], 0, 0, 0);
#else
                for (int f = 0; f < RH; f++) for (int t = 0; t < NT; t++) { acc[f][t][0] += cur.h[t].x + cur.l[t].y + ah[f].x + al[f].y; }
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < RH; f++) { ah[f] = ahn[f]; al[f] = aln[f]; }
            }
        }
    }
}

/* The short contractions (L1: K = 32, attention: K = 112, L3: K = 128 per wave) have too few MFMAs per block to hide a weight
 * fetch behind: all their weight fragments are requested up front (KB <= 8 blocks: at most 64 registers), so the wave pays ONE L2
 * round trip instead of one per block (phase timers: attention 21 %, L3 10 %, L1 8 % of a tile before this). */
template <int NT, int KB, int RH, int R>
__device__ __forceinline__ void hhp_gemm_h_short(const float4 *__restrict__ a_hi, const float4 *__restrict__ a_lo, int kb0, int row0,
                                                 const float4 *__restrict__ b_hi, const float4 *__restrict__ b_lo, int bkb0, int J, int j0, int lane,
                                                 hh_f32x16 (&acc)[RH][NT]) {
    const int h = lane >> 5, i = lane & 31;
    const unsigned bstep = 2u * (unsigned)J, boff = (unsigned)((bkb0 * 2 + h) * J + j0 + i);
    float4 bh[KB][NT], bl[KB][NT];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
#pragma unroll
        for (int t = 0; t < NT; t++) { bh[kb][t] = b_hi[boff + kb * bstep + t * 32]; bl[kb][t] = b_lo[boff + kb * bstep + t * 32]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const int p = (kb0 + kb) * 2 + h;
        float4 ah[RH], al[RH];
#pragma unroll
        for (int f = 0; f < RH; f++) { ah[f] = a_hi[p * R + row0 + f * 32 + (i ^ (p & 7))]; al[f] = a_lo[p * R + row0 + f * 32 + (i ^ (p & 7))]; }
#pragma unroll
        for (int f = 0; f < RH; f++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(bh[kb][t]), hhp_as_h8(ah[f]), acc[f][t], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < RH; f++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(bl[kb][t]), hhp_as_h8(ah[f]), acc[f][t], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < RH; f++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(bh[kb][t]), hhp_as_h8(al[f]), acc[f][t], 0, 0, 0);
    }
}

/* tuning builds only (-DHHP_PROFILE): s_memtime deltas per phase of wave 0 of every tile, summed into hhp_prof[] */
#ifdef HHP_PROFILE
__device__ unsigned long long hhp_prof[48]; /* 0..15 hhp_forward_tiles, 16..31 / 32..47 actor / critic tiles of hh_k_policy_ppo */
#define HHP_T(k) do { if (tid == 0) { unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&hhp_prof[k], t_ - pt_); pt_ = t_; } } while (0)
#define HHP_T0 unsigned long long pt_ = __builtin_readcyclecounter()
#else
#define HHP_T(k)
#define HHP_T0
#endif

/* LDS (bytes, x RH): Zh 32 KB | Zl 32 KB | Xh 2 KB | Xl 2 KB | L3 partials 16 KB alias Zh | logits 4 KB | rows x 2, norm partials */
#define HHPH_OFF_ZL(RH) (32768 * (RH))
#define HHPH_OFF_XH(RH) (65536 * (RH))
#define HHPH_OFF_XL(RH) (67584 * (RH))
#define HHPH_OFF_LG(RH) (69632 * (RH))
#define HHPH_OFF_ROWS(RH) (73728 * (RH))
#define HHPH_OFF_NP(RH) (73984 * (RH))
#define HHPH_OFF_BIAS(RH) ((73984 + 512) * (RH)) /* b1[512] | bs[512] | bov[128] of the tile's network (floats) */
#define HHPH_BIAS_FLOATS 1152
#define HHPH_LDS_BYTES(RH) ((73984 + 512) * (RH) + 4864)

/* global tile index -> (network, tile of that network, rows of that network); false past the last tile */
template <int R>
__device__ __forceinline__ bool hhp_locate(const int (&cn)[HH_POLICY_MAX_NETS], int gt, int &net, int &tile, int &cnt) {
    net = -1; tile = gt; cnt = 0;
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) {
        const int nt = (cn[n] + R - 1) / R;
        if (net < 0) {
            if (tile < nt) { net = n; cnt = cn[n]; }
            else tile -= nt;
        }
    }
    return net >= 0;
}

/* RH row halves per tile, 4 RH waves per workgroup.  RH = 1 (default): 32-row tiles, 256 threads, two workgroups per CU.
 * RH = 2 (HH_POLICY_TILE=64): 64-row tiles, 512 threads = two waves per SIMD, 145 KB of LDS, one workgroup per CU: in L1 / L2 a
 * wave owns 64 columns (2 MFMA tiles) of BOTH row halves, so every weight fragment it fetches serves two MFMA row blocks — half the
 * weight stream per row; the attention block and the split-K logits give each wave one (column or k quarter, row half) pair.
 * Persistent over tiles (grid-stride): the next tile's row list and observation rows are fetched while the current tile is in its
 * later layers, so a tile starts with its X operand already in LDS.  Measured (`tools/policy_bench.py`, Fight1 + Fight2 rows): 84 us
 * against 88 us per 32768 rows (back-to-back calls), 122 against 127 at 49152 — but workgroups of 145 KB come in whole rounds of
 * 64 n_cu rows (24576 rows: 78 us against 70), the row count that carries a network is known on the device only (launching both
 * widths and letting the binned counts choose costs the 2 us it gains), and `bench.py --workload rollout` / `hier --pilot net` come
 * out level.  Kept as an A/B instance; what separates both forms from the matrix pipe's 30 k cycles per 64 rows is that a tile's
 * epilogues (tanh, hi/lo split, 2-byte LDS scatter) and GEMMs alternate instead of overlapping. */
/* the forward over the tiles first_tile, first_tile + tile_stride, ... of the row lists (cn = rows per network): the body of
 * hh_k_policy_h */
template <int RH>
__device__ __forceinline__ void hhp_forward_tiles(const HhpBank &bank, const HhpBankH &bankh, const int (&cn)[HH_POLICY_MAX_NETS], const float *__restrict__ obs,
                                                  int obs_stride, const int *__restrict__ lists, int max_rows, int8_t *__restrict__ actions,
                                                  float *__restrict__ logits_out, unsigned char *ldsb, int first_tile, int tile_stride) {
    constexpr int R = 32 * RH, NTH = 256 * RH, NT = 4 / RH, WC = 32 * NT; /* rows per tile, threads, column tiles and columns per wave in L1 / L2 */
    constexpr int XPT = R * HHP_XK / NTH;                                 /* observation elements per thread (8 / 4) */
    _Float16 *Zh = reinterpret_cast<_Float16 *>(ldsb);                      /* [32][2][R][8] hi plane of the activation tile */
    _Float16 *Zl = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_ZL(RH));    /* lo plane */
    _Float16 *Xh = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_XH(RH));    /* [2][2][R][8] observation tile */
    _Float16 *Xl = reinterpret_cast<_Float16 *>(ldsb + HHPH_OFF_XL(RH));
    float *Pz = reinterpret_cast<float *>(ldsb);                            /* L3 split-K partials [4][R][32] (after S is dead) */
    float *Lg = reinterpret_cast<float *>(ldsb + HHPH_OFF_LG(RH));          /* [R][32] logits */
    int *rowsb = reinterpret_cast<int *>(ldsb + HHPH_OFF_ROWS(RH));         /* [2][R] row lists of this tile and the next */
    float *npart = reinterpret_cast<float *>(ldsb + HHPH_OFF_NP(RH));       /* [4][R] */
    float *bl = reinterpret_cast<float *>(ldsb + HHPH_OFF_BIAS(RH));        /* the network's biases by column */
    constexpr int PZS = 36;                                                 /* row stride of the L3 partials (floats): 16-byte rows, 4-way banks at worst */

    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wq = wave & 3, wrow0 = (wave >> 2) * 32; /* attention / L3: column or k quarter, row half of this wave */
    int gt = first_tile, net, tile, cnt;
    if (!hhp_locate<R>(cn, gt, net, tile, cnt)) return;
    HHP_T0;
#define HHP_STAGE_BIAS(NETV)                                                                          \
    do {                                                                                              \
        const HhpNet &sn_ = bank.net[NETV];                                                           \
        for (int e_ = tid; e_ < 512; e_ += NTH) { bl[e_] = sn_.b1[e_]; bl[512 + e_] = sn_.bs[e_]; }   \
        if (sn_.has_att) for (int e_ = tid; e_ < 128; e_ += NTH) bl[1024 + e_] = sn_.bov[e_];         \
    } while (0)
    { /* first tile: rows and observation in the open; later tiles find both prefetched */
        const int tid = tid0;
        HHP_STAGE_BIAS(net);
        if (tid < R) {
            const int q = tile * R + tid;
            rowsb[tid] = q < cnt ? lists[(size_t)net * max_rows + q] : -1;
        }
        __syncthreads();
        const int od = bank.net[net].obs_dim;
        for (int e = tid; e < R * HHP_XK; e += NTH) {
            const int i = e >> 5, c = e & 31, r = rowsb[i];
            hhp_split_store<R>(Xh, Xl, hhp_haidx<R>(c, i), (r >= 0 && c < od) ? obs[(size_t)r * obs_stride + c] : 0.0f);
        }
        __syncthreads();
        HHP_T(0);
    }

    for (int it = 0;; it++) {
        /* the lane index is laundered once per tile: otherwise the compiler hoists every lane-dependent LDS address of the body out of
         * the tile loop (~150 loop-invariant registers, most of them spilled) */
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int tid = wave * 64 + lane, ci = lane & 31;
        int *rows = rowsb + (it & 1) * R, *rows_next = rowsb + ((it + 1) & 1) * R;
        const HhpNet N = bank.net[net];
        const HhpNetH H = bankh.net[net];
        /* every bias this lane will add, requested now: a global round trip in front of each epilogue is ~1.4 k cycles */
        const float bar = N.ba[ci];
        const int g = lane >> 5; /* which four of a register group's eight columns this lane holds */
        /* the next tile of this workgroup: its row list is requested here and parked in LDS behind the L1 barrier */
        int nnet, ntile, ncnt, nrow = -1;
        const bool more = hhp_locate<R>(cn, gt + tile_stride, nnet, ntile, ncnt);
        if (more && tid < R) {
            const int q = ntile * R + tid;
            nrow = q < ncnt ? lists[(size_t)nnet * max_rows + q] : -1;
        }

        /* ---- L1 ---- */
        {
            hh_f32x16 acc[RH][NT];
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[f][t] = hhp_zero16();
            hhp_gemm_h_short<NT, HHP_XK / 16, RH, R>(reinterpret_cast<const float4 *>(Xh), reinterpret_cast<const float4 *>(Xl), 0, 0, H.w1h, H.w1l, 0, HHP_H, wave * WC, lane, acc);
            HHP_T(1);
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    hhp_store_tile_t<R>(Zh, Zl, wave * WC + t * 32, f * 32 + ci, g, acc[f][t], bl, [](hh_f2 a) { return hhp_tanh2(a); });
        }
        if (more && tid < R) rows_next[tid] = nrow;
        __syncthreads();
        HHP_T(2);

        /* ---- fight nets: x <- normalize(x + Wov x + bov) on columns 400..499 (K = 112: blocks 25..31 of the tile) ---- */
        if (N.has_att) {
            hh_f32x16 acc[1][1];
            acc[0][0] = hhp_zero16();
            hhp_gemm_h_short<1, 7, 1, R>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), 25, wrow0, H.wovh, H.wovl, 0, HHP_ATT_J, wq * 32, lane, acc);
            HHP_T(9);
            const int row = wrow0 + ci;
            float y[16];
            float ssum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int jb = wq * 32 + 8 * q + 4 * g; /* first of this group's four columns; 100 % 4 == 0: valid or not as a whole */
                const bool ok = jb < 100;
                const int col = 400 + (ok ? jb : 0), plane = col >> 3;
                const int off = (plane * R + (row ^ (plane & 7))) * 8 + 4 * g;
                const hh_h4 xh = *reinterpret_cast<const hh_h4 *>(Zh + off), xl = *reinterpret_cast<const hh_h4 *>(Zl + off);
                const float4 b = *reinterpret_cast<const float4 *>(bl + 1024 + (ok ? jb : 0));
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float x = (float)xh[i] + (float)xl[i];
                    const float v = ok ? x + (acc[0][0][4 * q + i] + bb[i]) : 0.0f;
                    y[4 * q + i] = v;
                    ssum += v * v;
                }
            }
            { /* the row's other sixteen columns of this quarter sit on lane ^ 32: low half + high half, the same order on both */
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(ssum), __float_as_int(ssum), false, false);
                const float other = __int_as_float(g ? sw[0] : sw[1]);
                const float tot = g ? other + ssum : ssum + other;
                if (!g) npart[wq * R + row] = tot;
            }
            HHP_T(10);
            __syncthreads();
            HHP_T(11);
            {
                const float nn = ((npart[row] + npart[R + row]) + npart[2 * R + row]) + npart[3 * R + row];
                const float den = fmaxf(sqrtf(nn), 1e-12f);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int jb = wq * 32 + 8 * q + 4 * g;
                    if (jb < 100) {
                        const int col = 400 + jb, plane = col >> 3;
                        const int off = (plane * R + (row ^ (plane & 7))) * 8 + 4 * g;
                        hh_h4 h, l;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float v = y[4 * q + i] / den;
                            h[i] = (_Float16)v;
                            l[i] = (_Float16)(v - (float)h[i]);
                        }
                        *reinterpret_cast<hh_h4 *>(Zh + off) = h;
                        *reinterpret_cast<hh_h4 *>(Zl + off) = l;
                    }
                }
            }
            __syncthreads();
        }

        HHP_T(3);
        /* the next tile's observation rows: requested before the long GEMM, written to the (idle) X planes after it */
        float xv[XPT];
        if (more) {
            const int od = bank.net[nnet].obs_dim;
#pragma unroll
            for (int u = 0; u < XPT; u++) {
                const int e = tid + u * NTH, i = e >> 5, c = e & 31, r = rows_next[i];
                xv[u] = (r >= 0 && c < od) ? obs[(size_t)r * obs_stride + c] : 0.0f;
            }
        }
        /* ---- L2 ---- */
        {
            hh_f32x16 acc[RH][NT];
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[f][t] = hhp_zero16();
            hhp_gemm_h<NT, RH>(reinterpret_cast<const float4 *>(Zh), reinterpret_cast<const float4 *>(Zl), 0, HHP_H / 16, H.wsh, H.wsl, 0, HHP_H, wave * WC, lane, acc);
            HHP_T(4);
            /* ---- L3 straight from the registers: the tanh of this wave's 32 WC columns never goes to LDS.  Every C^T tile is two
             * 16-k B fragments as it stands (register group pairs), the output weights come in the matching k order (hhp_hidx_t), and
             * the wave accumulates ITS k range of the logits for its rows; the WV partials meet in LDS afterwards. ---- */
            float4 wfh[NT][2], wfl[NT][2];
            {
                const int hh_ = lane >> 5;
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const int idx = ((((wave * WC + t * 32) >> 4) + b) * 2 + hh_) * HHP_OUT + ci;
                        wfh[t][b] = H.wah[idx]; wfl[t][b] = H.wal[idx];
                    }
            }
            constexpr int NP = NT / 2; /* partial sums per wave: one per 64 columns, so that every tile width adds the same eight ranges in the same order */
            hh_f32x16 lacc[RH][NP];
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int pp = 0; pp < NP; pp++) lacc[f][pp] = hhp_zero16();
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 b = *reinterpret_cast<const float4 *>(bl + 512 + wave * WC + t * 32 + 8 * q + 4 * g);
                        const hh_f2 p0 = hhp_tanh2(hh_f2{acc[f][t][4 * q + 0], acc[f][t][4 * q + 1]} + hh_f2{b.x, b.y});
                        const hh_f2 p1 = hhp_tanh2(hh_f2{acc[f][t][4 * q + 2], acc[f][t][4 * q + 3]} + hh_f2{b.z, b.w});
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
                        lacc[f][t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfh[t][b]), xh, lacc[f][t >> 1], 0, 0, 0);
                        lacc[f][t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfl[t][b]), xh, lacc[f][t >> 1], 0, 0, 0);
                        lacc[f][t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhp_as_h8(wfh[t][b]), xl, lacc[f][t >> 1], 0, 0, 0);
                    }
                }
            HHP_T(5);
            __syncthreads(); /* every wave is done reading Z: the partials go there (8 x R rows x PZS floats from the start of the hi plane) */
            HHP_T(6);
#pragma unroll
            for (int f = 0; f < RH; f++)
#pragma unroll
                for (int pp = 0; pp < NP; pp++)
#pragma unroll
                    for (int q = 0; q < 4; q++) /* the lane's row, output columns 8 q + 4 g .. + 3 */
                        *reinterpret_cast<float4 *>(Pz + (wave * NP + pp) * (R * PZS) + (f * 32 + ci) * PZS + 8 * q + 4 * g) =
                            make_float4(lacc[f][pp][4 * q], lacc[f][pp][4 * q + 1], lacc[f][pp][4 * q + 2], lacc[f][pp][4 * q + 3]);
        }
        if (more) {
            if (nnet != net) HHP_STAGE_BIAS(nnet); /* wave-uniform; this tile's L1 / attention / L2 biases are spent */
#pragma unroll
            for (int u = 0; u < XPT; u++) {
                const int e = tid + u * NTH;
                hhp_split_store<R>(Xh, Xl, hhp_haidx<R>(e & 31, e >> 5), xv[u]);
            }
        }
        __syncthreads();
        HHP_T(7);
        for (int e = tid; e < R * HHP_OUT; e += NTH) {
            const int i = e >> 5, c = e & 31, pe = i * PZS + c;
            float v = Pz[pe];
#pragma unroll
            for (int w_ = 1; w_ < 8; w_++) v += Pz[w_ * (R * PZS) + pe]; /* eight 64-column ranges in column order, whatever the tile width */
            v += bar; /* c == tid & 31 == ci for every e of this thread */
            Lg[e] = v;
            if (logits_out && rows[i] >= 0) logits_out[(size_t)rows[i] * HH_POLICY_LOGITS + c] = c < N.n_out ? v : 0.0f;
        }
        __syncthreads();
        /* greedy decode (env_base.py:373-382), one thread per (row, MultiDiscrete component): first maximum of its segment */
        if (tid < R * 4) {
            const int row = tid >> 2, k = tid & 3;
            const int lo = k == 0 ? 0 : (k == 1 ? 13 : (k == 2 ? 22 : 24)), hi = k == 0 ? 13 : (k == 1 ? 22 : (k == 2 ? 24 : 26));
            const float *lg = Lg + row * 32;
            int best = lo;
            if (k < (N.n_out == 26 ? 4 : 3))
                for (int c = lo + 1; c < hi; c++) if (lg[c] > lg[best]) best = c;
            int a = (best - lo) << (8 * k);
            a |= __builtin_amdgcn_mov_dpp(a, 0xB1, 0xf, 0xf, true); /* OR over the quad of the row's four components */
            a |= __builtin_amdgcn_mov_dpp(a, 0x4E, 0xf, 0xf, true);
            if (k == 0 && rows[row] >= 0) reinterpret_cast<int *>(actions)[rows[row]] = a;
        }
        HHP_T(8);
        if (!more) break;
        gt += tile_stride; net = nnet; tile = ntile; cnt = ncnt;
        __syncthreads(); /* the partials (hi plane) and the logits are read: the next tile may write Z */
    }
}

template <int RH>
__global__ __launch_bounds__(256 * RH, RH == 1 ? 2 : 1) void hh_k_policy_h(HhpBank bank, HhpBankH bankh, int n_nets, const float *__restrict__ obs, int obs_stride,
                                                                           int *counts, const int *__restrict__ lists, int max_rows,
                                                                           int8_t *__restrict__ actions, float *__restrict__ logits_out, int consume) {
    extern __shared__ __align__(16) unsigned char ldsb[];
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(hhp_row_count(counts, n, consume), max_rows) : 0;
    hhp_forward_tiles<RH>(bank, bankh, cn, obs, obs_stride, lists, max_rows, actions, logits_out, ldsb, (int)blockIdx.x, (int)gridDim.x);
    hhp_consume_counts(counts, consume);
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
/* the same in the k order of hhp_hidx_t (the output layer, contracted with the shared layer's C tile in registers) */
static inline void hhp_split_put_t(std::vector<uint16_t> &hi, std::vector<uint16_t> &lo, size_t off, int k, int col, int J, float v) {
    const uint16_t h = hhp_f2h(v);
    hi[off + hhp_hidx_t(k, col, J)] = h;
    lo[off + hhp_hidx_t(k, col, J)] = hhp_f2h(v - hhp_h2f(h));
}

#endif /* HH_POLICY_KERNEL_H16_H */
