/*
 * hh_policy_kernel.h — the frozen pilot / opponent networks as ONE fused gfx950 kernel (C ABI: include/hh_policy.h).
 *
 * Replaces envs/env_base.py:349-398 `_policy_actions` (one PyTorch forward of batch 1 per live unit per tick) for every
 * unit of every arena at once.  Actor half of models/ac_models_hetero.py (Esc1 29-103, Esc2 105-180, Fight1 181-291,
 * Fight2 293-404):
 *     three input FCs (tanh) on column ranges of the observation, concatenated to 500
 *     [fight nets: the 100-wide third block x gets  x <- normalize(x + out_proj(v_proj(x)))  — MultiheadAttention over a
 *      sequence of length 1, where the softmax is identically 1]
 *     shared layer 500 -> 500 (tanh), act_out 500 -> 26 | 24 logits, arg-max per MultiDiscrete component.
 *
 * This is the one dense contraction on the whole path, so it runs on the matrix cores.  gfx950 has no tf32-like mode and its fp32-in MFMA runs
 * at the vector rate, so every operand is split into two fp16 halves and a product is accumulated in fp32 as hi hi + lo hi + hi lo
 * (hh_policy_kernel_h16.h: logits within 1e-6 of a float64 forward, like PyTorch's own fp32 forward).  This header holds what the forms share
 * — the bank of loaded networks (weights packed on the host), the row binning by network (hh_k_policy_bin), the choice of a form per call —
 * and the C ABI; the kernels:
 *     hh_k_policy_h<1|2>     hh_policy_kernel_h16.h   activations in an LDS tile of 32 | 64 rows, weights out of L2              small calls
 *     hh_k_policy_w16<4|8>   hh_policy_kernel_w16.h   weights streamed through LDS, activations in registers, 64 | 128-row tiles  >= 40 rows x n_CU
 *     hh_k_policy_ppo, hh_k_policy_w16_ppo            the PPO sampler's step (hh_policy_sample): actor + Categorical draw + value branch
 * (Round 6 retired the fp32-MFMA forward hh_k_policy of round 2 and the 32-rows-per-wave form hh_k_policy_w of round 4: no row count selected them.)
 * Rows of different networks are first binned into per-network lists; rows without a network get a zero action.  Per-row results do not depend
 * on which tile a row lands in, so the outputs are deterministic although the binning order is not.
 */
#ifndef HH_POLICY_KERNEL_H
#define HH_POLICY_KERNEL_H

#include <hip/hip_runtime.h>
#include <math.h>

#include "hh_policy.h"

typedef float hh_f32x16 __attribute__((ext_vector_type(16)));

#define HHP_ROWS 32        /* rows per workgroup tile */
#define HHP_H 512          /* padded hidden width (500) */
#define HHP_XK 32          /* padded observation width (<= 30) */
#define HHP_ATT_K 104      /* padded attention width (100) as K */
#define HHP_ATT_J 128      /* ... and as output columns */
#define HHP_OUT 32         /* padded logits (26 | 24) */

struct HhpNet {
    const float *w1p, *b1;   /* [4][2][512][4], [512] */
    const float *wovp, *bov; /* [13][2][128][4], [128]  (fight nets) */
    const float *wsp, *bs;   /* [64][2][512][4], [512] */
    const float *wap, *ba;   /* [64][2][32][4], [32] */
    int kind, n_out, obs_dim, has_att;
};
struct HhpBank {
    HhpNet net[HH_POLICY_MAX_NETS];
};

/* packed index of element (k, col) of a [K x J] operand */
__host__ __device__ inline size_t hhp_pidx(int k, int col, int J) { return ((size_t)((k >> 3) * 2 + (k & 1)) * J + col) * 4 + ((k >> 1) & 3); }
/* two at a time: the multiply, the add and the final multiply-add are packed instructions (v_pk_mul / add / fma_f32: one issue slot
 * for both values); the exponential and the reciprocal stay one transcendental each */
typedef float hh_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hh_f2 hhp_tanh2(hh_f2 x) {
    const hh_f2 t = x * (hh_f2)(2.885390081777926815f); /* 2 log2(e) */
    hh_f2 e;
    e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
    const hh_f2 d = e + (hh_f2)(1.0f);
    hh_f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return __builtin_elementwise_fma(r, (hh_f2)(-2.0f), (hh_f2)(1.0f));
}

/* rows -> per-network lists.  One atomic ROUND TRIP per WORKGROUP: a per-row atomic on two or four hot counters serialises (measured
 * 187 us for 32768 rows; one returning atomic per wave and network 13.6 us, all of a wave's in one instruction 10.9 us — device-scope
 * atomics on one address cost ~13 ns apiece however they are issued).  The four waves' ballots meet in LDS, lane n - 1 of wave 0
 * carries the workgroup's count for network n, the (up to eight) atomics leave as one instruction, and every row takes its slot from
 * the base of its network + the counts of the waves before its own + its rank in its wave's ballot.  The counters are found zero:
 * the forward kernel that consumed the previous lists cleared them (hhp_consume_counts). */
__global__ __launch_bounds__(256) void hh_k_policy_bin(int n_rows, const uint8_t *__restrict__ sel, const uint8_t *__restrict__ lut,
                                                       int max_rows, int *__restrict__ counts, int *__restrict__ lists,
                                                       int8_t *__restrict__ actions) {
    __shared__ int wcnt[4][HH_POLICY_MAX_NETS], wbase[HH_POLICY_MAX_NETS];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = r < n_rows ? (int)lut[sel[r]] : 0;
    if (r < n_rows && s == 0) reinterpret_cast<int *>(actions)[r] = 0;
    int rank = 0;
#pragma unroll
    for (int n = 1; n <= HH_POLICY_MAX_NETS; n++) {
        const unsigned long long m = __ballot(s == n);
        if (lane == n - 1) wcnt[wave][n - 1] = __popcll(m);
        if (s == n) rank = __popcll(m & ((1ULL << lane) - 1ULL));
    }
    __syncthreads();
    if (threadIdx.x < HH_POLICY_MAX_NETS) {
        const int tot = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        wbase[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x * HH_BIN_STRIDE], tot) : 0;
    }
    __syncthreads();
    if (s > 0) {
        int at = wbase[s - 1] + rank;
        for (int v = 0; v < wave; v++) at += wcnt[v][s - 1];
        if (at < max_rows) lists[(size_t)(s - 1) * max_rows + at] = r;
    }
}

/* The forward kernel that runs over freshly binned lists clears the row counters behind itself (consume bit 0): a workgroup takes a
 * ticket when it is done (it read the counters at its start), and the last one to do so saves the counts for later calls that re-use
 * the lists (sel == NULL; they read the saved copy: consume bit 1) and clears the live ones for the next binning pass — no clear
 * kernel, and nothing that depends on how a HIP graph strings the calls together.  Tickets are sharded over 32 sub-counters with one
 * top-level ticket per shard: ~500 workgroups finish together and same-address atomics serialise (a single ticket counter measured
 * +5 us per launch).  counts[] in units of HH_BIN_STRIDE ints (every counter on its own 128-byte line): 0..7 rows per network,
 * 8 top-level ticket, 9..40 shard tickets, 41..48 saved rows per network. */
#define HHP_COUNTS_INTS (49 * HH_BIN_STRIDE)
#define HHP_CONSUME 1
#define HHP_FROM_SAVED 2
__device__ __forceinline__ int hhp_row_count(const int *counts, int n, int consume) {
    return counts[((consume & HHP_FROM_SAVED) ? 41 + n : n) * HH_BIN_STRIDE];
}
__device__ __forceinline__ void hhp_consume_counts(int *counts, int consume) {
    if ((consume & HHP_CONSUME) && threadIdx.x == 0) {
        const int j = blockIdx.x & 31;
        const int nj = ((int)gridDim.x - j + 31) >> 5; /* workgroups of this shard */
        if (atomicAdd(&counts[(9 + j) * HH_BIN_STRIDE], 1) == nj - 1) {
            counts[(9 + j) * HH_BIN_STRIDE] = 0;
            const int shards = min((int)gridDim.x, 32);
            if (atomicAdd(&counts[HH_POLICY_MAX_NETS * HH_BIN_STRIDE], 1) == shards - 1) {
#pragma unroll
                for (int n = 0; n < HH_POLICY_MAX_NETS; n++) {
                    counts[(41 + n) * HH_BIN_STRIDE] = counts[n * HH_BIN_STRIDE];
                    counts[n * HH_BIN_STRIDE] = 0;
                }
                counts[HH_POLICY_MAX_NETS * HH_BIN_STRIDE] = 0;
            }
        }
    }
}

__device__ __forceinline__ hh_f32x16 hhp_zero16() {
    hh_f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}
/* C/D layout of the 32x32 MFMA: lane holds column (lane & 31), rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) */
__device__ __forceinline__ int hhp_crow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

#include "hh_policy_kernel_h16.h"
#include "hh_policy_kernel_ppo.h"
#include "hh_policy_kernel_w.h"
#include "hh_policy_kernel_w16.h"

/* ===================================================================== host side */
#define HHP_SLOT_BYTES ((size_t)4 << 20) /* fp32 blob 1.19 MB + fp16 planes 1.18 MB per network, padded to 4 MB */
struct hh_policy {
    int device, max_rows;
    HhpBank bank;
    HhpBankH bankh;
    int binned_rows;          /* n_rows of the call that built the current row lists (0: none) */
    int tile_rows;            /* HH_POLICY_TILE=32 / 64: that tile instance of the split-fp16 kernel (A/B runs); 0 = unset: chosen per call by the row count */
    int n_cu;
    int persist;              /* HH_POLICY_PERSIST=0: 64-row tiles one workgroup per tile instead of a grid-stride walk (A/B runs) */
    int n_nets;               /* highest loaded slot + 1 */
    float *blob[HH_POLICY_MAX_NETS];
    uint16_t *blobh[HH_POLICY_MAX_NETS];
    char *slab;               /* ONE allocation for every network's weights (slot stride HHP_SLOT_BYTES): large, 2 MB-aligned mappings keep
                                 the weight stream on a handful of TLB entries instead of a fresh small allocation per network */
    uint8_t *lut;             /* [256] dev */
    int *counts, *lists;      /* counters (HHP_COUNTS_INTS), [MAX_NETS][max_rows] dev */
    hh_world *bound;          /* hh_bind_policy: the world whose kernels write the lists (one world per bank), or nullptr */
    HhpBankX bankx;           /* the weights-through-LDS forms (hh_policy_kernel_w16.h): one linear stream of 1 KB fragments per network */
    unsigned char *xblob[HH_POLICY_MAX_NETS];
    int wform;                /* HH_POLICY_W: 3 / 2 = always hh_k_policy_w16<8> / <4>, 0 = the tile forms only, unset (-1) = by row count (hhp_choose_form) */
    HhpCritBank cbank;        /* hh_policy_set_critic: the value branches of the trainable policies (hh_policy_sample) */
    char *cblob[HH_POLICY_MAX_NETS]; /* one allocation per loaded value branch */
    HhpCritBankX cbankx;      /* the same as fragment streams for hh_k_policy_w16_ppo */
    char *cxblob[HH_POLICY_MAX_NETS];
};
static void hhp_forget_world(hh_policy *p) { p->bound = nullptr; }
static void hhp_unbind(hh_policy *p) {
    if (p->bound) {
        hh_world *w = p->bound;
        w->P.pol_lut = nullptr; w->P.pol_counts = nullptr; w->P.pol_lists = nullptr; w->P.pol_max_rows = 0;
        w->bound_policy = nullptr;
        p->bound = nullptr;
    }
}

static const int HHP_INPUTS[4][3][3] = {
    /* first column, last + 1, width: models/ac_models_hetero.py Fight1 214-231, Fight2 326-343, Esc1 46-63, Esc2 122-139 */
    {{0, 12, 200}, {12, 26, 200}, {0, 26, 100}},
    {{0, 10, 200}, {10, 24, 200}, {0, 24, 100}},
    {{0, 7, 150}, {7, 25, 250}, {25, 30, 100}},
    {{0, 6, 150}, {6, 24, 250}, {24, 29, 100}},
};

extern "C" int hh_policy_create(int device, int32_t max_rows, hh_policy **out) {
    if (!out || max_rows <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device"; return HH_E_NODEV; }
    if (device < 0 || device >= ndev) { g_err = "bad device index"; return HH_E_ARG; }
    DeviceGuard guard_(device);
    if (!guard_.ok) { g_err = "hipSetDevice failed"; return HH_E_HIP; }
    hh_policy *p = new (std::nothrow) hh_policy();
    if (!p) { g_err = "hh_policy_create: out of host memory"; return HH_E_HIP; }
    p->device = device; p->max_rows = max_rows; p->n_nets = 0; p->binned_rows = 0; p->bound = nullptr;
    memset(&p->bank, 0, sizeof(p->bank));
    memset(&p->bankh, 0, sizeof(p->bankh));
    { const char *e = getenv("HH_POLICY_TILE"); p->tile_rows = e ? atoi(e) : 0; } /* 32 / 64: that instance; unset: by row count (hhp_rows_suit_wide_tiles) */
    { const char *e = getenv("HH_POLICY_PERSIST"); p->persist = e ? atoi(e) : 1; }
    { hipDeviceProp_t prop; p->n_cu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256; }
    for (int i = 0; i < HH_POLICY_MAX_NETS; i++) { p->blob[i] = nullptr; p->blobh[i] = nullptr; }
    p->slab = nullptr;
    memset(&p->cbank, 0, sizeof(p->cbank));
    memset(&p->bankx, 0, sizeof(p->bankx));
    memset(&p->cbankx, 0, sizeof(p->cbankx));
    for (int i = 0; i < HH_POLICY_MAX_NETS; i++) p->cxblob[i] = nullptr;
    for (int i = 0; i < HH_POLICY_MAX_NETS; i++) { p->cblob[i] = nullptr; p->xblob[i] = nullptr; }
    { const char *e = getenv("HH_POLICY_W"); p->wform = e ? atoi(e) : -1; }
    p->lut = nullptr; p->counts = nullptr; p->lists = nullptr;
    hipError_t e = hipMalloc(&p->lut, 256);
    if (e == hipSuccess) e = hipMemset(p->lut, 0, 256);
    if (e == hipSuccess) e = hipMalloc(&p->counts, HHP_COUNTS_INTS * sizeof(int)); /* rows per network + the tickets of hhp_consume_counts */
    if (e == hipSuccess) e = hipMemset(p->counts, 0, HHP_COUNTS_INTS * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&p->lists, (size_t)HH_POLICY_MAX_NETS * max_rows * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&p->slab, HHP_SLOT_BYTES * HH_POLICY_MAX_NETS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_h<1>), hipFuncAttributeMaxDynamicSharedMemorySize, HHPH_LDS_BYTES(1));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_h<2>), hipFuncAttributeMaxDynamicSharedMemorySize, HHPH_LDS_BYTES(2));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_ppo), hipFuncAttributeMaxDynamicSharedMemorySize, HHPP_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_w16<4>), hipFuncAttributeMaxDynamicSharedMemorySize, HHX_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_w16<8>), hipFuncAttributeMaxDynamicSharedMemorySize, HHX_LDS_BYTES_NB(4));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hh_k_policy_w16_ppo), hipFuncAttributeMaxDynamicSharedMemorySize, HHXC_LDS_BYTES);
    if (e != hipSuccess) {
        g_err = std::string("hh_policy_create: ") + hipGetErrorString(e);
        if (p->lut) (void)hipFree(p->lut);
        if (p->counts) (void)hipFree(p->counts);
        if (p->lists) (void)hipFree(p->lists);
        if (p->slab) (void)hipFree(p->slab);
        delete p;
        return HH_E_HIP;
    }
    *out = p;
    return HH_OK;
}

extern "C" int hh_policy_destroy(hh_policy *p) {
    if (!p) return HH_E_ARG;
    hhp_unbind(p); /* a world still bound to this bank goes back to emitting selector bytes only */
    DeviceGuard guard_(p->device);
    (void)hipFree(p->slab);
    for (int i = 0; i < HH_POLICY_MAX_NETS; i++) { if (p->cblob[i]) (void)hipFree(p->cblob[i]); if (p->xblob[i]) (void)hipFree(p->xblob[i]); if (p->cxblob[i]) (void)hipFree(p->cxblob[i]); }
    (void)hipFree(p->lut); (void)hipFree(p->counts); (void)hipFree(p->lists);
    delete p;
    return HH_OK;
}

static int hhp_set_net(hh_policy *p, int32_t slot, const hh_net_weights *w);
extern "C" int hh_policy_set_net(hh_policy *p, int32_t slot, const hh_net_weights *w) {
    try { /* the repacking buffers are std::vectors: an allocation failure must not unwind through the C ABI */
        return hhp_set_net(p, slot, w);
    } catch (const std::exception &e) {
        g_err = std::string("hh_policy_set_net: ") + e.what();
        return HH_E_HIP;
    }
}
static int hhp_set_net(hh_policy *p, int32_t slot, const hh_net_weights *w) {
    if (!p || !w || slot < 0 || slot >= HH_POLICY_MAX_NETS || w->kind < 0 || w->kind > 3) { g_err = "bad argument"; return HH_E_ARG; }
    const bool att = w->kind <= HH_NET_FIGHT2;
    if (!w->shared_w || !w->shared_b || !w->out_w || !w->out_b || (att && (!w->att_in_proj_w || !w->att_in_proj_b || !w->att_out_w || !w->att_out_b))) {
        g_err = "hh_policy_set_net: missing weight pointer"; return HH_E_ARG;
    }
    for (int k = 0; k < 3; k++) if (!w->inp_w[k] || !w->inp_b[k]) { g_err = "hh_policy_set_net: missing input layer"; return HH_E_ARG; }
    HH_GUARD(p);
    /* the slot's value branch holds its own permuted copy of the shared layer and the input widths of the kind it was loaded for: a new
     * actor makes both stale.  hh_policy_sample refuses vf for the slot until hh_policy_set_critic is called again (set_net + set_critic
     * are a pair; the old copy stays allocated and is simply never read) */
    p->cbank.c[slot].loaded = 0;
    p->cbankx.c[slot].loaded = 0;
    const int n_out = (w->kind == HH_NET_FIGHT1 || w->kind == HH_NET_ESC1) ? 26 : 24;
    /* blob: w1p | b1 | wovp | bov | wsp | bs | wap | ba   (float counts; every section 16-byte aligned) */
    const size_t n_w1 = 4 * 2 * HHP_H * 4, n_wov = 13 * 2 * HHP_ATT_J * 4, n_ws = 64 * 2 * HHP_H * 4, n_wa = 64 * 2 * HHP_OUT * 4;
    const size_t o_w1 = 0, o_b1 = o_w1 + n_w1, o_wov = o_b1 + HHP_H, o_bov = o_wov + n_wov, o_ws = o_bov + HHP_ATT_J, o_bs = o_ws + n_ws,
                 o_wa = o_bs + HHP_H, o_ba = o_wa + n_wa, total = o_ba + HHP_OUT;
    std::vector<float> B(total, 0.0f);
    /* the same four matrices as (hi, lo) fp16 fragment planes for hh_k_policy_h: w1 | wov | ws | wa, each [K/16][2][J][8] */
    const size_t h_w1 = 0, h_wov = h_w1 + (size_t)2 * 2 * HHP_H * 8, h_ws = h_wov + (size_t)7 * 2 * HHP_ATT_J * 8, h_wa = h_ws + (size_t)32 * 2 * HHP_H * 8,
                 h_total = h_wa + (size_t)32 * 2 * HHP_OUT * 8;
    std::vector<uint16_t> Hh(h_total, 0), Hl(h_total, 0);
    int off = 0, obs_dim = 0;
    for (int k = 0; k < 3; k++) {
        const int c0 = HHP_INPUTS[w->kind][k][0], c1 = HHP_INPUTS[w->kind][k][1], wd = HHP_INPUTS[w->kind][k][2];
        for (int o = 0; o < wd; o++) {
            for (int c = c0; c < c1; c++) {
                const float v = w->inp_w[k][(size_t)o * (c1 - c0) + (c - c0)];
                B[o_w1 + hhp_pidx(c, off + o, HHP_H)] = v;
                hhp_split_put(Hh, Hl, h_w1, c, off + o, HHP_H, v);
            }
            B[o_b1 + off + o] = w->inp_b[k][o];
        }
        off += wd;
        if (c1 > obs_dim) obs_dim = c1;
    }
    if (att) { /* out_proj(v_proj(x)): Wov = Wo Wv, bov = Wo bv + bo, folded in double */
        const float *wv = w->att_in_proj_w + (size_t)200 * 100, *bv = w->att_in_proj_b + 200;
        for (int j = 0; j < 100; j++) {
            for (int k = 0; k < 100; k++) {
                double s = 0.0;
                for (int m = 0; m < 100; m++) s += (double)w->att_out_w[(size_t)j * 100 + m] * (double)wv[(size_t)m * 100 + k];
                B[o_wov + hhp_pidx(k, j, HHP_ATT_J)] = (float)s;
                hhp_split_put(Hh, Hl, h_wov, k, j, HHP_ATT_J, (float)s);
            }
            double s = (double)w->att_out_b[j];
            for (int m = 0; m < 100; m++) s += (double)w->att_out_w[(size_t)j * 100 + m] * (double)bv[m];
            B[o_bov + j] = (float)s;
        }
    }
    for (int j = 0; j < 500; j++) {
        for (int k = 0; k < 500; k++) { B[o_ws + hhp_pidx(k, j, HHP_H)] = w->shared_w[(size_t)j * 500 + k]; hhp_split_put(Hh, Hl, h_ws, k, j, HHP_H, w->shared_w[(size_t)j * 500 + k]); }
        B[o_bs + j] = w->shared_b[j];
    }
    for (int j = 0; j < n_out; j++) {
        for (int k = 0; k < 500; k++) { B[o_wa + hhp_pidx(k, j, HHP_OUT)] = w->out_w[(size_t)j * 500 + k]; hhp_split_put_t(Hh, Hl, h_wa, k, j, HHP_OUT, w->out_w[(size_t)j * 500 + k]); }
        B[o_ba + j] = w->out_b[j];
    }
    { /* the same four matrices as ONE linear stream of 1 KB fragments in consumption order (hh_policy_kernel_w16.h) */
        auto w1 = [&](int k, int col) { return k < HHP_XK ? B[o_w1 + hhp_pidx(k, col, HHP_H)] : 0.0f; };
        auto wov = [&](int k, int col) { return (att && k < 100 && col < 100) ? B[o_wov + hhp_pidx(k, col, HHP_ATT_J)] : 0.0f; };
        auto wsf = [&](int k, int col) { return (k < 500 && col < 500) ? w->shared_w[(size_t)col * 500 + k] : 0.0f; };
        auto waf = [&](int k, int col) { return (k < 500 && col < n_out) ? w->out_w[(size_t)col * 500 + k] : 0.0f; };
        /* the fragment shape of hh_k_policy_w16 (16 columns x 32 k per piece; chunk order of hh_policy_kernel_w16.h) */
        std::vector<uint16_t> X((size_t)HHX_STREAM_PIECES * (HHW_PIECE / 2), 0);
        for (int T = 0; T < 32; T++)
            for (int k = 0; k < 32; k++)
                for (int c = 0; c < 16; c++) hhx_put(X, (size_t)T * 2, k, 16 * T + c, true, w1(k, 16 * T + c));
        for (int j = 0; j < 7; j++)
            for (int kb = 0; kb < 4; kb++)
                for (int wq = 0; wq < 32; wq++)
                    for (int c = 0; c < 16; c++) { /* K = hidden columns 384 + 32 kb + wq; the block's own index is that - 400 */
                        const int kh = 384 + 32 * kb + wq - 400;
                        hhx_put(X, (size_t)HHX_L1_PIECES + (size_t)(j * 4 + kb) * 2, wq, 16 * j + c, false, kh >= 0 ? wov(kh, 16 * j + c) : 0.0f);
                    }
        for (int pp = 0; pp < 8; pp++)
            for (int q = 0; q < 4; q++)
                for (int kk = 0; kk < 4; kk++)
                    for (int t = 0; t < 4; t++)
                        for (int wq = 0; wq < 32; wq++)
                            for (int c = 0; c < 16; c++)
                                hhx_put(X, (size_t)HHX_L1_PIECES + HHX_ATT_PIECES + (size_t)((pp * 4 + q) * 16 + kk * 4 + t) * 2, wq, c, false,
                                        wsf(32 * (4 * q + kk) + wq, 16 * (4 * pp + t) + c));
        for (int kb = 0; kb < 16; kb++)
            for (int t = 0; t < 2; t++)
                for (int wq = 0; wq < 32; wq++)
                    for (int c = 0; c < 16; c++)
                        hhx_put(X, (size_t)HHX_L1_PIECES + HHX_ATT_PIECES + HHX_L2_PIECES + (size_t)(kb * 2 + t) * 2, wq, c, false, waf(32 * kb + wq, 16 * t + c));
        if (!p->xblob[slot]) HIPCHK(hipMalloc(&p->xblob[slot], (size_t)HHX_STREAM_PIECES * HHW_PIECE));
        HIPCHK(hipMemcpy(p->xblob[slot], X.data(), (size_t)HHX_STREAM_PIECES * HHW_PIECE, hipMemcpyHostToDevice));
        p->bankx.stream[slot] = p->xblob[slot];
    }
    static_assert(HHP_SLOT_BYTES >= (size_t)2 * 1024 * 1024 + 2 * 309248 * 2, "slot too small");
    if (total * sizeof(float) > (size_t)2 * 1024 * 1024 || 2 * h_total * sizeof(uint16_t) > HHP_SLOT_BYTES - (size_t)2 * 1024 * 1024) { g_err = "internal: blob exceeds its slot"; return HH_E_ARG; }
    p->blob[slot] = reinterpret_cast<float *>(p->slab + (size_t)slot * HHP_SLOT_BYTES);                                   /* first 2 MB of the slot */
    p->blobh[slot] = reinterpret_cast<uint16_t *>(p->slab + (size_t)slot * HHP_SLOT_BYTES + (size_t)2 * 1024 * 1024);      /* second 2 MB */
    HIPCHK(hipMemcpy(p->blob[slot], B.data(), total * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->blobh[slot], Hh.data(), h_total * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->blobh[slot] + h_total, Hl.data(), h_total * sizeof(uint16_t), hipMemcpyHostToDevice));
    {
        HhpNetH &Hn = p->bankh.net[slot];
        const uint16_t *bh = p->blobh[slot], *bl = p->blobh[slot] + h_total;
        Hn.w1h = reinterpret_cast<const float4 *>(bh + h_w1); Hn.w1l = reinterpret_cast<const float4 *>(bl + h_w1);
        Hn.wovh = reinterpret_cast<const float4 *>(bh + h_wov); Hn.wovl = reinterpret_cast<const float4 *>(bl + h_wov);
        Hn.wsh = reinterpret_cast<const float4 *>(bh + h_ws); Hn.wsl = reinterpret_cast<const float4 *>(bl + h_ws);
        Hn.wah = reinterpret_cast<const float4 *>(bh + h_wa); Hn.wal = reinterpret_cast<const float4 *>(bl + h_wa);
    }
    HhpNet &N = p->bank.net[slot];
    float *b = p->blob[slot];
    N.w1p = b + o_w1; N.b1 = b + o_b1; N.wovp = b + o_wov; N.bov = b + o_bov; N.wsp = b + o_ws; N.bs = b + o_bs; N.wap = b + o_wa; N.ba = b + o_ba;
    N.kind = w->kind; N.n_out = n_out; N.obs_dim = obs_dim; N.has_att = att ? 1 : 0;
    if (slot + 1 > p->n_nets) p->n_nets = slot + 1;
    return HH_OK;
}

extern "C" int hh_policy_set_lut(hh_policy *p, const uint8_t *lut) {
    if (!p || !lut) { g_err = "null argument"; return HH_E_ARG; }
    for (int i = 0; i < 256; i++) if (lut[i] > HH_POLICY_MAX_NETS || (lut[i] && !p->blob[lut[i] - 1])) { g_err = "hh_policy_set_lut: selector maps to an empty slot"; return HH_E_ARG; }
    HH_GUARD(p);
    HIPCHK(hipMemcpy(p->lut, lut, 256, hipMemcpyHostToDevice));
    return HH_OK;
}

extern "C" int hh_policy_set_tile_rows(hh_policy *p, int32_t rows) {
    if (!p || (rows != 0 && rows != 32 && rows != 64)) { g_err = "hh_policy_set_tile_rows: rows must be 0 (by row count), 32 or 64"; return HH_E_ARG; }
    p->tile_rows = rows;
    return HH_OK;
}

/* Which tile width (HH_POLICY_TILE unset): the 64-row instance streams half the weights per row and is 6 - 9 % faster when its tiles come
 * in whole rounds of one per CU (16384, 32768, 49152 ... rows on 256 CUs: 42.2 against 46.2 us, 79.4 against 84.2), and slower when the last
 * round is mostly empty (24576 rows: 78 against 67 us; 8192: half the CUs idle) — the 32-row instance packs two workgroups per CU and fills
 * a partial round better.  The host knows the total row count; how the rows spread over the networks only rounds each network up by a tile. */
static inline bool hhp_rows_suit_wide_tiles(int n_rows, int n_cu) {
    const int tiles = (n_rows + 63) / 64, rem = tiles % n_cu;
    return tiles >= n_cu && (rem == 0 || rem * 4 > n_cu * 3);
}
/* Which form for how many rows (tools/policy_bench.py, Fight1 + Fight2 rows, us per call back to back on one MI355X):
 *      rows            4096   8192  12288  16384  20480  24576  32768  49152  65536
 *      hh_k_policy_h   23.7   28.0   43.5   41.9   62.8   67.4   78.9  114.0  144.9     (LDS activation tile: 32 / 64 rows per workgroup)
 *      hh_k_policy_w   46.7   47.5   48.4   50.7   53.6   57.2   63.4  108.8  120.6     (32 rows per wave, one wave per SIMD: retired in round 6, never ahead)
 *      hh_k_policy_w16 30.8   32.6   34.4   38.7   51.9   55.8   64.1   93.0  117.4     (weights through LDS, 64 rows per workgroup, two workgroups per CU)
 * A weights-through-LDS tile takes ~31 us however few CUs have one; the tile forms are faster while the rows fit one round of 32-row tiles
 * (two per CU) with room to spare.  So: from 10 k rows that carry a network upwards hh_k_policy_w16, below the tile forms. */
static inline bool hhp_rows_suit_w(int n_rows, int n_cu) { return (long long)n_rows > (long long)n_cu * 40; }
/* ... and its eight-wave instance (128-row tiles, one workgroup per CU, the weight stream shared by twice the rows, three chunks ahead) when those tiles come
 * in whole rounds of one per CU: 59.4 against 61.7 us at 32768 rows, 111.3 against 112.5 at 65536, but 48 against 37 at 16384 (half the CUs idle) */
static inline bool hhp_rows_suit_w8(int n_rows, int n_cu) {
    const int tiles = (n_rows + 127) / 128, rem = tiles % n_cu;
    return tiles >= n_cu && (rem == 0 || rem * 4 > n_cu * 3);
}
enum { HHP_FORM_H32, HHP_FORM_H64, HHP_FORM_W16, HHP_FORM_W16X8 };
/* the form of a forward over heur_rows rows that carry a network: HH_POLICY_W (3 / 2 / 1 = always hh_k_policy_w16<8> / <4> / hh_k_policy_w, 0 = the tile forms only,
 * unset = by row count), then the tile width (hh_policy_set_tile_rows / HH_POLICY_TILE, 0 = by row count) */
static int hhp_choose_form(const hh_policy *p, int heur_rows) {
    if (p->wform == 3) return HHP_FORM_W16X8;
    if (p->wform == 2) return HHP_FORM_W16;
    if (p->wform < 0 && p->tile_rows == 0 && hhp_rows_suit_w(heur_rows, p->n_cu)) return hhp_rows_suit_w8(heur_rows, p->n_cu) ? HHP_FORM_W16X8 : HHP_FORM_W16;
    if (p->tile_rows == 64 || (p->tile_rows == 0 && hhp_rows_suit_wide_tiles(heur_rows, p->n_cu))) return HHP_FORM_H64;
    return HHP_FORM_H32;
}
/* the forward kernel over the current row lists; consume: the last workgroup to read the counters clears them */
static int hhp_launch_forward(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, int8_t *actions, float *logits, int consume, hipStream_t st,
                              int32_t live_rows = -1) {
    const int grid = (n_rows + HHP_ROWS - 1) / HHP_ROWS + p->n_nets; /* upper bound of the tiles over all networks */
    /* the form is chosen by the rows that CARRY a network: a bound HighLevelEnv world lists one side's units per call, half of its
     * [N, 6] row buffer at most (a full-buffer count picked 64-row tiles for 1.5 rounds of work: 78 against 67 us at 8192 arenas) */
    const int form = hhp_choose_form(p, live_rows >= 0 ? live_rows : n_rows);
    /* the streamed forms walk their tiles grid-stride: a grid for 1.5 x the rows the caller expects to carry a network (all of them if it gave no estimate).  Measured
     * on the commander step with the networks in the loop, four sub-worlds, estimate 0.30 of the slots against 0.32 listed: 110 % -> 5.35e6 (some workgroups take a
     * second tile: twice the call's latency), 125 % -> 5.90e6, 150 % -> 6.23e6, every slot -> 6.01e6 commander-steps/s */
    static const int grid_pct = getenv("HH_POLICY_GRID_PCT") ? atoi(getenv("HH_POLICY_GRID_PCT")) : 150; /* tuning: grid as a percentage of the estimate (0 = every row slot) */
    const long long exp_rows = (live_rows >= 0 && grid_pct > 0) ? ((long long)live_rows * grid_pct + 99) / 100 : (long long)n_rows;
    const int cover = (int)(exp_rows < (long long)n_rows ? exp_rows : (long long)n_rows);
    if (form == HHP_FORM_W16) { /* weights through LDS, activations in registers, 16 rows per wave: 64-row tiles, two workgroups per CU */
        hipLaunchKernelGGL(hh_k_policy_w16<4>, dim3((cover + 63) / 64 + p->n_nets), dim3(256), HHX_LDS_BYTES, st, p->bank, p->bankx, p->n_nets, obs, obs_stride, p->counts,
                           p->lists, p->max_rows, actions, logits, consume);
    } else if (form == HHP_FORM_W16X8) { /* the same with eight waves per workgroup: 128-row tiles, half the weight stream per row */
        hipLaunchKernelGGL(hh_k_policy_w16<8>, dim3((cover + 127) / 128 + p->n_nets), dim3(512), HHX_LDS_BYTES_NB(4), st, p->bank, p->bankx, p->n_nets, obs, obs_stride, p->counts,
                           p->lists, p->max_rows, actions, logits, consume);
    } else if (form == HHP_FORM_H64) { /* persistent: one workgroup per CU walks the tiles grid-stride */
        const int tiles = (n_rows + 63) / 64 + p->n_nets;
        hipLaunchKernelGGL(hh_k_policy_h<2>, dim3(p->persist && tiles > p->n_cu ? p->n_cu : tiles), dim3(512), HHPH_LDS_BYTES(2), st, p->bank, p->bankh, p->n_nets,
                           obs, obs_stride, p->counts, p->lists, p->max_rows, actions, logits, consume);
    } else {
        hipLaunchKernelGGL(hh_k_policy_h<1>, dim3(grid), dim3(256), HHPH_LDS_BYTES(1), st, p->bank, p->bankh, p->n_nets, obs, obs_stride, p->counts, p->lists,
                           p->max_rows, actions, logits, consume);
    }
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* which kernel instance a forward over n_rows rows (live_rows of them carrying a network; < 0: all) launches, as a profiler prints it */
static const char *hhp_form_name(const hh_policy *p, int n_rows, int live_rows) {
    switch (hhp_choose_form(p, live_rows >= 0 ? live_rows : n_rows)) {
    case HHP_FORM_W16X8: return "hh_k_policy_w16<8>";
    case HHP_FORM_W16: return "hh_k_policy_w16<4>";
    case HHP_FORM_H64: return "hh_k_policy_h<2>";
    default: return "hh_k_policy_h<1>";
    }
}
/* hh_policy_sample: the weights-through-LDS form for large calls (HH_POLICY_W = 2 always, 0 / 1 never), the tile form otherwise */
static bool hhp_sampler_is_w16(const hh_policy *p, int n_rows) {
    return p->wform == 2 || p->wform == 3 || (p->wform < 0 && p->tile_rows == 0 && hhp_rows_suit_w(n_rows, p->n_cu));
}
extern "C" int hh_policy_kernel_name(hh_policy *p, int32_t n_rows, int32_t sampler, char *buf, int32_t len) {
    if (!p || !buf || len <= 0 || n_rows <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    const bool one_side = p->bound && p->bound->cfg.env_kind == HH_ENV_HIGHLEVEL;
    snprintf(buf, (size_t)len, "%s", sampler ? (hhp_sampler_is_w16(p, n_rows) ? "hh_k_policy_w16_ppo" : "hh_k_policy_ppo") : hhp_form_name(p, n_rows, one_side ? n_rows / 2 : -1));
    return HH_OK;
}

extern "C" int hh_policy_act(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, const uint8_t *sel, int8_t *actions,
                             float *logits, void *stream) {
    if (!p || !obs || !actions || n_rows <= 0 || obs_stride <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    if (!sel && p->binned_rows != n_rows) { g_err = "hh_policy_act: sel == NULL re-uses the row lists of the previous call, which had another n_rows"; return HH_E_ARG; }
    if (n_rows > p->max_rows) { g_err = "hh_policy_act: n_rows exceeds max_rows of hh_policy_create"; return HH_E_ARG; }
    if (p->n_nets == 0) { g_err = "hh_policy_act: no network loaded"; return HH_E_ARG; }
    HH_GUARD(p);
    hipStream_t st = (hipStream_t)stream;
    if (sel) { /* sel == NULL: the selectors are the ones of the previous call (a fixed network per unit slot): the lists stand */
        hipLaunchKernelGGL(hh_k_policy_bin, dim3((n_rows + 255) / 256), dim3(256), 0, st, n_rows, sel, p->lut, p->max_rows, p->counts, p->lists, actions);
        p->binned_rows = n_rows;
    }
    return hhp_launch_forward(p, obs, n_rows, obs_stride, actions, logits, sel ? HHP_CONSUME : HHP_FROM_SAVED, st);
}

/* the world's kernels bin the policy rows they emit (HighLevelEnv pilots; LowLevelEnv levels 4-5 opponents) into this bank's lists themselves */
extern "C" int hh_bind_policy(hh_world *w, hh_policy *p) {
    if (!w) { g_err = "null argument"; return HH_E_ARG; }
    if (w->cfg.env_kind != HH_ENV_HIGHLEVEL && !w->dc.ext_opp) { g_err = "hh_bind_policy: a LowLevelEnv world flies policies at levels 4-5 only (ext_opp_actions)"; return HH_E_ARG; }
    if (w->bound_policy) hhp_unbind(w->bound_policy);
    if (!p) return HH_OK;
    if (p->bound) hhp_unbind(p); /* one world per bank: the row lists and counters are the bank's */
    if (p->device != w->device) { g_err = "hh_bind_policy: world and policy bank live on different devices"; return HH_E_ARG; }
    const long long rows = (long long)w->dc.N * (w->cfg.env_kind == HH_ENV_HIGHLEVEL ? w->dc.A : w->dc.nO);
    if ((long long)p->max_rows < rows) { g_err = "hh_bind_policy: the bank's max_rows is smaller than the rows of one policy call (n_arenas x 6, LowLevelEnv: x n_opps)"; return HH_E_ARG; }
    if (p->n_nets == 0) { g_err = "hh_bind_policy: no network loaded"; return HH_E_ARG; }
    HH_GUARD(w);
    HIPCHK(hipMemset(p->counts, 0, HHP_COUNTS_INTS * sizeof(int)));
    w->P.pol_lut = p->lut; w->P.pol_counts = p->counts; w->P.pol_lists = p->lists; w->P.pol_max_rows = p->max_rows;
    w->bound_policy = p; p->bound = w;
    p->binned_rows = 0; /* the lists now belong to the world's kernels: a later sel == NULL call of hh_policy_act must re-bin */
    return HH_OK;
}

extern "C" int hh_policy_act_binned(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, int8_t *actions, float *logits, void *stream) {
    if (!p || !obs || !actions || n_rows <= 0 || obs_stride <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    if (n_rows > p->max_rows) { g_err = "hh_policy_act_binned: n_rows exceeds max_rows of hh_policy_create"; return HH_E_ARG; }
    if (p->n_nets == 0) { g_err = "hh_policy_act_binned: no network loaded"; return HH_E_ARG; }
    HH_GUARD(p);
    const bool one_side = p->bound && p->bound->cfg.env_kind == HH_ENV_HIGHLEVEL;
    return hhp_launch_forward(p, obs, n_rows, obs_stride, actions, logits, HHP_CONSUME, (hipStream_t)stream, one_side ? n_rows / 2 : -1);
}

/* the same with the caller's estimate of the rows that carry a network (the form is chosen by it): the variant-row phases list ~0.3 of their [N, 15] slots */
extern "C" int hh_policy_act_binned_live(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, int8_t *actions, float *logits, int32_t live_rows, void *stream) {
    if (!p || !obs || !actions || n_rows <= 0 || obs_stride <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    if (n_rows > p->max_rows) { g_err = "hh_policy_act_binned_live: n_rows exceeds max_rows of hh_policy_create"; return HH_E_ARG; }
    if (p->n_nets == 0) { g_err = "hh_policy_act_binned_live: no network loaded"; return HH_E_ARG; }
    HH_GUARD(p);
    return hhp_launch_forward(p, obs, n_rows, obs_stride, actions, logits, HHP_CONSUME, (hipStream_t)stream, live_rows);
}

/* ---- the value branches of the trainable policies + one sampler step (hh_policy_kernel_ppo.h) ---- */
static const int HHC_DIMS[4][4] = {
    /* own obs, own act, other obs, other act: train_hetero.py:183-198 observer spaces */
    {26, 4, 24, 3}, {24, 3, 26, 4}, {30, 4, 29, 3}, {29, 3, 30, 4},
};
static int hhp_set_critic(hh_policy *p, int32_t slot, const hh_critic_weights *w);
extern "C" int hh_policy_set_critic(hh_policy *p, int32_t slot, const hh_critic_weights *w) {
    try {
        return hhp_set_critic(p, slot, w);
    } catch (const std::exception &e) {
        g_err = std::string("hh_policy_set_critic: ") + e.what();
        return HH_E_HIP;
    }
}
static int hhp_set_critic(hh_policy *p, int32_t slot, const hh_critic_weights *w) {
    if (!p || !w || slot < 0 || slot >= HH_POLICY_MAX_NETS || w->kind < 0 || w->kind > 3) { g_err = "bad argument"; return HH_E_ARG; }
    if (!p->blob[slot] || p->bank.net[slot].kind != w->kind) { g_err = "hh_policy_set_critic: load the actor of the same kind into this slot first (hh_policy_set_net)"; return HH_E_ARG; }
    const bool att = w->kind <= HH_NET_FIGHT2;
    if (!w->v_w[0] || !w->v_b[0] || !w->shared_w || !w->shared_b || !w->val_w || !w->val_b ||
        (att && (!w->v_w[1] || !w->v_b[1] || !w->v_w[2] || !w->v_b[2] || !w->att_in_proj_w || !w->att_in_proj_b || !w->att_out_w || !w->att_out_b))) {
        g_err = "hh_policy_set_critic: missing weight pointer"; return HH_E_ARG;
    }
    HH_GUARD(p);
    const int d1 = HHC_DIMS[w->kind][0], a1 = HHC_DIMS[w->kind][1], d2 = HHC_DIMS[w->kind][2], a2 = HHC_DIMS[w->kind][3];
    const int n_in = d1 + a1 + d2 + a2; /* 57 | 66 */
    /* fragment planes (halves): w1 [80 x 512] | wov [160 x 256] | ws [512 x 512] | wa [512 x 32]; then the float biases b1 | bs | bov | ba */
    const size_t h_w1 = 0, h_wov = h_w1 + (size_t)(HHC_XK / 16) * 2 * HHP_H * 8, h_ws = h_wov + (size_t)(HHC_ATT_W / 16) * 2 * HHC_ATT_J * 8,
                 h_wa = h_ws + (size_t)32 * 2 * HHP_H * 8, h_total = h_wa + (size_t)32 * 2 * HHP_OUT * 8;
    std::vector<uint16_t> Hh(h_total, 0), Hl(h_total, 0);
    const size_t o_b1 = 0, o_bs = 512, o_bov = 1024, o_ba = 1024 + HHC_ATT_W, n_bias = o_ba + HHP_OUT;
    std::vector<float> Bf(n_bias, 0.0f);
    /* hidden column of the reference's concatenation cat(v1, v2, y3) (ac_models_hetero.py:274-281) in the kernel's order y3 | pad | v1 | v2 */
    auto hcol = [att](int c) { return att ? (c < 350 ? HHC_V12_OFF + c : HHC_V3_OFF + (c - 350)) : c; };
    if (att) {
        const int in0[3] = {0, d1 + a1, 0}, in1[3] = {d1 + a1, n_in, n_in}, wd[3] = {175, 175, 150}, out0[3] = {0, 175, 350};
        for (int b = 0; b < 3; b++)
            for (int o = 0; o < wd[b]; o++) {
                for (int c = in0[b]; c < in1[b]; c++) hhp_split_put(Hh, Hl, h_w1, c, hcol(out0[b] + o), HHP_H, w->v_w[b][(size_t)o * (in1[b] - in0[b]) + (c - in0[b])]);
                Bf[o_b1 + hcol(out0[b] + o)] = w->v_b[b][o];
            }
        /* att_val over one key: out_proj(v_proj(t)), folded in double like the actor's */
        const float *wv = w->att_in_proj_w + (size_t)300 * 150, *bv = w->att_in_proj_b + 300;
        for (int j = 0; j < 150; j++) {
            for (int k = 0; k < 150; k++) {
                double s = 0.0;
                for (int m = 0; m < 150; m++) s += (double)w->att_out_w[(size_t)j * 150 + m] * (double)wv[(size_t)m * 150 + k];
                hhp_split_put(Hh, Hl, h_wov, k, j, HHC_ATT_J, (float)s);
            }
            double s = (double)w->att_out_b[j];
            for (int m = 0; m < 150; m++) s += (double)w->att_out_w[(size_t)j * 150 + m] * (double)bv[m];
            Bf[o_bov + j] = (float)s;
        }
    } else {
        for (int o = 0; o < 500; o++) {
            for (int c = 0; c < n_in; c++) hhp_split_put(Hh, Hl, h_w1, c, o, HHP_H, w->v_w[0][(size_t)o * n_in + c]);
            Bf[o_b1 + o] = w->v_b[0][o];
        }
    }
    for (int j = 0; j < 500; j++) {
        for (int k = 0; k < 500; k++) hhp_split_put(Hh, Hl, h_ws, hcol(k), j, HHP_H, w->shared_w[(size_t)j * 500 + k]);
        Bf[o_bs + j] = w->shared_b[j];
    }
    for (int k = 0; k < 500; k++) hhp_split_put_t(Hh, Hl, h_wa, k, 0, HHP_OUT, w->val_w[k]);
    Bf[o_ba] = w->val_b[0];
    { /* the same value branch as ONE linear stream of 1 KB fragments for hh_k_policy_w16_ppo (chunk order of hhx_critic_tile) + its biases */
        std::vector<float> W1d((size_t)96 * 512, 0.0f), Wovd((size_t)160 * 160, 0.0f), Wsd((size_t)512 * 512, 0.0f);
        if (att) {
            const int in0[3] = {0, d1 + a1, 0}, in1[3] = {d1 + a1, n_in, n_in}, wd[3] = {175, 175, 150}, out0[3] = {0, 175, 350};
            for (int b = 0; b < 3; b++)
                for (int o = 0; o < wd[b]; o++)
                    for (int c = in0[b]; c < in1[b]; c++) W1d[(size_t)c * 512 + hcol(out0[b] + o)] = w->v_w[b][(size_t)o * (in1[b] - in0[b]) + (c - in0[b])];
            const float *wv = w->att_in_proj_w + (size_t)300 * 150;
            for (int j = 0; j < 150; j++)
                for (int k = 0; k < 150; k++) {
                    double sacc = 0.0;
                    for (int m = 0; m < 150; m++) sacc += (double)w->att_out_w[(size_t)j * 150 + m] * (double)wv[(size_t)m * 150 + k];
                    Wovd[(size_t)k * 160 + j] = (float)sacc;
                }
        } else {
            for (int o = 0; o < 500; o++)
                for (int c = 0; c < n_in; c++) W1d[(size_t)c * 512 + o] = w->v_w[0][(size_t)o * n_in + c];
        }
        for (int j = 0; j < 500; j++)
            for (int k = 0; k < 500; k++) Wsd[(size_t)hcol(k) * 512 + j] = w->shared_w[(size_t)j * 500 + k];
        std::vector<uint16_t> X((size_t)HHXC_STREAM_PIECES * (HHW_PIECE / 2), 0);
        for (int T = 0; T < 32; T++) /* chunk c = T / 4: tile (T & 3), k-block kb: hi, lo */
            for (int kb = 0; kb < 3; kb++)
                for (int wq = 0; wq < 32; wq++)
                    for (int c = 0; c < 16; c++) hhx_put(X, (size_t)(T * 3 + kb) * 2, wq, c, true, W1d[(size_t)(32 * kb + wq) * 512 + 16 * T + c]);
        for (int j = 0; j < 10; j++)
            for (int kb = 0; kb < 5; kb++)
                for (int wq = 0; wq < 32; wq++)
                    for (int c = 0; c < 16; c++) hhx_put(X, (size_t)HHXC_L1_PIECES + (size_t)(j * 5 + kb) * 2, wq, c, false, Wovd[(size_t)(32 * kb + wq) * 160 + 16 * j + c]);
        for (int pp = 0; pp < 8; pp++)
            for (int q = 0; q < 4; q++)
                for (int kk = 0; kk < 4; kk++)
                    for (int t = 0; t < 4; t++)
                        for (int wq = 0; wq < 32; wq++)
                            for (int c = 0; c < 16; c++)
                                hhx_put(X, (size_t)HHXC_L1_PIECES + HHXC_ATT_PIECES + (size_t)((pp * 4 + q) * 16 + kk * 4 + t) * 2, wq, c, false,
                                        Wsd[(size_t)(32 * (4 * q + kk) + wq) * 512 + 16 * (4 * pp + t) + c]);
        for (int kb = 0; kb < 16; kb++)
            for (int wq = 0; wq < 32; wq++) { const int k = 32 * kb + wq; hhx_put(X, (size_t)HHXC_L1_PIECES + HHXC_ATT_PIECES + HHX_L2_PIECES + (size_t)kb * 2, wq, 0, false, k < 500 ? w->val_w[k] : 0.0f); }
        const size_t xbytes = (size_t)HHXC_STREAM_PIECES * HHW_PIECE, xbias = (size_t)(512 + 512 + 160 + 32) * sizeof(float);
        if (!p->cxblob[slot]) HIPCHK(hipMalloc(&p->cxblob[slot], xbytes + xbias));
        HIPCHK(hipMemcpy(p->cxblob[slot], X.data(), xbytes, hipMemcpyHostToDevice));
        std::vector<float> Xb(512 + 512 + 160 + 32, 0.0f);
        for (int i = 0; i < 512; i++) { Xb[i] = Bf[o_b1 + i]; Xb[512 + i] = Bf[o_bs + i]; }
        for (int i = 0; i < 160; i++) Xb[1024 + i] = Bf[o_bov + i];
        Xb[1184] = Bf[o_ba];
        HIPCHK(hipMemcpy(p->cxblob[slot] + xbytes, Xb.data(), xbias, hipMemcpyHostToDevice));
        HhpCritX &Cx = p->cbankx.c[slot];
        const float *xb = reinterpret_cast<const float *>(p->cxblob[slot] + xbytes);
        Cx.stream = reinterpret_cast<const unsigned char *>(p->cxblob[slot]);
        Cx.b1 = xb; Cx.bs = xb + 512; Cx.bov = xb + 1024; Cx.ba = xb + 1184;
        Cx.d1 = d1; Cx.a1 = a1; Cx.d2 = d2; Cx.a2 = a2; Cx.has_att = att ? 1 : 0; Cx.loaded = 1;
    }
    const size_t plane_bytes = h_total * sizeof(uint16_t), bytes = 2 * plane_bytes + n_bias * sizeof(float);
    if (!p->cblob[slot]) HIPCHK(hipMalloc(&p->cblob[slot], bytes));
    char *d = p->cblob[slot];
    HIPCHK(hipMemcpy(d, Hh.data(), plane_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + plane_bytes, Hl.data(), plane_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + 2 * plane_bytes, Bf.data(), n_bias * sizeof(float), hipMemcpyHostToDevice));
    HhpCrit &Cn = p->cbank.c[slot];
    const uint16_t *bh = reinterpret_cast<const uint16_t *>(d), *bl = reinterpret_cast<const uint16_t *>(d + plane_bytes);
    const float *bf = reinterpret_cast<const float *>(d + 2 * plane_bytes);
    Cn.w1h = reinterpret_cast<const float4 *>(bh + h_w1); Cn.w1l = reinterpret_cast<const float4 *>(bl + h_w1);
    Cn.wovh = reinterpret_cast<const float4 *>(bh + h_wov); Cn.wovl = reinterpret_cast<const float4 *>(bl + h_wov);
    Cn.wsh = reinterpret_cast<const float4 *>(bh + h_ws); Cn.wsl = reinterpret_cast<const float4 *>(bl + h_ws);
    Cn.wah = reinterpret_cast<const float4 *>(bh + h_wa); Cn.wal = reinterpret_cast<const float4 *>(bl + h_wa);
    Cn.b1 = bf + o_b1; Cn.bs = bf + o_bs; Cn.bov = bf + o_bov; Cn.ba = bf + o_ba;
    Cn.d1 = d1; Cn.a1 = a1; Cn.d2 = d2; Cn.a2 = a2; Cn.has_att = att ? 1 : 0; Cn.loaded = 1;
    return HH_OK;
}

extern "C" int hh_policy_sample(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, const uint8_t *sel, hh_world *w,
                                const double *uniforms, const float *crit_act, int32_t greedy, int8_t *actions, float *logp, float *vf, float *logits,
                                void *stream) {
    if (!p || !obs || !actions || n_rows <= 0 || obs_stride <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    if (!sel && p->binned_rows != n_rows) { g_err = "hh_policy_sample: sel == NULL re-uses the row lists of the previous call, which had another n_rows"; return HH_E_ARG; }
    if (n_rows > p->max_rows) { g_err = "hh_policy_sample: n_rows exceeds max_rows of hh_policy_create"; return HH_E_ARG; }
    if (p->n_nets == 0) { g_err = "hh_policy_sample: no network loaded"; return HH_E_ARG; }
    if (!greedy && !uniforms && !w) { g_err = "hh_policy_sample: a draw needs the world (keyed RNG) or explicit uniforms"; return HH_E_ARG; }
    if (vf) {
        if (n_rows & 1) { g_err = "hh_policy_sample: the value branch pairs row r with row r ^ 1 (rows = [n_arenas, 2]): n_rows must be even"; return HH_E_ARG; }
        for (int n = 0; n < p->n_nets; n++)
            if (p->blob[n] && !p->cbank.c[n].loaded) { g_err = "hh_policy_sample: vf asked for, but a loaded network has no value branch (hh_policy_set_critic)"; return HH_E_ARG; }
    }
    HhpSampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.uniforms = uniforms; sa.crit_act = crit_act; sa.greedy = greedy ? 1 : 0;
    sa.actions = actions; sa.logp = logp; sa.vf = vf; sa.logits_out = logits;
    sa.rows_per_arena = 1;
    if (w && !uniforms && !greedy) {
        if (w->device != p->device) { g_err = "hh_policy_sample: world and policy bank live on different devices"; return HH_E_ARG; }
        if (n_rows % w->dc.N != 0) { g_err = "hh_policy_sample: n_rows is not a multiple of the world's arenas"; return HH_E_ARG; }
        sa.ar_pack = w->P.ar_pack; sa.seed = w->dc.seed; sa.arena_offset = w->dc.arena_offset; sa.rows_per_arena = n_rows / w->dc.N;
    }
    HH_GUARD(p);
    hipStream_t st = (hipStream_t)stream;
    if (sel) {
        hipLaunchKernelGGL(hh_k_policy_bin, dim3((n_rows + 255) / 256), dim3(256), 0, st, n_rows, sel, p->lut, p->max_rows, p->counts, p->lists, actions);
        p->binned_rows = n_rows;
    }
    if (hhp_sampler_is_w16(p, n_rows)) { /* weights through LDS, 64-row tiles: the form of large calls (hh_policy_kernel_w16.h) */
        const int tiles = (n_rows + 63) / 64 + p->n_nets;
        hipLaunchKernelGGL(hh_k_policy_w16_ppo, dim3(vf ? 2 * tiles : tiles), dim3(256), HHXC_LDS_BYTES, st, p->bank, p->bankx, p->cbankx, p->n_nets, obs, obs_stride,
                           p->counts, p->lists, p->max_rows, sa, vf ? 1 : 0, sel ? HHP_CONSUME : HHP_FROM_SAVED);
    } else {
        const int tiles = (n_rows + HHP_ROWS - 1) / HHP_ROWS + p->n_nets;
        hipLaunchKernelGGL(hh_k_policy_ppo, dim3(vf ? 2 * tiles : tiles), dim3(256), HHPP_LDS_BYTES, st, p->bank, p->bankh, p->cbank, p->n_nets, obs, obs_stride,
                           p->counts, p->lists, p->max_rows, sa, vf ? 1 : 0, sel ? HHP_CONSUME : HHP_FROM_SAVED);
    }
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* tuning probe: workgroups of a forward kernel form that the runtime reports co-resident per CU (which: 0 hh_k_policy_h<1>, 1 hh_k_policy_h<2>,
 * 2 retired, 3 hh_k_policy_w16, 4 hh_k_policy_ppo) */
extern "C" int hh_policy_occupancy(int32_t which, int32_t *blocks_per_cu) {
    if (!blocks_per_cu) return HH_E_ARG;
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(hh_k_policy_h<1>), 256, HHPH_LDS_BYTES(1));
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(hh_k_policy_h<2>), 512, HHPH_LDS_BYTES(2));
    else if (which == 2) { g_err = "hh_policy_occupancy: form 2 (hh_k_policy_w) was retired in round 6"; return HH_E_ARG; }
    else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(hh_k_policy_w16<4>), 256, HHX_LDS_BYTES);
    else if (which == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(hh_k_policy_ppo), 256, HHPP_LDS_BYTES);
    if (e != hipSuccess) { g_err = std::string("hh_policy_occupancy: ") + hipGetErrorString(e); return HH_E_HIP; }
    *blocks_per_cu = n;
    return HH_OK;
}

#ifdef HHP_PROFILE
extern "C" int hh_policy_prof_read(unsigned long long *out16, int reset) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hhp_prof), 16 * 8));
    if (reset) { unsigned long long z[48] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(hhp_prof), z, 48 * 8)); }
    return HH_OK;
}
extern "C" int hh_policy_prof_read_ppo(unsigned long long *out32) { /* the 2 x 16 phase timers of hh_k_policy_ppo's actor / critic tiles */
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(hhp_prof), 32 * 8, 16 * 8));
    return HH_OK;
}
#endif

#ifdef HH_TIMELINE
/* tuning builds: the workgroup timeline (hh_device.h: hh_tl).  out = [n][4] words; returns the number of entries written, clears the buffer's counter */
extern "C" long long hh_debug_timeline(unsigned long long *out, long long cap) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned long long n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(hh_tl), 8) != hipSuccess) return -1;
    if (n > HH_TL_CAP) n = HH_TL_CAP;
    if ((long long)n > cap) n = (unsigned long long)cap;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(hh_tl), n * 32, 32) != hipSuccess) return -1;
    const unsigned long long z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(hh_tl), &z, 8) != hipSuccess) return -1;
    return (long long)n;
}
#endif

#endif /* HH_POLICY_KERNEL_H */
