/*
 * hh_world.hip — C-ABI (include/hh_abi.h) host side of the batched air-combat world; the gfx950
 * kernels live in hh_kernels.h (one translation unit).
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "hh_kernels_hier.h"
#include "hh_kernels_quad.h"
#include "hh_kernels_oct.h"
#include "hh_gae.h"

/* ===================================================================== host side */
static thread_local std::string g_err;
extern "C" const char *hh_last_error(void) { return g_err.c_str(); }

#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                        \
            return HH_E_HIP;                                                               \
        }                                                                                  \
    } while (0)

/* Every entry point that launches or copies runs with the world's device current and restores the caller's device on
 * return (two worlds on two GPUs in one process; torch's current device may differ from the world's). */
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define HH_GUARD(w)                                                        \
    DeviceGuard guard_((w)->device);                                       \
    if (!guard_.ok) { g_err = "hipSetDevice(world device) failed"; return HH_E_HIP; }

struct hh_world {
    hh_config cfg;
    DevCfg dc;
    DevPtrs P;
    int device;
    int n_simd;   /* SIMDs of this device (multiProcessorCount x 4): one wave per SIMD up to here, then two */
    int block; /* threads per workgroup */
    void *slab;
    size_t slab_bytes;
    int *counter; /* device word: arenas still inside their macro step */
    int force_w;  /* 0 = choose the kernel variant by arena count, 1 / 2 = force (HH_FORCE_W, experiments/tests) */
    int no_quad;  /* HH_NO_QUAD=1: 2-vs-2 rollouts on the generic LDS-exchange kernel (A/B tests; same results) */
    int no_spec;  /* HH_NO_SPEC=1: never pick the instance compiled for the default level-3 configuration */
    int no_two;   /* HH_NO_TWO=1: never pick the two-wave (simulation + output wave) form for small worlds */
    int no_dual;  /* HH_NO_DUAL=1: the 8-arenas-per-wave 2-vs-2 form without helper lanes (A/B) */
    int no_oct;   /* HH_NO_OCT=1: HighLevelEnv macro steps on the LDS-exchange kernel instead of the register-exchange one (A/B) */
    int apw;      /* HH_APW=16: never pick the 8-arenas-per-wave form of the two-wave kernel */
    int no_owt;   /* HH_NO_OWT=1: general two-wave instances keep the pair table on the simulation wave (A/B; the presets always hand it to the output wave) */
    void *trace_mem; /* trajectory ring buffer + cursors (hh_trace_enable), separate allocation */
    struct hh_policy *bound_policy; /* hh_bind_policy: the bank whose row lists P.pol_* point into (it points back at this world) */
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int hh_world_create(const hh_config *cfg, int device, hh_world **out) {
    if (!cfg || !out) { g_err = "null argument"; return HH_E_ARG; }
    int A = cfg->n_agents + cfg->n_opps;
    if (cfg->n_arenas <= 0 || cfg->n_agents < 1 || cfg->n_opps < 1) { g_err = "unsupported configuration: need n_arenas, n_agents, n_opps > 0"; return HH_E_ARG; }
    if (cfg->reserved0 != 0) { g_err = "hh_config.reserved0 must be 0 (a caller built against an older hh_abi.h, whose struct had no opp_side_selector / reserved0 pair?)"; return HH_E_ARG; }
    if (cfg->env_kind == HH_ENV_HIGHLEVEL) {
        /* evaluation.py's n-vs-m scenarios (README.md:43): any 1..3 agents against 1..3 opponents live in the six unit slots of the
         * 3-vs-3 kernels — agents in slots 0..n_agents-1, opponents behind them, the remaining slots are never alive; with more than
         * three on a side (up to five) the arena has ten slots and runs on the LDS-exchange kernels (hh_k_hier<10, ...>) */
        if (cfg->n_agents > HH_SIDE_MAX || cfg->n_opps > HH_SIDE_MAX) { g_err = "HighLevelEnv: at most 5 aircraft per side"; return HH_E_ARG; }
        A = HH_HL_SLOTS(cfg->n_agents, cfg->n_opps);
    } else if (A != 4) {
        g_err = "LowLevelEnv is 2-vs-2 (envs/env_hetero.py:24-44)";
        return HH_E_ARG;
    }
    if (cfg->env_kind == HH_ENV_LOWLEVEL && (cfg->n_agents != 2 || cfg->n_opps != 2)) {
        g_err = "LowLevelEnv is 2-vs-2 (envs/env_hetero.py:24-44)";
        return HH_E_ARG;
    }
    if (cfg->env_kind == HH_ENV_LOWLEVEL && cfg->level >= 4 && !cfg->ext_opp_actions) {
        g_err = "levels 4-5 use frozen opponent policies: set ext_opp_actions and supply their actions";
        return HH_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device"; return HH_E_NODEV; }
    if (device < 0 || device >= ndev) { g_err = "bad device index"; return HH_E_ARG; }
    DeviceGuard guard_(device);
    if (!guard_.ok) { g_err = "hipSetDevice failed"; return HH_E_HIP; }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    hh_world *w = new (std::nothrow) hh_world();
    if (!w) { g_err = "hh_world_create: out of host memory"; return HH_E_HIP; }
    w->cfg = *cfg;
    w->device = device;
    w->n_simd = prop.multiProcessorCount * 4; /* 4 SIMDs per CU; follows the partition mode (SPX 256 CUs, CPX 32) */
    DevCfg &d = w->dc;
    d.N = cfg->n_arenas; d.env_kind = cfg->env_kind; d.nA = cfg->n_agents; d.nO = cfg->n_opps; d.A = A;
    d.level = cfg->level; d.agent_mode = cfg->agent_mode; d.horizon = cfg->horizon;
    d.friendly_kill = cfg->friendly_kill; d.friendly_punish = cfg->friendly_punish; d.esc_dist_rew = cfg->esc_dist_rew;
    d.hier_action_assess = cfg->hier_action_assess; d.hier_opp_fight_ratio = cfg->hier_opp_fight_ratio;
    d.auto_reset = cfg->auto_reset; d.ext_opp = cfg->ext_opp_actions;
    d.sel_side = (cfg->env_kind == HH_ENV_HIGHLEVEL && cfg->opp_side_selector) ? 1 : 0;
    d.D = cfg->env_kind == HH_ENV_HIGHLEVEL ? HH_OBS_HL : (cfg->agent_mode == HH_MODE_FIGHT ? HH_OBS_FIGHT_AC1 : HH_OBS_ESC_AC1);
    d.n_ctrl = cfg->ext_opp_actions ? A : cfg->n_agents;
    d.glob_frac = cfg->glob_frac; d.rew_scale = cfg->rew_scale;
    double ms = cfg->map_size;
    d.lat_hi = HH_MAP_LAT0 + ms; d.lon_hi = HH_MAP_LON0 + ms;
    d.ext_lat = d.lat_hi - HH_MAP_LAT0; d.ext_lon = d.lon_hi - HH_MAP_LON0;
    d.inv_ext_lat = 1.0 / d.ext_lat; d.inv_ext_lon = 1.0 / d.ext_lon;
    d.inv_diag = (1.0 - 0.0) / (__builtin_sqrt(2.0 * (ms * ms)) - 0.0);
    d.seed = cfg->seed; d.arena_offset = cfg->arena_offset;
    w->block = HH_BLOCK;
    w->trace_mem = nullptr;
    w->bound_policy = nullptr;
    { const char *fw = getenv("HH_FORCE_W"); w->force_w = fw ? atoi(fw) : 0; }
    { const char *e = getenv("HH_APW"); w->apw = e ? atoi(e) : 0; }
    { const char *nq = getenv("HH_NO_QUAD"); w->no_quad = nq ? atoi(nq) : 0; }
    { const char *ns = getenv("HH_NO_SPEC"); w->no_spec = ns ? atoi(ns) : 0; }
    { const char *nt = getenv("HH_NO_TWO"); w->no_two = nt ? atoi(nt) : 0; }
    { const char *no = getenv("HH_NO_OCT"); w->no_oct = no ? atoi(no) : 0; }
    { const char *nd = getenv("HH_NO_DUAL"); w->no_dual = nd ? atoi(nd) : 0; }
    { const char *e = getenv("HH_NO_OWT"); w->no_owt = e ? atoi(e) : 0; }
    /* one slab, 256-byte aligned sub-arrays */
    size_t U = (size_t)d.N * A, N = (size_t)d.N;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_f[6]; for (int k = 0; k < 6; k++) o_f[k] = take(U * 8);
    size_t o_pack = take(U * 16), o_tgt = take((A > 8 ? HH_TGT_K_WIDE : HH_TGT_K) * U * 8);
    size_t o_rk[4]; for (int k = 0; k < 4; k++) o_rk[k] = take(U * 8);
    size_t o_rkp = take(U * 8), o_ar = take(N * 16), o_ep = take(N * 8), o_lr = take(N * 4), o_ll = take(N * 4);
    size_t o_lo = take(N), o_ev = take(N * 4), o_acc = take(U * 8), o_cnt = take(256);
    size_t o_el = take(N * HH_EVAL_K * 4), o_et = take(N * HH_EVAL_K * 4), o_af = take(N * 4);
    w->slab_bytes = off;
    {
        hipError_t e = hipMalloc(&w->slab, off);
        if (e == hipSuccess) e = hipMemset(w->slab, 0, off);
        if (e != hipSuccess) {
            g_err = std::string("hipMalloc/hipMemset of the world slab: ") + hipGetErrorString(e);
            if (w->slab) (void)hipFree(w->slab);
            delete w;
            return HH_E_HIP;
        }
    }
    char *b = (char *)w->slab;
    DevPtrs &P = w->P;
    P.lat = (double *)(b + o_f[0]); P.lon = (double *)(b + o_f[1]); P.hdg = (double *)(b + o_f[2]);
    P.spd = (double *)(b + o_f[3]); P.cmd_hdg = (double *)(b + o_f[4]); P.cmd_spd = (double *)(b + o_f[5]);
    P.pack = (int4 *)(b + o_pack); P.tgt_d = (double *)(b + o_tgt);
    P.rk_lat = (double *)(b + o_rk[0]); P.rk_lon = (double *)(b + o_rk[1]); P.rk_hdg = (double *)(b + o_rk[2]); P.rk_cmd = (double *)(b + o_rk[3]);
    P.rk_pack = (int2 *)(b + o_rkp); P.ar_pack = (int4 *)(b + o_ar); P.ep_ret = (double *)(b + o_ep);
    P.last_ret = (float *)(b + o_lr); P.last_len = (int *)(b + o_ll); P.last_outcome = (int8_t *)(b + o_lo);
    P.ev_mask = (uint32_t *)(b + o_ev);
    P.act_fault = (uint32_t *)(b + o_af);
    P.acc_rew = (double *)(b + o_acc);
    P.eval_last = (int *)(b + o_el); P.eval_tot = (int *)(b + o_et);
    w->counter = (int *)(b + o_cnt);
    /* arenas start "done" (must be reset first); outcome 2 = no finished episode yet */
    {
        std::vector<int4> ar(N);
        for (size_t i = 0; i < N; i++) { ar[i].x = 0; ar[i].y = 0; ar[i].z = 1 << 16; ar[i].w = 0; }
        HIPCHK(hipMemcpy(P.ar_pack, ar.data(), N * 16, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(P.last_outcome, 2, N));
    }
    *out = w;
    return HH_OK;
}

static void hhp_forget_world(struct hh_policy *p); /* hh_policy_kernel.h */

extern "C" int hh_world_destroy(hh_world *w) {
    if (!w) return HH_E_ARG;
    if (w->bound_policy) hhp_forget_world(w->bound_policy);
    DeviceGuard guard_(w->device);
    (void)hipFree(w->slab);
    if (w->trace_mem) (void)hipFree(w->trace_mem);
    delete w;
    return HH_OK;
}

extern "C" int hh_trace_enable(hh_world *w, int32_t n_arenas, int32_t capacity) {
    if (!w || n_arenas < 0 || capacity < 0) { g_err = "bad argument"; return HH_E_ARG; }
    HH_GUARD(w);
    HIPCHK(hipDeviceSynchronize());
    if (w->trace_mem) { (void)hipFree(w->trace_mem); w->trace_mem = nullptr; }
    w->P.trace = nullptr; w->P.trace_pos = nullptr; w->P.trace_K = 0; w->P.trace_cap = 0;
    if (n_arenas == 0 || capacity == 0) return HH_OK; /* tracing off */
    const int K = n_arenas < w->dc.N ? n_arenas : w->dc.N;
    const size_t rows = (size_t)capacity * K * w->dc.A * HH_TRACE_F * sizeof(float), pos = align_up((size_t)K * sizeof(int), 256);
    HIPCHK(hipMalloc(&w->trace_mem, pos + rows));
    HIPCHK(hipMemset(w->trace_mem, 0, pos + rows));
    w->P.trace_pos = (int *)w->trace_mem;
    w->P.trace = (float *)((char *)w->trace_mem + pos);
    w->P.trace_K = K; w->P.trace_cap = capacity;
    return HH_OK;
}

extern "C" int hh_trace_read(hh_world *w, float *rows, int32_t *count) {
    if (!w || !w->trace_mem) { g_err = "tracing is not enabled (hh_trace_enable)"; return HH_E_ARG; }
    HH_GUARD(w);
    HIPCHK(hipDeviceSynchronize());
    if (rows) HIPCHK(hipMemcpy(rows, w->P.trace, (size_t)w->P.trace_cap * w->P.trace_K * w->dc.A * HH_TRACE_F * sizeof(float), hipMemcpyDeviceToHost));
    if (count) HIPCHK(hipMemcpy(count, w->P.trace_pos, (size_t)w->P.trace_K * sizeof(int), hipMemcpyDeviceToHost));
    return HH_OK;
}

extern "C" int hh_obs_dim(const hh_world *w) { return w ? w->dc.D : HH_E_ARG; }
extern "C" int hh_n_ctrl(const hh_world *w) { return w ? w->dc.n_ctrl : HH_E_ARG; }

static int launch_hier(hh_world *w, int phase, const int8_t *cmd, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode,
                       float *obs, float *reward, uint8_t *valid, uint8_t *done, const uint8_t *mask, hipStream_t st) {
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_HIGHLEVEL || (c.A != 6 && c.A != 10)) { g_err = "not a HighLevelEnv world"; return HH_E_ARG; }
    HH_GUARD(w);
    if (c.A == 10) { /* more than three aircraft on a side: six ten-lane arenas per wave on the LDS-exchange kernel */
        hipLaunchKernelGGL((hh_k_hier<10, HH_BLOCK, 1>), dim3((c.N + 5) / 6), dim3(HH_BLOCK), 0, st, w->P, c, phase, cmd, actions, pilot_obs, pilot_mode, obs,
                           reward, valid, done, w->counter, mask);
        HIPCHK(hipGetLastError());
        return HH_OK;
    }
    if (!w->no_oct && phase <= HH_HL_END) { /* register-exchange form (hh_kernels_oct.h): one arena per 8-lane group */
        const int grid8 = (c.N + 7) / 8;
        const bool two8 = w->force_w == 2 || (w->force_w == 0 && grid8 > w->n_simd);
#define HH_LAUNCH_OCT(W_, PH_) hipLaunchKernelGGL((hh_k_hier_oct<W_, PH_>), dim3(grid8), dim3(64), 0, st, w->P, c, cmd, actions, pilot_obs, pilot_mode, obs, reward, valid, done, w->counter)
#define HH_LAUNCH_OCT_W(PH_) do { if (two8) HH_LAUNCH_OCT(2, PH_); else HH_LAUNCH_OCT(1, PH_); } while (0)
        switch (phase) {
        case HH_HL_BEGIN: HH_LAUNCH_OCT_W(HH_HL_BEGIN); break;
        case HH_HL_AGENTS_ACT: HH_LAUNCH_OCT_W(HH_HL_AGENTS_ACT); break;
        case HH_HL_TICK: HH_LAUNCH_OCT_W(HH_HL_TICK); break;
        default: HH_LAUNCH_OCT_W(HH_HL_END); break;
        }
#undef HH_LAUNCH_OCT_W
#undef HH_LAUNCH_OCT
        HIPCHK(hipGetLastError());
        return HH_OK;
    }
    constexpr int B = HH_BLOCK, GPB = B / 6;
    int grid = (c.N + GPB - 1) / GPB;
    /* the W = 2 instance stages the pilot rows per SIDE (three slots each): n-vs-m arenas use the W = 1 instance */
    const bool two = (w->force_w == 2 || (w->force_w == 0 && grid > w->n_simd)) && c.nA == 3 && c.nO == 3;
    if (two)
        hipLaunchKernelGGL((hh_k_hier<6, B, 2>), dim3(grid), dim3(B), 0, st, w->P, c, phase, cmd, actions, pilot_obs, pilot_mode, obs, reward,
                           valid, done, w->counter, mask);
    else
        hipLaunchKernelGGL((hh_k_hier<6, B, 1>), dim3(grid), dim3(B), 0, st, w->P, c, phase, cmd, actions, pilot_obs, pilot_mode, obs, reward,
                           valid, done, w->counter, mask);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

static int launch(hh_world *w, int run, int T, const int8_t *actions, const uint8_t *mask, float *obs, float *reward,
                  uint8_t *valid, uint8_t *done, hipStream_t st) {
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) {
        if (run == HH_RUN_ROLLOUT) { g_err = "HighLevelEnv steps through hh_hl_begin / hh_hl_agents_act / hh_hl_tick / hh_hl_end"; return HH_E_ARG; }
        return launch_hier(w, run == HH_RUN_RESET ? HH_HL_RESET : HH_HL_REFRESH, nullptr, nullptr, nullptr, nullptr, obs, nullptr, nullptr, nullptr, mask, st);
    }
    if (c.A != 4) { g_err = "LowLevelEnv worlds are 2-vs-2"; return HH_E_ARG; }
    HH_GUARD(w);
    constexpr int B = HH_BLOCK, GPB = B / 4;
    const int grid = (c.N + GPB - 1) / GPB;
    const int waves = grid * (B / 64);
    const bool two = w->force_w == 2 || (w->force_w == 0 && waves > w->n_simd); /* more waves than SIMDs: hold two per SIMD */
    if (run == HH_RUN_ROLLOUT && !w->no_quad && !w->P.trace) { /* tracing runs on the generic kernel */
        static_assert(B == 64, "the register-exchange kernel is one wave per workgroup");
        const int pre = w->no_spec ? 0 : hh_cfg_preset(c); /* 1: level 3 fight, 2 / 3: levels 1 / 2 fight, 4: level 3 escape (hh_kernels_quad.h) */
        /* small worlds (every workgroup resident with a SIMD pair to itself): simulation wave + output wave per 16 arenas */
        const bool pair = !two && !w->no_two && waves <= w->n_simd / 2; /* 128-thread groups at one wave per SIMD: two per CU resident */
        /* smaller still (the two-wave form of 8-arena groups fits one wave per SIMD): 8 arenas per simulation wave, hh_kernels_quad.h */
        const int grid8 = (c.N + 7) / 8;
        const bool half = pair && w->apw != 16 && 2 * grid8 <= w->n_simd;
#define HH_QLAUNCH(Wv, Pv, TWOv) hipLaunchKernelGGL((hh_k_world_quad<Wv, Pv, TWOv>), dim3(grid), dim3(TWOv ? 128 : 64), 0, st, w->P, c, T, actions, obs, reward, valid, done)
#define HH_QLAUNCH8(Pv) do { if (w->no_dual) hipLaunchKernelGGL((hh_k_world_quad<1, Pv, true, 8>), dim3(grid8), dim3(128), 0, st, w->P, c, T, actions, obs, reward, valid, done); \
                            else hipLaunchKernelGGL((hh_k_world_quad<1, Pv, true, 8, true>), dim3(grid8), dim3(128), 0, st, w->P, c, T, actions, obs, reward, valid, done); } while (0)
        /* general two-wave instances without the escape distance shaping: the pair table goes to the output wave like in the presets (hh_kernels_quad.h: SHAPE) */
        const bool noshape = pre == 0 && !w->no_owt && !(c.esc_dist_rew && c.agent_mode == HH_MODE_ESCAPE);
#define HH_QLAUNCH8_NS() do { if (w->no_dual) hipLaunchKernelGGL((hh_k_world_quad<1, 0, true, 8, false, false>), dim3(grid8), dim3(128), 0, st, w->P, c, T, actions, obs, reward, valid, done); \
                              else hipLaunchKernelGGL((hh_k_world_quad<1, 0, true, 8, true, false>), dim3(grid8), dim3(128), 0, st, w->P, c, T, actions, obs, reward, valid, done); } while (0)
#define HH_QPRE(LAUNCH) switch (pre) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; default: LAUNCH(0); }
#define HH_Q2(Pv) HH_QLAUNCH(2, Pv, false)
#define HH_Q1(Pv) HH_QLAUNCH(1, Pv, false)
        if (two) { HH_QPRE(HH_Q2) }
        else if (half && noshape) { HH_QLAUNCH8_NS(); }
        else if (half) { HH_QPRE(HH_QLAUNCH8) }
        else if (pair) { /* 4097..8192 arenas: the benchmark's preset only */
            if (pre == 1) HH_QLAUNCH(1, 1, true);
            else if (!w->no_owt && !(c.esc_dist_rew && c.agent_mode == HH_MODE_ESCAPE))
                hipLaunchKernelGGL((hh_k_world_quad<1, 0, true, 16, false, false>), dim3(grid), dim3(128), 0, st, w->P, c, T, actions, obs, reward, valid, done);
            else HH_QLAUNCH(1, 0, true);
        }
        else { HH_QPRE(HH_Q1) }
#undef HH_Q2
#undef HH_Q1
#undef HH_QPRE
#undef HH_QLAUNCH
#undef HH_QLAUNCH8
#undef HH_QLAUNCH8_NS
    } else if (run >= HH_RUN_LL_BEGIN)
        hipLaunchKernelGGL((hh_k_world<4, B, 1, true>), dim3(grid), dim3(B), 0, st, w->P, c, run, T, actions, mask, obs, reward, valid, done);
    else if (two)
        hipLaunchKernelGGL((hh_k_world<4, B, 2, false>), dim3(grid), dim3(B), 0, st, w->P, c, run, T, actions, mask, obs, reward, valid, done);
    else
        hipLaunchKernelGGL((hh_k_world<4, B, 1, false>), dim3(grid), dim3(B), 0, st, w->P, c, run, T, actions, mask, obs, reward, valid, done);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* which instance hh_rollout / hh_step launches for this world on this device (bench.py and profiles name it) */
extern "C" int hh_rollout_kernel_name(hh_world *w, char *buf, int32_t len) {
    if (!w || !buf || len <= 0) return HH_E_ARG;
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) {
        const int grid = (c.N + 9) / 10;
        const bool two = (w->force_w == 2 || (w->force_w == 0 && grid > w->n_simd)) && c.nA == 3 && c.nO == 3;
        if (c.A == 10) snprintf(buf, (size_t)len, "hh_k_hier<10,64,1>");
        else snprintf(buf, (size_t)len, "hh_k_hier<6,64,%d>", two ? 2 : 1);
        return HH_OK;
    }
    const int waves = (c.N + 15) / 16;
    const bool two = w->force_w == 2 || (w->force_w == 0 && waves > w->n_simd);
    int pre = w->no_spec ? 0 : hh_cfg_preset(c);
    const bool pair = !two && !w->no_two && waves <= w->n_simd / 2;
    const bool half = pair && w->apw != 16 && 2 * ((c.N + 7) / 8) <= w->n_simd;
    if (w->no_quad) snprintf(buf, (size_t)len, "hh_k_world<4,64,%d,false>", two ? 2 : 1);
    else {
        if (pair && !half && pre != 1) pre = 0;
        static const char *names[5] = {"general", "L3 fight", "L1 fight", "L2 fight", "L3 escape"};
        snprintf(buf, (size_t)len, "hh_k_world_quad<W=%d,preset=%s,%s%s>", two ? 2 : 1, names[pre],
                 pair ? "simulation wave + output wave" : "single wave", half ? (w->no_dual ? ",8 arenas per wave" : ",8 arenas per wave + helper lanes") : "");
    }
    return HH_OK;
}

/* the same instance the way a profiler prints it (demangled template arguments): what bench.py matches the committed rocprofv3
 * counter evidence against.  which = 0: hh_rollout / hh_step (LowLevelEnv) or the hh_hl_* phase launches; 1: hh_hl_rollout */
extern "C" int hh_kernel_instance(hh_world *w, int32_t which, char *buf, int32_t len) {
    if (!w || !buf || len <= 0) return HH_E_ARG;
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) {
        const int grid = (c.N + 9) / 10, grid8 = (c.N + 7) / 8;
        if (c.A == 10) {
            snprintf(buf, (size_t)len, which == 0 ? "hh_k_hier<10, 64, 1>" : "hh_k_hier_macro<10, 64, 1, false, 6>");
            return HH_OK;
        }
        if (which == 0) {
            const bool two = (w->force_w == 2 || (w->force_w == 0 && grid > w->n_simd)) && c.nA == 3 && c.nO == 3;
            if (!w->no_oct) snprintf(buf, (size_t)len, "hh_k_hier_oct<%d, phase>", (w->force_w == 2 || (w->force_w == 0 && grid8 > w->n_simd)) ? 2 : 1);
            else snprintf(buf, (size_t)len, "hh_k_hier<6, 64, %d>", two ? 2 : 1);
            return HH_OK;
        }
        const bool hld = !w->no_spec && hh_cfg_is_hl_default(c);
        if (!w->no_oct) {
            const bool two8 = w->force_w == 2 || (w->force_w == 0 && grid8 > w->n_simd);
            snprintf(buf, (size_t)len, "hh_k_hier_macro_oct<%d, %s>", two8 ? 2 : 1, hld ? "true" : "false");
            return HH_OK;
        }
        const bool two = w->force_w == 2 || (w->force_w == 0 && grid > w->n_simd);
        const bool half = !two && w->apw != 16 && grid8 <= w->n_simd;
        snprintf(buf, (size_t)len, "hh_k_hier_macro<6, 64, %d, %s, %d>", two ? 2 : 1, hld ? "true" : "false", half ? 8 : 10);
        return HH_OK;
    }
    const int waves = (c.N + 15) / 16;
    const bool two = w->force_w == 2 || (w->force_w == 0 && waves > w->n_simd);
    int pre = w->no_spec ? 0 : hh_cfg_preset(c);
    const bool pair = !two && !w->no_two && waves <= w->n_simd / 2;
    const bool half = pair && w->apw != 16 && 2 * ((c.N + 7) / 8) <= w->n_simd;
    if (w->no_quad || w->P.trace) snprintf(buf, (size_t)len, "hh_k_world<4, 64, %d, false>", two ? 2 : 1);
    else {
        if (pair && !half && pre != 1) pre = 0;
        const bool noshape = pair && pre == 0 && !w->no_owt && !(c.esc_dist_rew && c.agent_mode == HH_MODE_ESCAPE);
        /* six template arguments, as a profiler prints the instance: W, PRE, TWO, APW, DUAL, SHAPE */
        if (noshape) snprintf(buf, (size_t)len, "hh_k_world_quad<1, 0, true, %d, %s, false>", half ? 8 : 16, (half && !w->no_dual) ? "true" : "false");
        else if (half && !w->no_dual) snprintf(buf, (size_t)len, "hh_k_world_quad<1, %d, true, 8, true, true>", pre);
        else snprintf(buf, (size_t)len, "hh_k_world_quad<%d, %d, %s, %d, false, true>", two ? 2 : 1, pre, pair ? "true" : "false", half ? 8 : 16);
    }
    return HH_OK;
}

/* cumulative number of arena-ticks HighLevelEnv macro steps have run on this world (sub-steps of arenas that were
 * still inside their macro step) — the honest numerator of a ticks/s figure, since arenas leave a macro step early */
extern "C" int hh_hl_tick_count(hh_world *w, uint64_t *out /* [host] */, void *stream) {
    if (!w || !out) return HH_E_ARG;
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(out, (const char *)w->counter + 8, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HH_OK;
}

extern "C" int hh_reset(hh_world *w, const uint8_t *mask, float *obs, void *stream) {
    if (!w) return HH_E_ARG;
    return launch(w, HH_RUN_RESET, 1, nullptr, mask, obs, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int hh_step(hh_world *w, const int8_t *actions, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done, void *stream) {
    if (!w || !actions) { g_err = "null argument"; return HH_E_ARG; }
    return launch(w, HH_RUN_ROLLOUT, 1, actions, nullptr, obs, reward, reward_valid, done, (hipStream_t)stream);
}

extern "C" int hh_rollout(hh_world *w, int32_t n_steps, const int8_t *actions, float *obs, float *reward, uint8_t *reward_valid,
                          uint8_t *done, void *stream) {
    if (!w || !actions || n_steps <= 0) { g_err = "bad argument"; return HH_E_ARG; }
    return launch(w, HH_RUN_ROLLOUT, n_steps, actions, nullptr, obs, reward, reward_valid, done, (hipStream_t)stream);
}

extern "C" int hh_episode_stats(hh_world *w, float *ret, int32_t *len, int8_t *outcome, void *stream) {
    if (!w) return HH_E_ARG;
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream;
    size_t N = (size_t)w->dc.N;
    if (ret) HIPCHK(hipMemcpyAsync(ret, w->P.last_ret, N * 4, hipMemcpyDeviceToDevice, st));
    if (len) HIPCHK(hipMemcpyAsync(len, w->P.last_len, N * 4, hipMemcpyDeviceToDevice, st));
    if (outcome) HIPCHK(hipMemcpyAsync(outcome, w->P.last_outcome, N, hipMemcpyDeviceToDevice, st));
    return HH_OK;
}

/* [N,3] f32 block (return, length, outcome) in ONE small launch: what the multi-GPU logging all-gather moves */
__global__ __launch_bounds__(256) void hh_k_pack_stats(int N, const float *__restrict__ ret, const int *__restrict__ len,
                                                       const int8_t *__restrict__ outcome, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x; /* one lane per output float: unit-stride stores */
    if (i >= 3 * N) return;
    const int n = i / 3, k = i - 3 * n;
    out[i] = k == 0 ? ret[n] : (k == 1 ? (float)len[n] : (float)outcome[n]);
}

extern "C" int hh_episode_stats_packed(hh_world *w, float *out, void *stream) {
    if (!w || !out) { g_err = "null argument"; return HH_E_ARG; }
    HH_GUARD(w);
    const int N = w->dc.N;
    hipLaunchKernelGGL(hh_k_pack_stats, dim3((3 * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, w->P.last_ret, w->P.last_len,
                       w->P.last_outcome, out);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

extern "C" int hh_eval_info(hh_world *w, int32_t *last, int32_t *total, int32_t clear_total, void *stream) {
    if (!w) return HH_E_ARG;
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)w->dc.N * HH_EVAL_K * 4;
    if (last) HIPCHK(hipMemcpyAsync(last, w->P.eval_last, bytes, hipMemcpyDeviceToDevice, st));
    if (total) HIPCHK(hipMemcpyAsync(total, w->P.eval_tot, bytes, hipMemcpyDeviceToDevice, st));
    if (clear_total) HIPCHK(hipMemsetAsync(w->P.eval_tot, 0, bytes, st));
    return HH_OK;
}

__global__ __launch_bounds__(256) void hh_k_arena_status(DevPtrs P, DevCfg c, int *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= c.N) return;
    int ag = 0, op = 0;
    for (int s = 0; s < c.A; s++) {
        const int al = (P.pack[(size_t)n * c.A + s].z >> 8) & 0xff;
        if (s < c.nA) ag += al; else op += al;
    }
    const int4 a = P.ar_pack[n];
    reinterpret_cast<int4 *>(out)[n] = make_int4(a.x, ag, op, (ag <= 0 || op <= 0 || a.x >= c.horizon) ? 1 : 0);
}

extern "C" int hh_arena_status(hh_world *w, int32_t *out, void *stream) {
    if (!w || !out) { g_err = "null argument"; return HH_E_ARG; }
    HH_GUARD(w);
    hipLaunchKernelGGL(hh_k_arena_status, dim3((w->dc.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->P, w->dc, out);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

extern "C" int hh_get_event_masks(hh_world *w, uint32_t *masks, void *stream) {
    if (!w || !masks) return HH_E_ARG;
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream; /* ordered after the step on the caller's stream; waits for that stream only */
    HIPCHK(hipMemcpyAsync(masks, w->P.ev_mask, (size_t)w->dc.N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HH_OK;
}

/* the keyed synthetic action tape (hh_abi.h): one lane per action word, unit-stride 4-byte stores */
__global__ __launch_bounds__(256) void hh_k_action_tape(uint64_t seed, uint64_t arena_offset, int step0, int T, int N, int nU, int *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_t = (size_t)N * nU;
    if (i >= per_t * (size_t)T) return;
    const size_t t = i / per_t, r = i - t * per_t;
    const size_t n = r / (size_t)nU;
    const int s = (int)(r - n * (size_t)nU);
    out[i] = (int)hh_rng_action_word(seed, arena_offset + n, (uint32_t)(step0 + (int)t), (uint32_t)(s + 1));
}

extern "C" int hh_action_tape_uniform(uint64_t seed, uint64_t arena_offset, int32_t step0, int32_t T, int32_t N, int32_t n_units, int8_t *out, void *stream) {
    if (T <= 0 || N <= 0 || n_units <= 0 || step0 < 0 || !out || (reinterpret_cast<uintptr_t>(out) & 3)) { g_err = "hh_action_tape_uniform: bad argument"; return HH_E_ARG; }
    const size_t words = (size_t)T * N * n_units;
    hipLaunchKernelGGL(hh_k_action_tape, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, arena_offset, step0, T, N, n_units,
                       reinterpret_cast<int *>(out));
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* sticky per-arena "a step ran on a sanitised action word" flags (hh_abi.h) */
__global__ __launch_bounds__(256) void hh_k_action_faults(int N, uint32_t *__restrict__ flags, uint8_t *__restrict__ out, int clear) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    if (out) out[n] = flags[n] ? 1 : 0;
    if (clear) flags[n] = 0;
}

extern "C" int hh_action_faults(hh_world *w, uint8_t *out, int32_t clear, void *stream) {
    if (!w || (!out && !clear)) { g_err = "hh_action_faults: nothing to do (no output buffer, no clear)"; return HH_E_ARG; }
    HH_GUARD(w);
    hipLaunchKernelGGL(hh_k_action_faults, dim3((w->dc.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->dc.N, w->P.act_fault, out, clear);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

extern "C" int hh_get_state(hh_world *w, hh_state_view *v) {
    if (!w || !v) return HH_E_ARG;
    HH_GUARD(w);
    HIPCHK(hipDeviceSynchronize());
    const DevCfg &c = w->dc;
    size_t U = (size_t)c.N * c.A, N = (size_t)c.N;
    const int K = c.A > 8 ? HH_TGT_K_WIDE : HH_TGT_K; /* entries of a stored target list (hh_spec.h: HH_TGT_K_OF) */
    std::vector<double> f[6], tg((size_t)K * U), rk[4];
    std::vector<int4> pack(U), ar(N);
    std::vector<int2> rkp(U);
    const double *src[6] = {w->P.lat, w->P.lon, w->P.hdg, w->P.spd, w->P.cmd_hdg, w->P.cmd_spd};
    const double *rsrc[4] = {w->P.rk_lat, w->P.rk_lon, w->P.rk_hdg, w->P.rk_cmd};
    for (int k = 0; k < 6; k++) { f[k].resize(U); HIPCHK(hipMemcpy(f[k].data(), src[k], U * 8, hipMemcpyDeviceToHost)); }
    for (int k = 0; k < 4; k++) { rk[k].resize(U); HIPCHK(hipMemcpy(rk[k].data(), rsrc[k], U * 8, hipMemcpyDeviceToHost)); }
    HIPCHK(hipMemcpy(pack.data(), w->P.pack, U * 16, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tg.data(), w->P.tgt_d, (size_t)K * U * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rkp.data(), w->P.rk_pack, U * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ar.data(), w->P.ar_pack, N * 16, hipMemcpyDeviceToHost));
    for (size_t u = 0; u < U; u++) {
        for (int k = 0; k < 6; k++) v->ac_f[u * HH_ACF_K + k] = f[k][u];
        int4 p = pack[u];
        int32_t *q = v->ac_i + u * HH_ACI_K;
        int n_tgt = (p.z >> 24) & 0xff;
        q[0] = (p.z >> 8) & 0xff; q[1] = p.z & 0xff; q[2] = p.x & 0xffff; q[3] = p.y & 0xff; q[4] = (p.x >> 16) & 0xffff;
        q[5] = (p.y >> 8) & 0xff; q[6] = (p.y >> 16) & 0xff; q[7] = (p.y >> 24) & 0xff; q[8] = (p.z >> 16) & 0xff;
        q[9] = n_tgt ? (p.w & 0xff) : 0;
        int2 r = rkp[u];
        int ids[HH_TGT_K_WIDE] = {p.w & 0xff, (p.w >> 8) & 0xff, (p.w >> 16) & 0xff, (r.x >> 24) & 15, (r.x >> 28) & 15};
        for (int k = 0; k < K; k++) {
            v->tgt_id[u * K + k] = k < n_tgt ? ids[k] : 0;
            v->tgt_d[u * K + k] = k < n_tgt ? tg[(size_t)k * U + u] : 0.0;
        }
        int alive = r.x & 0xff;
        for (int k = 0; k < 4; k++) v->rk_f[u * HH_RKF_K + k] = alive ? rk[k][u] : 0.0;
        int32_t *rp = v->rk_i + u * HH_RKI_K;
        rp[0] = alive; rp[1] = alive ? (r.x >> 8) & 0xff : 0; rp[2] = alive ? (r.x >> 16) & 0xff : 0; rp[3] = alive ? r.y : 0;
    }
    for (size_t n = 0; n < N; n++) {
        int ag = 0, op = 0;
        for (int s = 0; s < c.A; s++) {
            int al = (pack[n * c.A + s].z >> 8) & 0xff;
            if (s < c.nA) ag += al; else op += al;
        }
        int32_t *ai = v->ar_i + n * HH_ARI_K;
        ai[0] = ar[n].x; ai[1] = ag; ai[2] = op; ai[3] = ar[n].z & 0xff; ai[4] = (int)(int8_t)((ar[n].z >> 8) & 0xff); ai[5] = ar[n].y;
    }
    return HH_OK;
}

extern "C" int hh_set_state(hh_world *w, const hh_state_view *v) {
    if (!w || !v) return HH_E_ARG;
    HH_GUARD(w);
    HIPCHK(hipDeviceSynchronize());
    const DevCfg &c = w->dc;
    size_t U = (size_t)c.N * c.A, N = (size_t)c.N;
    const int K = c.A > 8 ? HH_TGT_K_WIDE : HH_TGT_K;
    std::vector<double> f[6], tg((size_t)K * U), rk[4];
    std::vector<int4> pack(U), ar(N);
    std::vector<int2> rkp(U);
    for (int k = 0; k < 6; k++) f[k].resize(U);
    for (int k = 0; k < 4; k++) rk[k].resize(U);
    for (size_t u = 0; u < U; u++) { /* the step keeps headings in [0, 360] and its one-turn modulo (hh_pymod_turn: -m <= x < 2 m) relies on it.  360.0
                                        itself is a value the step produces — Python's (tiny negative) % 360 rounds to 360.0 (ac1.py:94) — so a state read
                                        with hh_get_state must be accepted here */
        const double h = v->ac_f[u * HH_ACF_K + 2], ch = v->ac_f[u * HH_ACF_K + 4];
        if (!(h >= 0.0 && h <= 360.0 && ch >= 0.0 && ch <= 360.0)) { g_err = "hh_set_state: heading / commanded heading outside [0, 360]"; return HH_E_ARG; }
    }
    for (size_t u = 0; u < U; u++) {
        for (int k = 0; k < 6; k++) f[k][u] = v->ac_f[u * HH_ACF_K + k];
        const int32_t *q = v->ac_i + u * HH_ACI_K;
        int n_tgt = 0, ids[HH_TGT_K_WIDE] = {0, 0, 0, 0, 0};
        for (int k = 0; k < K; k++) {
            ids[k] = v->tgt_id[u * K + k];
            tg[(size_t)k * U + u] = v->tgt_d[u * K + k];
            if (ids[k]) n_tgt = k + 1;
        }
        int4 p;
        p.x = (q[2] & 0xffff) | ((q[4] & 0xffff) << 16);
        p.y = (q[3] & 0xff) | ((q[5] & 0xff) << 8) | ((q[6] & 0xff) << 16) | ((q[7] & 0xff) << 24);
        p.z = (q[1] & 0xff) | ((q[0] & 0xff) << 8) | ((q[8] & 0xff) << 16) | ((n_tgt & 0xff) << 24);
        p.w = (ids[0] & 0xff) | ((ids[1] & 0xff) << 8) | ((ids[2] & 0xff) << 16);
        pack[u] = p;
        for (int k = 0; k < 4; k++) rk[k][u] = v->rk_f[u * HH_RKF_K + k];
        const int32_t *rp = v->rk_i + u * HH_RKI_K;
        int2 r;
        r.x = (rp[0] & 0xff) | ((rp[1] & 0xff) << 8) | ((rp[2] & 0xff) << 16) | (int)(((unsigned)(ids[3] & 15) << 24) | ((unsigned)(ids[4] & 15) << 28));
        r.y = rp[3];
        rkp[u] = r;
    }
    for (size_t n = 0; n < N; n++) {
        const int32_t *ai = v->ar_i + n * HH_ARI_K;
        int ag = 0, op = 0, next_seq = 0;
        for (int s = 0; s < c.A; s++) {
            int al = v->ac_i[(n * c.A + s) * HH_ACI_K] != 0;
            if (s < c.nA) ag += al; else op += al;
            int sq = v->rk_i[(n * c.A + s) * HH_RKI_K + 3];
            if (sq > next_seq) next_seq = sq;
        }
        int done = ag <= 0 || op <= 0 || ai[0] >= c.horizon;
        ar[n].x = ai[0]; ar[n].y = ai[5];
        ar[n].z = (ai[3] & 0xff) | ((ai[4] & 0xff) << 8) | (done << 16);
        ar[n].w = next_seq;
    }
    double *dst[6] = {w->P.lat, w->P.lon, w->P.hdg, w->P.spd, w->P.cmd_hdg, w->P.cmd_spd};
    double *rdst[4] = {w->P.rk_lat, w->P.rk_lon, w->P.rk_hdg, w->P.rk_cmd};
    for (int k = 0; k < 6; k++) HIPCHK(hipMemcpy(dst[k], f[k].data(), U * 8, hipMemcpyHostToDevice));
    for (int k = 0; k < 4; k++) HIPCHK(hipMemcpy(rdst[k], rk[k].data(), U * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->P.pack, pack.data(), U * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->P.tgt_d, tg.data(), (size_t)K * U * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->P.rk_pack, rkp.data(), U * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->P.ar_pack, ar.data(), N * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(w->P.ep_ret, 0, N * 8));
    /* refresh opp_to_attack exactly like state() would (observe-only pass, no output buffer) */
    int rc = launch(w, HH_RUN_OBSERVE, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    return HH_OK;
}

/* current observation without stepping (hh_abi.h: state()) */
extern "C" int hh_observe(hh_world *w, float *obs, void *stream) {
    if (!w) return HH_E_ARG;
    return launch(w, HH_RUN_OBSERVE, 1, nullptr, nullptr, obs, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

/* ---- HighLevelEnv macro step (envs/env_hier.py:114-140), split so that pilot inference runs between launches ---- */
extern "C" int hh_hl_begin(hh_world *w, const int8_t *commander_actions, float *pilot_obs, uint8_t *pilot_mode, void *stream) {
    if (!w || !commander_actions) { g_err = "null argument"; return HH_E_ARG; }
    return launch_hier(w, HH_HL_BEGIN, commander_actions, nullptr, pilot_obs, pilot_mode, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int hh_hl_agents_act(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, void *stream) {
    if (!w || !actions) { g_err = "null argument"; return HH_E_ARG; }
    return launch_hier(w, HH_HL_AGENTS_ACT, nullptr, actions, pilot_obs, pilot_mode, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int hh_hl_tick(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, int32_t *running, void *stream) {
    if (!w || !actions) { g_err = "null argument"; return HH_E_ARG; }
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream;
    if (running) HIPCHK(hipMemsetAsync(w->counter, 0, 4, st));
    int rc = launch_hier(w, HH_HL_TICK, nullptr, actions, pilot_obs, pilot_mode, nullptr, nullptr, nullptr, nullptr, nullptr, st);
    if (rc) return rc;
    if (running) { /* host-visible count of arenas still inside the macro step (synchronises the stream) */
        HIPCHK(hipMemcpyAsync(running, w->counter, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return HH_OK;
}

/* the variant-row form: one launch (and one policy call) per sub-step — hh_kernels_oct.h: oct_phase_body<W, true> */
static int launch_hier_v(hh_world *w, int phase, const int8_t *cmd, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, hipStream_t st) {
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_HIGHLEVEL || c.A != 6) { g_err = "the variant-row phases serve HighLevelEnv worlds of up to 3 aircraft per side"; return HH_E_ARG; }
    if (!pilot_obs || !pilot_mode) { g_err = "null argument"; return HH_E_ARG; }
    if (w->P.pol_lut && (long long)w->P.pol_max_rows < (long long)c.N * HH_HL_VROWS) {
        g_err = "the bound policy bank's max_rows is smaller than n_arenas x 15 (hh_hl_begin_variants / hh_hl_act_tick)"; return HH_E_ARG;
    }
    HH_GUARD(w);
    const int grid8 = (c.N + 7) / 8;
    /* one instance per phase, at most 256 registers and 16.6 KB of LDS a wave: eight waves fill a CU like a policy tile (hh_kernels_oct.h) */
    const int wgs = (grid8 + HHV_WPB - 1) / HHV_WPB;
    if (phase == HH_HL_BEGIN_V) hipLaunchKernelGGL((hh_k_hier_oct_v<2, HH_HL_BEGIN_V, HHV_WPB>), dim3(wgs), dim3(64 * HHV_WPB), 0, st, w->P, c, cmd, actions, pilot_obs, pilot_mode, w->counter);
    else hipLaunchKernelGGL((hh_k_hier_oct_v<2, HH_HL_ACT_TICK, HHV_WPB>), dim3(wgs), dim3(64 * HHV_WPB), 0, st, w->P, c, cmd, actions, pilot_obs, pilot_mode, w->counter);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

extern "C" int hh_hl_begin_variants(hh_world *w, const int8_t *commander_actions, float *pilot_obs, uint8_t *pilot_mode, void *stream) {
    if (!w || !commander_actions) { g_err = "null argument"; return HH_E_ARG; }
    return launch_hier_v(w, HH_HL_BEGIN_V, commander_actions, nullptr, pilot_obs, pilot_mode, (hipStream_t)stream);
}

extern "C" int hh_hl_act_tick(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, int32_t *running, void *stream) {
    if (!w || !actions) { g_err = "null argument"; return HH_E_ARG; }
    HH_GUARD(w);
    hipStream_t st = (hipStream_t)stream;
    if (running) HIPCHK(hipMemsetAsync(w->counter, 0, 4, st));
    int rc = launch_hier_v(w, HH_HL_ACT_TICK, nullptr, actions, pilot_obs, pilot_mode, st);
    if (rc) return rc;
    if (running) {
        HIPCHK(hipMemcpyAsync(running, w->counter, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return HH_OK;
}

extern "C" int hh_hl_end(hh_world *w, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done, void *stream) {
    if (!w) return HH_E_ARG;
    return launch_hier(w, HH_HL_END, nullptr, nullptr, nullptr, nullptr, obs, reward, reward_valid, done, nullptr, (hipStream_t)stream);
}

/* the whole commander step in ONE launch when the pilots' actions are resident before it starts */
extern "C" int hh_hl_rollout(hh_world *w, const int8_t *commander_actions, const int8_t *pilot_tape, float *obs, float *reward, uint8_t *reward_valid,
                             uint8_t *done, void *stream) {
    if (!w || !commander_actions || !pilot_tape) { g_err = "null argument"; return HH_E_ARG; }
    const DevCfg &c = w->dc;
    if (w->cfg.env_kind != HH_ENV_HIGHLEVEL || (c.A != 6 && c.A != 10)) { g_err = "not a HighLevelEnv world"; return HH_E_ARG; }
    HH_GUARD(w);
    if (c.A == 10) {
        hipLaunchKernelGGL((hh_k_hier_macro<10, HH_BLOCK, 1, false>), dim3((c.N + 5) / 6), dim3(HH_BLOCK), 0, (hipStream_t)stream, w->P, c, commander_actions,
                           pilot_tape, obs, reward, reward_valid, done, w->counter);
        HIPCHK(hipGetLastError());
        return HH_OK;
    }
    constexpr int B = HH_BLOCK, GPB = B / 6;
    const int grid = (c.N + GPB - 1) / GPB;
    const bool two = w->force_w == 2 || (w->force_w == 0 && grid > w->n_simd);
    hipStream_t st = (hipStream_t)stream;
    const bool hld = !w->no_spec && hh_cfg_is_hl_default(c); /* the instance compiled for the reference's default HighLevelEnv configuration */
    const int grid8 = (c.N + 7) / 8;
    const bool half = !two && w->apw != 16 && grid8 <= w->n_simd; /* 8 arenas per wave while every workgroup still has a SIMD of its own */
    if (!w->no_oct) { /* register-exchange form (hh_kernels_oct.h): one arena per 8-lane group */
        const bool two8 = w->force_w == 2 || (w->force_w == 0 && grid8 > w->n_simd);
#define HH_OLAUNCH(Wv, Dv) hipLaunchKernelGGL((hh_k_hier_macro_oct<Wv, Dv>), dim3(grid8), dim3(64), 0, st, w->P, c, commander_actions, pilot_tape, obs, reward, reward_valid, done, w->counter)
        if (two8) { if (hld) HH_OLAUNCH(2, true); else HH_OLAUNCH(2, false); }
        else { if (hld) HH_OLAUNCH(1, true); else HH_OLAUNCH(1, false); }
#undef HH_OLAUNCH
        HIPCHK(hipGetLastError());
        return HH_OK;
    }
#define HH_MLAUNCH(Wv, Dv) hipLaunchKernelGGL((hh_k_hier_macro<6, B, Wv, Dv>), dim3(grid), dim3(B), 0, st, w->P, c, commander_actions, pilot_tape, obs, reward, reward_valid, done, w->counter)
#define HH_MLAUNCH8(Dv) hipLaunchKernelGGL((hh_k_hier_macro<6, B, 1, Dv, 8>), dim3(grid8), dim3(B), 0, st, w->P, c, commander_actions, pilot_tape, obs, reward, reward_valid, done, w->counter)
    if (two) { if (hld) HH_MLAUNCH(2, true); else HH_MLAUNCH(2, false); }
    else if (half) { if (hld) HH_MLAUNCH8(true); else HH_MLAUNCH8(false); }
    else { if (hld) HH_MLAUNCH(1, true); else HH_MLAUNCH(1, false); }
#undef HH_MLAUNCH8
#undef HH_MLAUNCH
    HIPCHK(hipGetLastError());
    return HH_OK;
}

#ifdef HH_PROFILE_PHASES
extern "C" int hh_prof_read(unsigned long long *out16, int reset) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hh_prof_cycles), 24 * 8)); /* [host] 24 counters (hh_kernels.h) */
    if (reset) { unsigned long long z[24] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(hh_prof_cycles), z, 24 * 8)); }
    return HH_OK;
}
#endif

/* ---- LowLevelEnv levels 4-5: the step split around the frozen opponent policy (env_hetero.py:160-172) ---- */
extern "C" int hh_step_begin(hh_world *w, const int8_t *agent_actions, int32_t opp_mode, float *opp_obs, void *stream) {
    if (!w || !agent_actions) { g_err = "null argument"; return HH_E_ARG; }
    if (!w->cfg.ext_opp_actions) { g_err = "hh_step_begin needs ext_opp_actions (levels 4-5)"; return HH_E_ARG; }
    return launch(w, HH_RUN_LL_BEGIN, opp_mode, agent_actions, nullptr, opp_obs, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int hh_step_finish(hh_world *w, const int8_t *opp_actions, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done, void *stream) {
    if (!w || !opp_actions) { g_err = "null argument"; return HH_E_ARG; }
    if (!w->cfg.ext_opp_actions) { g_err = "hh_step_finish needs ext_opp_actions (levels 4-5)"; return HH_E_ARG; }
    return launch(w, HH_RUN_LL_FINISH, 1, opp_actions, nullptr, obs, reward, reward_valid, done, (hipStream_t)stream);
}

extern "C" int hh_opp_policy(hh_world *w, int8_t *k_out, void *stream) {
    if (!w || !k_out) { g_err = "null argument"; return HH_E_ARG; }
    HH_GUARD(w);
    hipLaunchKernelGGL(hh_k_opp_policy, dim3((w->dc.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->P, w->dc, k_out);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* ---- rollout post-processing: GAE over the [T, N, n_agents] tensors of hh_rollout ---- */
extern "C" int hh_gae(int32_t T, int32_t N, int32_t n_agents, const float *reward, const float *value, const uint8_t *valid,
                      const uint8_t *done, float gamma, float lam, float *adv, float *ret, void *stream) {
    if (T <= 0 || N <= 0 || n_agents <= 0 || !reward || !value || !valid || !done || !adv || !ret) { g_err = "bad argument"; return HH_E_ARG; }
    size_t cols = (size_t)N * n_agents;
    int grid = (int)((cols + 255) / 256);
    hipLaunchKernelGGL(hh_k_gae, dim3(grid), dim3(256), 0, (hipStream_t)stream, T, N, n_agents, reward, value, valid, done, gamma, lam, adv, ret);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

extern "C" int hh_gae_rllib(int32_t T, int32_t N, int32_t n_agents, const float *reward, const float *value, const uint8_t *done, double gamma,
                            double lam, float *adv, float *ret, void *stream) {
    if (T <= 0 || N <= 0 || n_agents <= 0 || !reward || !value || !done || !adv || !ret) { g_err = "bad argument"; return HH_E_ARG; }
    size_t cols = (size_t)N * n_agents;
    int grid = (int)((cols + 255) / 256);
    hipLaunchKernelGGL(hh_k_gae_rllib, dim3(grid), dim3(256), 0, (hipStream_t)stream, T, N, n_agents, reward, value, done, gamma, lam, adv, ret);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* hh_math_eval: the shared math headers on the device (test probe, include/hh_abi.h) */
__global__ void hh_k_math_eval(int fn, int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ o0, double *__restrict__ o1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r0 = 0.0, r1 = 0.0;
    switch (fn) {
        case 0: hh_sincos(x, &r0, &r1); break;
        case 1: r0 = hh_atan2(x, y); break;
        case 2: r0 = hh_acos(x); break;
        case 3: hh_sincosd(x, &r0, &r1); break;
        case 4: r0 = hh_atan2d(x, y); break;
        case 5: r0 = hh_pymod(x, y); break;
        case 6: r0 = hh_remainder(x, y); break;
        case 7: r0 = hh_fmod(x, y); break;
        case 8: r0 = hh_round3(x); break;
        case 9: r0 = hh_div_known(x, y, 1.0 / y); break;
        case 10: r0 = hh_pymod_turn(x, y); break;
        case 11: r0 = hh_sqrt(x); break;
        case 12: r0 = hh_clip(x, 0.0, 1.0); break;
        case 13: r0 = hh_clip(x, -y, y); break;
        case 14: d_geo_move(x, y, o0[i], o1[i], r0, r1); break;
        default: break;
    }
    o0[i] = r0;
    if (o1) o1[i] = r1;
}
extern "C" int hh_math_eval(int32_t fn, int32_t n, const double *a, const double *b, double *o0, double *o1, void *stream) {
    if (fn < 0 || fn > 14 || n <= 0 || !a || !o0 || (fn == 14 && (!b || !o1))) { g_err = "hh_math_eval: bad argument"; return HH_E_ARG; }
    hipLaunchKernelGGL(hh_k_math_eval, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fn, n, a, b, o0, o1);
    HIPCHK(hipGetLastError());
    return HH_OK;
}

/* commander_actions after _action_assess expanded them (agents: validated action, opponents: drawn fight target /
 * escape; env_hier.py:142-190) — what evaluation.py's eval_info counters read (env_base.py:91-107) */
extern "C" int hh_hl_commands(hh_world *w, int8_t *out /* [host] [N, A] */) {
    if (!w || !out) return HH_E_ARG;
    HH_GUARD(w);
    HIPCHK(hipDeviceSynchronize());
    size_t U = (size_t)w->dc.N * w->dc.A;
    std::vector<int4> pack(U);
    HIPCHK(hipMemcpy(pack.data(), w->P.pack, U * 16, hipMemcpyDeviceToHost));
    for (size_t u = 0; u < U; u++) out[u] = (int8_t)((pack[u].w >> 24) & 0xff);
    return HH_OK;
}

/* ---- frozen pilot / opponent networks (C ABI in include/hh_policy.h) ---- */
#include "hh_policy_kernel.h"
