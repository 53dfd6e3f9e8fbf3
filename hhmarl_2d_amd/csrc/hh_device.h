/*
 * hh_device.h — device-side world layout and the per-lane helpers of the gfx950 kernels.
 *
 * Thread mapping (all kernels): ONE LANE PER AIRCRAFT SLOT.  An arena with A aircraft occupies
 * A consecutive lanes ("group"); a 64-lane workgroup (one wave) holds GPB = 64 / A arenas
 * (16 for 2-vs-2, 10 for 3-vs-3).  Unit u = arena * A + slot, so every per-unit array is read
 * and written with unit-stride-1 addresses: fully coalesced 8-byte (double) and 16-byte (packed
 * ints) accesses per lane.  The rocket launched by slot s lives in rocket slot s (the reference
 * allows at most one missile in flight per aircraft: ac1.py:73).
 *
 * What one lane needs from the other aircraft of its arena is exchanged either through LDS arrays
 * indexed by thread id (hh_kernels.h, any arena size) or, for 2-vs-2 where an arena is an aligned
 * quad of lanes, through DPP quad permutes with the pair table in registers (hh_kernels_quad.h).
 * The id-ordered kill semantics of cmano_simulator.py:142-144 are resolved on integer masks only
 * (SURVEY.md App. A.2).
 */
#ifndef HH_DEVICE_H
#define HH_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hh_abi.h"
#include "hh_envelope.h"
#include "hh_geodesic.h"
#include "hh_math.h"
#include "hh_rng.h"
#include "hh_spec.h"

/* Block placement hints.  At one wave per SIMD a SKIPPED region costs a taken branch when its body sits in line (s_cbranch_execz over the body: ~55 cycles,
 * tools/ubench/issue.hip); with the body marked cold, hipcc's block placement moves it behind the loop and the usual path falls through a not-taken
 * s_cbranch_execnz.  HH_RARE / HH_USUAL say which way a test goes on most wave-ticks (any lane of the wave counts); layout only, same results.
 * -DHH_NO_RARE builds the unhinted layout for the A/B. */
#ifdef HH_NO_RARE
#define HH_RARE(x) (x)
#define HH_USUAL(x) (x)
#else
#define HH_RARE(x) __builtin_expect(!!(x), 0)
#define HH_USUAL(x) __builtin_expect(!!(x), 1)
#endif

/* threads per workgroup: one 64-lane wave = 16 arenas (2-vs-2) / 10 arenas (3-vs-3).  Single-wave groups
 * measured 15-20 % faster than 256-thread groups: the per-tick barriers then never wait for another wave. */
#ifndef HH_BLOCK
#define HH_BLOCK 64
#endif

struct DevCfg {
    int N, env_kind, nA, nO, A, level, agent_mode, horizon;
    int friendly_kill, friendly_punish, esc_dist_rew, hier_action_assess, hier_opp_fight_ratio;
    int auto_reset, ext_opp, D, n_ctrl;
    int sel_side; /* hh_config.opp_side_selector: an opponent's fight row selects "fight_*_opp" (env_base.py:387-390) */
    double glob_frac, rew_scale, ext_lat, ext_lon, inv_ext_lat, inv_ext_lon, lat_hi, lon_hi, inv_diag;
    uint64_t seed, arena_offset;
};

/* The reference's default HighLevelEnv configuration (config.py:17-54 with --mode 1: 3-vs-3 on the 0.5 deg map, horizon 500,
 * friendly fire on, action assessment on, opponents fight 75 % of the time, no reward sharing / scaling) as literals: kernels
 * instantiated with HLD = true read these instead of the kernel arguments, so the unused configurations' code and ~35 scalar
 * registers drop out (the 2-vs-2 kernel does the same for level 3, hh_kernels_quad.h).  The launcher picks that instance only when
 * hh_cfg_is_hl_default() finds every one of these fields equal in the world's own DevCfg. */
__host__ __device__ inline void hh_cfg_set_hl_default(DevCfg &c) {
    c.env_kind = 1; c.nA = 3; c.nO = 3; c.A = 6; c.horizon = 500;
    c.friendly_kill = 1; c.friendly_punish = 0; c.esc_dist_rew = 0; c.hier_action_assess = 1; c.hier_opp_fight_ratio = 75;
    c.ext_opp = 0; c.D = 34; c.n_ctrl = 3; c.sel_side = 0;
    c.glob_frac = 0.0; c.rew_scale = 1.0;
    c.ext_lat = 0.5; c.ext_lon = 0.5; c.inv_ext_lat = 2.0; c.inv_ext_lon = 2.0; c.lat_hi = 5.5; c.lon_hi = 7.5;
    c.inv_diag = 1.0 / __builtin_sqrt(0.5);
}
inline bool hh_cfg_is_hl_default(const DevCfg &c) {
    DevCfg d = c;
    hh_cfg_set_hl_default(d);
    return d.env_kind == c.env_kind && d.nA == c.nA && d.nO == c.nO && d.A == c.A && d.horizon == c.horizon && d.friendly_kill == c.friendly_kill &&
           d.friendly_punish == c.friendly_punish && d.esc_dist_rew == c.esc_dist_rew && d.hier_action_assess == c.hier_action_assess &&
           d.hier_opp_fight_ratio == c.hier_opp_fight_ratio && d.ext_opp == c.ext_opp && d.sel_side == c.sel_side && d.D == c.D && d.n_ctrl == c.n_ctrl &&
           d.glob_frac == c.glob_frac && d.rew_scale == c.rew_scale && d.ext_lat == c.ext_lat && d.ext_lon == c.ext_lon &&
           d.inv_ext_lat == c.inv_ext_lat && d.inv_ext_lon == c.inv_ext_lon && d.lat_hi == c.lat_hi && d.lon_hi == c.lon_hi && d.inv_diag == c.inv_diag;
}

/* struct-of-arrays world in HBM; U = N * A units */
struct DevPtrs {
    double *lat, *lon, *hdg, *spd, *cmd_hdg, *cmd_spd;   /* [U]  a1/a6 */
    int4 *pack;                                          /* [U]  small ints, 16 B */
    double *tgt_d;                                       /* [3][U] stored target distances ([5][U] in ten-slot worlds) */
    double *rk_lat, *rk_lon, *rk_hdg, *rk_cmd;           /* [U]  rocket slot s <- launcher slot s */
    int2 *rk_pack;                                       /* [U] */
    int4 *ar_pack;                                       /* [N]  steps, episode, flags, next_seq */
    double *ep_ret;                                      /* [N] */
    float *last_ret;                                     /* [N] */
    int *last_len;                                       /* [N] */
    int8_t *last_outcome;                                /* [N] */
    uint32_t *ev_mask;                                   /* [N] */
    uint32_t *act_fault;                                 /* [N] sticky: a step of this arena ran on a sanitised action word (hh_action_faults) */
    double *acc_rew;                                     /* [U] rewards accumulated over a HighLevelEnv macro step */
    int *eval_last, *eval_tot;                           /* [N][HH_EVAL_K] eval_info of the last commander step / summed since cleared */
    /* optional trajectory ring buffer (hh_trace_enable; cmano_simulator.py:125-130,159-162 record_unit_trace): the first trace_K
     * arenas append one row of HH_TRACE_F floats per unit after reset and after every tick; slot = cursor % trace_cap */
    float *trace;                                        /* [trace_cap][trace_K][A][HH_TRACE_F] or nullptr */
    int *trace_pos;                                      /* [trace_K] rows written so far (monotonic) */
    int trace_K, trace_cap;
    /* optional policy bank bound with hh_bind_policy: the HighLevelEnv phase kernels bin the pilot rows they emit by network
     * themselves (selector byte -> slot through pol_lut), so hh_policy_act_binned needs no binning pass of its own */
    const uint8_t *pol_lut;                              /* [256] or nullptr */
    int *pol_counts, *pol_lists;                         /* rows per network [8] at stride HH_BIN_STRIDE (+ the bank's tickets), [8][pol_max_rows] */
    int pol_max_rows;
};


/* tuning builds only (-DHH_TIMELINE): every workgroup of the commander step's two kernels leaves (tag, CU, start, end) in s_memrealtime ticks (100 MHz) —
 * the true placement of concurrent streams, which a profiler's serialised trace does not show (tools/timeline.py) */
#ifdef HH_TIMELINE
#define HH_TL_CAP (1 << 20)
__device__ unsigned long long hh_tl[4 + 4 * (size_t)HH_TL_CAP];
struct HhTl { unsigned long long t0, marks; };
__device__ __forceinline__ void hh_tl_begin(HhTl &t) { t.t0 = __builtin_amdgcn_s_memrealtime(); t.marks = 0; }
/* up to four marks inside the workgroup: 20 ns units since the start, 11 bits each */
__device__ __forceinline__ void hh_tl_mark(HhTl &t, int k) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); /* what the phase asked for has arrived: the mark is the phase's end, not its issue */
    const unsigned long long d = __builtin_amdgcn_s_memrealtime() - t.t0;
    t.marks |= ((d >> 1) > 2047 ? 2047ULL : (d >> 1)) << (11 * k);
}
__device__ __forceinline__ void hh_tl_end(const HhTl &t, unsigned tag, unsigned aux) {
    if ((threadIdx.x & 63) != 0) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long at = atomicAdd(&hh_tl[0], 1ULL);
    if (at < HH_TL_CAP) {
        unsigned long long *e = &hh_tl[4 + 4 * at];
        e[0] = (unsigned long long)tag | ((unsigned long long)(threadIdx.x >> 6) << 8) | ((unsigned long long)aux << 32);
        e[1] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32);
        e[2] = t.t0;
        e[3] = ((t1 - t.t0) & 0xfffffULL) | (t.marks << 20); /* duration (20 bits of 10 ns) | the marks */
    }
}
#else
struct HhTl {};
__device__ __forceinline__ void hh_tl_begin(HhTl &) {}
__device__ __forceinline__ void hh_tl_mark(HhTl &, int) {}
__device__ __forceinline__ void hh_tl_end(const HhTl &, unsigned, unsigned) {}
#endif

/* register-resident state of one aircraft slot (+ its rocket slot) */
struct Unit {
    double lat, lon, hdg, spd, cmd_hdg, cmd_spd;
    int cannon_remain, cannon_max, burst, missile_remain, rocket_max, missile_wait;
    int ac_type, alive, has_missile, n_tgt, tgt0, tgt1, tgt2, cmd_act;
    double tgt_d0, tgt_d1, tgt_d2;
    int rk_alive, rk_target, rk_life, rk_seq;
    double rk_lat, rk_lon, rk_hdg, rk_cmd;
};
/* ten-slot HighLevelEnv arenas (more than 3 aircraft on a side): entries 4 and 5 of an opponent's stored list of agents (env_hier.py:97).
 * Only the kernels instantiated for A = 10 hold this type; they hand it to the shared device functions as a Unit & and the three places
 * that need the extra entries (hl_target_slot<true>, hl_pilot_obs, hl_commander_obs, unit_load / unit_store<true>) cast back */
struct UnitW : Unit {
    int tgt3, tgt4;
    double tgt_d3, tgt_d4;
};

/* register-resident arena scalars, replicated on every lane of the group */
struct Arena {
    int steps, episode, escaping, escaping_time, done, next_seq;
    int hl_s, hl_run; /* HighLevelEnv macro step: sub-step counter, still running (env_hier.py:125) */
    uint64_t akey;
    uint64_t tkey; /* hh_rng_tick_key(akey, episode, steps), refreshed whenever steps/episode change */
};

__device__ __forceinline__ void arena_rekey(Arena &a) { a.tkey = hh_rng_tick_key(a.akey, (uint32_t)a.episode, (uint32_t)a.steps); }

/* Workgroup barrier for LDS hand-overs.  Every kernel of this path runs ONE wave per workgroup (B = 64): the LDS unit executes a
 * wave's instructions in order, so a hand-over between lanes only needs the compiler kept from moving the accesses and the returned
 * data waited for — s_waitcnt lgkmcnt(0).  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: the vmcnt(0) also waits
 * for every global load and store in flight (the action word requested a sub-step ahead, the output rows of the previous phase),
 * once per barrier, ~25 times per tick.  Barriers that order GLOBAL accesses (ev_mask clear -> atomicOr) stay __syncthreads(). */
template <int B>
__device__ __forceinline__ void hh_wg_sync() {
    if constexpr (B == 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}
template <int B>
__device__ __forceinline__ int hh_wg_sync_or(int x) {
    if constexpr (B == 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return __any(x); }
    else return __syncthreads_or(x);
}

/* rows -> per-network lists, one atomic ROUND TRIP per wave (the scheme of hh_k_policy_bin): lane n - 1 carries the wave's count for
 * network n, the (up to eight) atomics leave as one instruction, every row takes its slot from the base of its network plus its rank
 * in that network's ballot.  Two halves so that the caller can put work (its output stores) between the request and the use of the
 * returned base.  Call from wave-uniform control flow; slot = 0 for lanes without a row. */
#define HH_BIN_STRIDE 32 /* ints between two counters: each on its own 128-byte line (atomics on one line serialise across addresses) */
struct HhBinTicket { int base, rank; };
__device__ __forceinline__ HhBinTicket hh_bin_rows_issue(int *__restrict__ counts, int slot) {
    const int lane = threadIdx.x & 63;
    int mine = 0;
    HhBinTicket t{0, 0};
#pragma unroll
    for (int n = 1; n <= 8; n++) {
        const unsigned long long m = __ballot(slot == n);
        if (lane == n - 1) mine = __popcll(m);
        if (slot == n) t.rank = __popcll(m & ((1ULL << lane) - 1ULL));
    }
    if (lane < 8 && mine) t.base = atomicAdd(&counts[lane * HH_BIN_STRIDE], mine);
    return t;
}
__device__ __forceinline__ void hh_bin_rows_finish(const HhBinTicket &t, int *__restrict__ lists, int max_rows, int row, int slot) {
    const int base = __shfl(t.base, slot > 0 ? slot - 1 : 0);
    if (slot > 0 && base + t.rank < max_rows) lists[(size_t)(slot - 1) * max_rows + base + t.rank] = row;
}

/* the same for lanes that list up to FOUR rows of one network (variant rows of hh_k_hier_oct_v): vmask bit v = the lane lists a row in round v;
 * a network's rows of round v follow its rows of the rounds before.  Still one atomic round trip per wave. */
struct HhBinTicket4 { int base; int rank[4]; };
__device__ __forceinline__ HhBinTicket4 hh_bin_rows_issue4(int *__restrict__ counts, int slot, int vmask) {
    const int lane = threadIdx.x & 63;
    int mine = 0;
    HhBinTicket4 t{0, {0, 0, 0, 0}};
#pragma unroll
    for (int n = 1; n <= 8; n++) {
        int acc = 0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const unsigned long long m = __ballot(slot == n && ((vmask >> v) & 1));
            if (slot == n) t.rank[v] = acc + __popcll(m & ((1ULL << lane) - 1ULL));
            acc += __popcll(m);
        }
        if (lane == n - 1) mine = acc;
    }
    if (lane < 8 && mine) t.base = atomicAdd(&counts[lane * HH_BIN_STRIDE], mine);
    return t;
}
__device__ __forceinline__ void hh_bin_rows_finish4(const HhBinTicket4 &t, int *__restrict__ lists, int max_rows, int row0, int slot, int vmask) {
    const int base = __shfl(t.base, slot > 0 ? slot - 1 : 0);
#pragma unroll
    for (int v = 0; v < 4; v++)
        if (slot > 0 && ((vmask >> v) & 1) && base + t.rank[v] < max_rows) lists[(size_t)(slot - 1) * max_rows + base + t.rank[v]] = row0 + v;
}

/* The lane's action word -> its four components, sanitised where it is loaded (hh_spec.h: hh_action_sanitize — heading / speed component
 * clamped to their ranges, fire components as booleans); `fault` collects "a component of a CONSUMED word was out of range" for
 * hh_act_fault_commit.  consumed = the lane's unit is alive in a running arena and it is its side's turn: exactly the rows the reference
 * hands to _take_base_action (rows of dead units / finished arenas / the other side may hold anything and never raise a fault). */
__device__ __forceinline__ void hh_act_unpack(int w, int8_t (&act)[4], int &fault, bool consumed) {
    int bad = 0;
    const uint32_t q = hh_action_sanitize((uint32_t)w, &bad);
    fault |= consumed ? bad : 0;
    act[0] = (int8_t)(q & 0xff); act[1] = (int8_t)((q >> 8) & 0xff); act[2] = (int8_t)((q >> 16) & 0xff); act[3] = (int8_t)((q >> 24) & 0xff);
}
/* once per launch, rare body: the arena's sticky flag (never cleared by a step; hh_action_faults reads / clears it) */
__device__ __forceinline__ void hh_act_fault_commit(const DevPtrs &P, int n, bool owner, int fault) {
    if (owner && fault) atomicOr(&P.act_fault[n], 1u);
}

/* one trace row of the lane's unit: lat, lon, heading, speed, alive, rocket lat, rocket lon, rocket alive + 16 * episode */
__device__ __forceinline__ void trace_append(const DevPtrs &P, int A, int n, int s, const Unit &m, const Arena &ar, int &cursor) {
    if (P.trace == nullptr || n >= P.trace_K) return;
    float4 *row = reinterpret_cast<float4 *>(P.trace + (((size_t)(cursor % P.trace_cap) * P.trace_K + n) * A + s) * HH_TRACE_F);
    row[0] = make_float4((float)m.lat, (float)m.lon, (float)m.hdg, (float)m.spd);
    row[1] = make_float4((float)m.alive, (float)m.rk_lat, (float)m.rk_lon, (float)(m.rk_alive + 16 * ar.episode));
    cursor += 1;
}

/* X: the world's arenas have ten unit slots — target-list entries 4 and 5 travel in the free top byte of rk_pack.x (two 4-bit ids) and
 * in planes 3 and 4 of tgt_d (allocated for such worlds only) */
template <bool X = false>
__device__ __forceinline__ void unit_load(const DevPtrs &P, size_t U, size_t u, Unit &m) {
    m.lat = P.lat[u]; m.lon = P.lon[u]; m.hdg = P.hdg[u]; m.spd = P.spd[u];
    m.cmd_hdg = P.cmd_hdg[u]; m.cmd_spd = P.cmd_spd[u];
    int4 p = P.pack[u];
    m.cannon_remain = p.x & 0xffff; m.cannon_max = (p.x >> 16) & 0xffff;
    m.burst = p.y & 0xff; m.missile_remain = (p.y >> 8) & 0xff; m.rocket_max = (p.y >> 16) & 0xff;
    m.missile_wait = (p.y >> 24) & 0xff;
    m.ac_type = p.z & 0xff; m.alive = (p.z >> 8) & 0xff; m.has_missile = (p.z >> 16) & 0xff; m.n_tgt = (p.z >> 24) & 0xff;
    m.tgt0 = p.w & 0xff; m.tgt1 = (p.w >> 8) & 0xff; m.tgt2 = (p.w >> 16) & 0xff; m.cmd_act = (int)(int8_t)((p.w >> 24) & 0xff);
    m.tgt_d0 = P.tgt_d[u]; m.tgt_d1 = P.tgt_d[U + u]; m.tgt_d2 = P.tgt_d[2 * U + u];
    int2 r = P.rk_pack[u];
    m.rk_alive = r.x & 0xff; m.rk_target = (r.x >> 8) & 0xff; m.rk_life = (r.x >> 16) & 0xff; m.rk_seq = r.y;
    m.rk_lat = P.rk_lat[u]; m.rk_lon = P.rk_lon[u]; m.rk_hdg = P.rk_hdg[u]; m.rk_cmd = P.rk_cmd[u];
    if constexpr (X) {
        UnitW &x = static_cast<UnitW &>(m);
        x.tgt3 = (r.x >> 24) & 15; x.tgt4 = (r.x >> 28) & 15;
        x.tgt_d3 = P.tgt_d[3 * U + u]; x.tgt_d4 = P.tgt_d[4 * U + u];
    }
}

template <bool X = false>
__device__ __forceinline__ void unit_store(const DevPtrs &P, size_t U, size_t u, const Unit &m) {
    P.lat[u] = m.lat; P.lon[u] = m.lon; P.hdg[u] = m.hdg; P.spd[u] = m.spd;
    P.cmd_hdg[u] = m.cmd_hdg; P.cmd_spd[u] = m.cmd_spd;
    int4 p;
    p.x = (m.cannon_remain & 0xffff) | ((m.cannon_max & 0xffff) << 16);
    p.y = (m.burst & 0xff) | ((m.missile_remain & 0xff) << 8) | ((m.rocket_max & 0xff) << 16) | ((m.missile_wait & 0xff) << 24);
    p.z = (m.ac_type & 0xff) | ((m.alive & 0xff) << 8) | ((m.has_missile & 0xff) << 16) | ((m.n_tgt & 0xff) << 24);
    p.w = (m.tgt0 & 0xff) | ((m.tgt1 & 0xff) << 8) | ((m.tgt2 & 0xff) << 16) | ((m.cmd_act & 0xff) << 24);
    P.pack[u] = p;
    P.tgt_d[u] = m.tgt_d0; P.tgt_d[U + u] = m.tgt_d1; P.tgt_d[2 * U + u] = m.tgt_d2;
    int2 r;
    r.x = (m.rk_alive & 0xff) | ((m.rk_target & 0xff) << 8) | ((m.rk_life & 0xff) << 16);
    if constexpr (X) {
        const UnitW &x = static_cast<const UnitW &>(m);
        r.x |= (int)(((unsigned)(x.tgt3 & 15) << 24) | ((unsigned)(x.tgt4 & 15) << 28));
        P.tgt_d[3 * U + u] = x.tgt_d3; P.tgt_d[4 * U + u] = x.tgt_d4;
    }
    r.y = m.rk_seq;
    P.rk_pack[u] = r;
    P.rk_lat[u] = m.rk_lat; P.rk_lon[u] = m.rk_lon; P.rk_hdg[u] = m.rk_hdg; P.rk_cmd[u] = m.rk_cmd;
}

__device__ __forceinline__ void arena_load(const DevPtrs &P, const DevCfg &c, int n, Arena &a) {
    int4 p = P.ar_pack[n];
    a.steps = p.x; a.episode = p.y;
    a.escaping = p.z & 0xff; a.escaping_time = (int)(int8_t)((p.z >> 8) & 0xff); a.done = (p.z >> 16) & 0xff;
    a.hl_s = (p.z >> 24) & 0x1f; a.hl_run = (p.z >> 29) & 1;
    a.next_seq = p.w;
    a.akey = hh_rng_arena_key(c.seed, c.arena_offset + (uint64_t)n);
    arena_rekey(a);
}

__device__ __forceinline__ void arena_store(const DevPtrs &P, int n, const Arena &a) {
    int4 p;
    p.x = a.steps; p.y = a.episode;
    p.z = (a.escaping & 0xff) | ((a.escaping_time & 0xff) << 8) | ((a.done & 0xff) << 16) | ((a.hl_s & 0x1f) << 24) | ((a.hl_run & 1) << 29);
    p.w = a.next_seq;
    P.ar_pack[n] = p;
}

/* ---- angle helpers (warsim/utils/angles.py:10-29) ---- */
__device__ __forceinline__ double d_normalize_angle(double a) {
    while (a >= 360.0) a -= 360.0;
    while (a < 0.0) a += 360.0;
    return a;
}
__device__ __forceinline__ double d_signed_heading_diff(double actual, double desired) {
    double delta = desired - actual;
    if (delta < -180.0) delta = 360.0 + delta;
    if (delta > 180.0) delta = -360.0 + delta;
    return delta;
}

/* cmano_simulator.py:167-174: range [km] and bearing [deg in [0,360)] from one Inverse solution */
__device__ __forceinline__ void d_dist_bearing(double lat1, double lon1, double lat2, double lon2, double &km, double &brg) {
    double s12, azi1;
    hh_geo_inverse(lat1, lon1, lat2, lon2, &s12, &azi1);
    km = s12 / 1000.0;
    brg = d_normalize_angle(azi1);
}

/* The exact envelope predicates (ac1.py:72-79,135-146, rocket_unit.py:39,49) on the Karney solution.
 * Out of line: reached only for the ~1e-5 of tests the estimate filter cannot decide.
 * kind 0 missile launch, 1 cannon cone, 2/3 rocket fuse; returns 1 inside the envelope. */
__device__ __forceinline__ int d_envelope_exact_body(int kind, int ac_type, double la1, double lo1, double la2, double lo2, double hdg) {
    double km, brg;
    d_dist_bearing(la1, lo1, la2, lo2, km, brg);
    if (kind == 0) {
        if (km <= HH_MISSILE_RANGE_KM) {
            double delta = hh_fabs(d_signed_heading_diff(d_normalize_angle(hdg + HH_MISSILE_HALF_DEG), brg));
            return (int)delta <= (int)HH_MISSILE_HALF_DEG;
        }
        return 0;
    }
    if (kind == 1) {
        if (km < HH_AC_CANNON_KM(ac_type)) return hh_fabs(d_signed_heading_diff(hdg, brg)) <= HH_AC_CANNON_HALF(ac_type);
        return 0;
    }
    return km < HH_ROCKET_FUSE_KM;
}
/* Out of line for the kernels that own a whole SIMD (the call keeps the hot loop's allocation and instruction cache
 * footprint small).  A callee saves registers into AGPRs, which counts against the caller: kernels held to two waves per SIMD
 * (256 registers in all) inline the body instead, where the rare path's pressure turns into spills in cold blocks only. */
__device__ __noinline__ int d_envelope_exact(int kind, int ac_type, double la1, double lo1, double la2, double lo2, double hdg) {
    return d_envelope_exact_body(kind, ac_type, la1, lo1, la2, lo2, hdg);
}

/* Exactness-preserving prefilter for "geodesic range < R km" tests.  Below 25 deg latitude one
 * degree of latitude or longitude is > 100 km on WGS84 (110.57 / >= 100.9 km), so a separation
 * of more than R/100 degrees along either axis proves range > R: the Inverse solve is skipped
 * and the mask bit is the same as the full computation would give. */
__device__ __forceinline__ bool d_maybe_within_km(double lat1, double lon1, double lat2, double lon2, double r_km) {
    double lim = r_km / 100.0;
    bool near = hh_fabs(lat2 - lat1) <= lim && hh_fabs(lon2 - lon1) <= lim;
    bool lowlat = hh_fabs(lat1) < 25.0 && hh_fabs(lat2) < 25.0;
    return near || !lowlat;
}

/* cmano_simulator.py:65-72 position update: short-step RK4 Direct (hh_geodesic.h) with the general
 * Karney Direct as an out-of-line fallback outside its domain (same selection as hh_geo_move) */
__device__ __noinline__ void d_geo_direct_general(double lat1, double lon1, double azi1, double s12, double *lat2, double *lon2) {
    hh_geo_direct(lat1, lon1, azi1, s12, lat2, lon2);
}
__device__ __forceinline__ void d_geo_move(double lat1, double lon1, double azi1, double s12, double &lat2, double &lon2) {
    double a, b;
    if (HH_USUAL(s12 <= HH_GEO_SHORT_MAX_M && hh_fabs(lat1) <= HH_GEO_SHORT_MAX_LAT && hh_fabs(lon1) < 170.0 && hh_fabs(azi1) <= 360.0))
        hh_geo_direct_short(lat1, lon1, azi1, s12, &a, &b);
    else
        d_geo_direct_general(lat1, lon1, azi1, s12, &a, &b);
    lat2 = a;
    lon2 = b;
}

/* two independent short-step moves in one basic block so that their RK4 chains interleave (ILP); each
 * result is bit-identical to d_geo_move of the same arguments */
__device__ __forceinline__ void d_geo_move2(double lat_a, double lon_a, double azi_a, double s_a, double &lat2_a, double &lon2_a,
                                            double lat_b, double lon_b, double azi_b, double s_b, double &lat2_b, double &lon2_b) {
    bool ok_a = s_a <= HH_GEO_SHORT_MAX_M && hh_fabs(lat_a) <= HH_GEO_SHORT_MAX_LAT && hh_fabs(lon_a) < 170.0 && hh_fabs(azi_a) <= 360.0;
    bool ok_b = s_b <= HH_GEO_SHORT_MAX_M && hh_fabs(lat_b) <= HH_GEO_SHORT_MAX_LAT && hh_fabs(lon_b) < 170.0 && hh_fabs(azi_b) <= 360.0;
    if (HH_USUAL(ok_a && ok_b)) {
        double a0, a1, b0, b1;
        hh_geo_direct_short(lat_a, lon_a, azi_a, s_a, &a0, &a1);
        hh_geo_direct_short(lat_b, lon_b, azi_b, s_b, &b0, &b1);
        lat2_a = a0; lon2_a = a1; lat2_b = b0; lon2_b = b1;
    } else {
        d_geo_move(lat_a, lon_a, azi_a, s_a, lat2_a, lon2_a);
        d_geo_move(lat_b, lon_b, azi_b, s_b, lat2_b, lon2_b);
    }
}

__device__ __forceinline__ double d_rng(const Arena &a, int unit_id, int site, int sub) {
    return hh_rng_u01(a.tkey, (uint32_t)unit_id, (uint32_t)site, (uint32_t)sub);
}

#endif /* HH_DEVICE_H */
