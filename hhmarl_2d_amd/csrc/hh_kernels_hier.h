/*
 * hh_kernels_hier.h — HighLevelEnv (3-vs-3 commander, envs/env_hier.py) on the same world.
 *
 * One commander step of the reference = _action_assess (env_hier.py:142-190) + up to 16 sub-steps of
 * { every live unit: frozen pilot policy on lowlevel_state -> _take_base_action } -> do_tick -> rewards
 * (env_hier.py:125-138), then state() (49-98).  The pilot networks run in PyTorch BETWEEN kernels, and
 * the reference lets units observe and act in id order, so the opponents' observations contain the
 * agents' same-sub-step weapon flags.  The macro step is therefore split into phases, one launch each:
 *
 *   HL_BEGIN       action assessment, opponents' fight/escape draws     -> agents' pilot observations
 *   HL_AGENTS_ACT  agents' _take_base_action (incl. missile envelope)   -> opponents' pilot observations
 *   HL_TICK        opponents' _take_base_action, do_tick, rewards, kill / surrounding events, s += 1
 *                                                                       -> agents' observations of the next sub-step
 *   HL_END         done, commander observation + stored target lists, episode statistics, auto-reset
 *   HL_REFRESH / HL_RESET   commander observation + lists only (hh_observe, hh_set_state) / masked reset first
 *
 * Arenas whose macro step ended early (kill or surrounding event) idle until HL_END; the host loops
 * a fixed 16 sub-steps or until the running counter reaches zero.
 */
#ifndef HH_KERNELS_HIER_H
#define HH_KERNELS_HIER_H

#include "hh_kernels.h"
#include <type_traits>

enum { HH_HL_BEGIN = 0, HH_HL_AGENTS_ACT = 1, HH_HL_TICK = 2, HH_HL_END = 3, HH_HL_REFRESH = 4, HH_HL_RESET = 5,
       /* the variant-row form (hh_kernels_oct.h: hh_k_hier_oct_v): one launch and one policy call per sub-step */
       HH_HL_ACT_TICK = 6, HH_HL_BEGIN_V = 7 };
/* HH_HL_VROWS (hh_abi.h) = 15 pilot row slots of an arena in the variant-row form: 3 agents + 3 opponents x 4 variants */

/* env_hier.py:100-112 lowlevel_state of the lane's unit -> 30 floats (zero padded) + policy type */
template <int A, int B>
__device__ __forceinline__ int hl_pilot_obs(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, const Unit &m, float *out) {
    for (int k = 0; k < 30; k++) out[k] = 0.0f;
    Near3 fr;
    nearby(c, sh, tid, base, s, true, fr);
    int n = 0;
    out[n++] = sh.nlat[tid];
    out[n++] = sh.nlon[tid];
    out[n++] = sh.nspd[tid];
    out[n++] = sh.nhdg[tid];
    int mode;
    /* the stored list by value (a select between member addresses would pin the Unit in scratch memory) */
    const int cmd_act = m.cmd_act, n_tgt = m.n_tgt, t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2;
    const double d0 = m.tgt_d0, d1 = m.tgt_d1, d2 = m.tgt_d2;
    if (cmd_act != 0) { /* fight the commander-chosen target, with the stale stored distance */
        mode = 1;
        double dist = d0;
        int oj = t0 - 1;
        if (cmd_act == 2) { dist = d1; oj = t1 - 1; }
        if (cmd_act >= 3) { dist = d2; oj = t2 - 1; }
        if constexpr (A > 8) { /* an opponent's list of up to five agents */
            const UnitW &x = static_cast<const UnitW &>(m);
            const int t3 = x.tgt3, t4 = x.tgt4;
            const double d3 = x.tgt_d3, d4 = x.tgt_d4;
            if (cmd_act == 4) { dist = d3; oj = t3 - 1; }
            if (cmd_act >= 5) { dist = d4; oj = t4 - 1; }
        }
        out[n++] = (float)norm180(sh.p_foc[oj][tid]);
        out[n++] = (float)aspect(sh.p_foc[s][base + oj]);
        out[n++] = sh.p_hd[oj][tid];
        out[n++] = (float)dist;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) {
            out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
            out[n++] = m.missile_wait == 0 ? 1.0f : 0.0f;
            out[n++] = (m.has_missile || m.burst > 0) ? 1.0f : 0.0f;
        } else {
            out[n++] = m.burst > 0 ? 1.0f : 0.0f;
        }
        n += opp_block(c, sh, 0, tid, base, s, oj, dist, out + n);
    } else {
        mode = 2;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
        out[n++] = (sh.flags[tid] & FL_SHOT) ? 1.0f : 0.0f;
        if (n_tgt >= 1) opp_block(c, sh, 1, tid, base, s, t0 - 1, d0, out + n);
        if (n_tgt >= 2) opp_block(c, sh, 1, tid, base, s, t1 - 1, d1, out + n + 9);
        n += 18;
    }
    if (fr.n) friend_block(c, sh, tid, base, s, fr.i0, out + n);
    return mode;
}

/* env_base.py:400-422 _nearby_object, every live unit of the other side (up to five) — the list an opponent of a ten-slot arena keeps
 * (env_hier.py:97); same distances, same stable order as nearby() */
struct Near5 {
    int n, i[5];
    double d[5];
};
template <int A, int B>
__device__ __forceinline__ void nearby5(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, Near5 &o) {
    o.n = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { o.i[k] = 0; o.d[k] = 0.0; }
    const bool me_agent = s < c.nA;
    const int lo = me_agent ? c.nA : 0, hi = me_agent ? A : c.nA;
#pragma unroll
    for (int j = 0; j < A; j++) {
        if (j < lo || j >= hi || j == s || !(sh.flags[base + j] & FL_ALIVE)) continue;
        const double dn = c.inv_diag * sh.p_dist[j][tid];
        int p = 0; /* stable: behind every entry that is not farther */
#pragma unroll
        for (int k = 0; k < 5; k++) p += (k < o.n && o.d[k] <= dn) ? 1 : 0;
        if (p >= 5) continue;
#pragma unroll
        for (int k = 4; k >= 1; k--) if (k > p) { o.i[k] = o.i[k - 1]; o.d[k] = o.d[k - 1]; }
#pragma unroll
        for (int k = 0; k < 5; k++) if (k == p) { o.i[k] = j; o.d[k] = dn; }
        if (o.n < 5) o.n++;
    }
}

/* env_hier.py:49-98 state(): commander observation (agents) and the stored sorted target lists (all units) */
template <int A, int B>
__device__ __forceinline__ void hl_commander_obs(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, Unit &m, float *out) {
    const bool agent = s < c.nA;
    if (agent) for (int k = 0; k < HH_OBS_HL; k++) out[k] = 0.0f;
    m.n_tgt = 0; m.tgt0 = m.tgt1 = m.tgt2 = 0; m.tgt_d0 = m.tgt_d1 = m.tgt_d2 = 0.0;
    if constexpr (A > 8) { UnitW &x = static_cast<UnitW &>(m); x.tgt3 = x.tgt4 = 0; x.tgt_d3 = x.tgt_d4 = 0.0; }
    if (!m.alive) return;
    if constexpr (A > 8) {
        if (!agent) { /* up to five agents (env_hier.py:97) */
            Near5 n5;
            nearby5(c, sh, tid, base, s, n5);
            m.n_tgt = n5.n;
            if (n5.n >= 1) { m.tgt0 = n5.i[0] + 1; m.tgt_d0 = n5.d[0]; }
            if (n5.n >= 2) { m.tgt1 = n5.i[1] + 1; m.tgt_d1 = n5.d[1]; }
            if (n5.n >= 3) { m.tgt2 = n5.i[2] + 1; m.tgt_d2 = n5.d[2]; }
            UnitW &x = static_cast<UnitW &>(m);
            if (n5.n >= 4) { x.tgt3 = n5.i[3] + 1; x.tgt_d3 = n5.d[3]; }
            if (n5.n >= 5) { x.tgt4 = n5.i[4] + 1; x.tgt_d4 = n5.d[4]; }
            return;
        }
    }
    Near3 nb;
    nearby(c, sh, tid, base, s, false, nb);
    if (agent) {
        if (nb.n == 0) return;
        int n = 0;
        out[n++] = sh.nlat[tid];
        out[n++] = sh.nlon[tid];
        out[n++] = sh.nspd[tid];
        out[n++] = sh.nhdg[tid];
        opp_block(c, sh, 2, tid, base, s, nb.i0, nb.d0, out + n);
        m.n_tgt = 1; m.tgt0 = nb.i0 + 1; m.tgt_d0 = nb.d0;
        if (nb.n >= 2) {
            opp_block(c, sh, 2, tid, base, s, nb.i1, nb.d1, out + n + 10);
            m.n_tgt = 2; m.tgt1 = nb.i1 + 1; m.tgt_d1 = nb.d1;
        }
        n += HH_N_OPP_HL * 10;
        Near3 fr;
        nearby(c, sh, tid, base, s, true, fr);
        if (fr.n >= 1) friend_block(c, sh, tid, base, s, fr.i0, out + n);
        if (fr.n >= 2) friend_block(c, sh, tid, base, s, fr.i1, out + n + 5);
    } else { /* opponents keep the full sorted list of agents (env_hier.py:97) */
        m.n_tgt = nb.n;
        if (nb.n >= 1) { m.tgt0 = nb.i0 + 1; m.tgt_d0 = nb.d0; }
        if (nb.n >= 2) { m.tgt1 = nb.i1 + 1; m.tgt_d1 = nb.d1; }
        if (nb.n >= 3) { m.tgt2 = nb.i2 + 1; m.tgt_d2 = nb.d2; }
    }
}

/* pilot_mode byte of a unit that flies a policy: policy type (1 fight / 2 escape) | aircraft type << 2, and with
 * hh_config.opp_side_selector the side bit on an opponent's fight row (env_base.py:387-390: "fight_{type}_opp") */
__device__ __forceinline__ int hl_selector(const DevCfg &c, int mode, int ac_type, bool agent) {
    return mode | (ac_type << 2) | ((c.sel_side && !agent && mode == 1) ? HH_SEL_OPP_SIDE : 0);
}

__device__ __forceinline__ int hl_gcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* env_hier.py:142-190 _action_assess for the lane's unit; returns the shaping reward of an agent */
template <int A, int B>
__device__ __forceinline__ double hl_action_assess(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, Unit &m,
                                                   const Arena &ar, int cmd) {
    const int id = s + 1;
    double rew = 0.0;
    if (!m.alive) { m.cmd_act = 0; return 0.0; }
    if (s < c.nA) {
        int cc = cmd;
        if (cc > 0) {
            const int t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2; /* by value: see hl_target_slot */
            int opp = 0;
            if (cc - 1 < m.n_tgt) { opp = t2; if (cc == 2) opp = t1; if (cc == 1) opp = t0; }
            else cc = 1;
            if (!opp) rew = -0.1;
            if (c.hier_action_assess && opp) {
                int oj = opp - 1;
                rew = (sh.p_dist[oj][tid] < 0.1 && sh.p_foc[oj][tid] < 15.0 && sh.p_foc[s][base + oj] > 40.0) ? 0.1 : 0.0;
            }
        } else if (c.hier_action_assess) {
            int oj = m.tgt0 - 1;
            if (sh.p_dist[s][base + oj] < 0.1 && sh.p_foc[s][base + oj] < 15.0 && sh.p_foc[oj][tid] > 40.0) rew = 0.1;
        }
        m.cmd_act = cc;
    } else {
        int g = hl_gcd(c.hier_opp_fight_ratio, 100);
        int num = c.hier_opp_fight_ratio / (g ? g : 1), den = 100 / (g ? g : 1);
        double total = (double)den + 0.0;
        int fight = d_rng(ar, id, HH_SITE_HL_FIGHT, 0) * total >= (double)(den - num);
        int ag = 0;
        if (fight) {
            int possible = m.n_tgt;
            if (possible > 1 && (d_rng(ar, id, HH_SITE_HL_OTHER, 0) * 4.0 >= 1.0))
                ag = hh_rng_randint(d_rng(ar, id, HH_SITE_HL_PICK, 0), 2, possible);
            else
                ag = 1;
        }
        m.cmd_act = ag;
    }
    return rew;
}

/* ---- the phases of a commander step as device functions: the phase-by-phase kernel (pilot networks between launches) and
 *      the persistent macro-step kernel (actions from a tape) run the SAME code, so their results are bit-identical ---- */

/* everything a lane carries through a macro step (X: a lane of a ten-slot arena, hh_device.h UnitW) */
template <bool X>
struct HlLaneT {
    typename std::conditional<X, UnitW, Unit>::type m;
    Arena ar;
    double acc;     /* reward accumulated over the macro step (agents) */
    double ep_ret;  /* episode return (lane s == 0) */
    uint32_t evm;   /* event bits of the current sub-step */
    int tcur;       /* trace cursor of the lane's arena (hh_trace_enable) */
};
using HlLane = HlLaneT<false>;

/* consumed: the row is one the reference hands to _take_base_action in this phase (hh_device.h: hh_act_unpack) */
__device__ __forceinline__ void hl_load_act(const int8_t *__restrict__ actions, size_t row, bool active, int8_t (&act)[4], int &fault, bool consumed) {
    act[0] = act[1] = act[2] = act[3] = 0;
    if (active) {
        const int w = *reinterpret_cast<const int *>(actions + row * 4);
        hh_act_unpack(w, act, fault, consumed);
    }
}

/* HL_BEGIN: env_hier.py:142-190 _action_assess + the opponents' draws */
template <int A, int B>
__device__ __forceinline__ void hl_do_begin(const DevCfg &c, Shared<A, B> &sh, int tid, int base, int s, int n, bool active, HlLaneT<(A > 8)> &L,
                                            const int8_t *__restrict__ cmd) {
    const bool agent = s < c.nA;
    L.ar.hl_s = 0;
    L.ar.hl_run = active && !L.ar.done;
    L.acc = 0.0;
    if (L.ar.hl_run) {
        int cc = agent ? (int)cmd[(size_t)n * c.nA + s] : 0;
        double r = hl_action_assess(c, sh, tid, base, s, L.m, L.ar, cc);
        if (agent) L.acc = r;
    }
}

/* HL_AGENTS_ACT: the agents' _take_base_action (incl. the missile envelope) */
template <int A, int B, int W>
__device__ __forceinline__ void hl_do_agents_act(const DevCfg &c, Shared<A, B> &sh, int tid, int base, int s, bool active, HlLaneT<(A > 8)> &L,
                                                 const int8_t (&act)[4]) {
    double pr = 0.0, os0 = 0.0;
    int vl = 0;
    act_phase<A, B, (W >= 2)>(c, sh, tid, s, base, active, L.ar.hl_run != 0, L.m, L.ar, act, s < c.nA, true, pr, os0, vl, L.evm);
}

/* HL_TICK: the opponents' _take_base_action, do_tick, rewards, kill / surrounding events, s += 1 (env_hier.py:125-138).
 * TAB: a pair table of the pre-tick positions is in LDS (persistent kernel: the previous tick left it); the phase kernel
 * has none at this point and computes the one entry a launch test needs on demand — the same expression, the same bits.
 * Returns 1 iff this lane's arena ran the tick. */
template <int A, int B, int W, bool TAB>
__device__ __forceinline__ int hl_do_tick(const DevPtrs &P, const DevCfg &c, Shared<A, B> &sh, int tid, int g, int base, int s, int n, bool active,
                                          HlLaneT<(A > 8)> &L, const int8_t (&act)[4]) {
    const bool agent = s < c.nA;
    { double pr = 0.0, os0 = 0.0; int vl = 0; act_phase<A, B, (W >= 2), TAB>(c, sh, tid, s, base, active, L.ar.hl_run != 0, L.m, L.ar, act, !agent, true, pr, os0, vl, L.evm); }
    StepOut so;
    so.reward = 0.0; so.valid = 0; so.opp_stat0 = 0.0;
    const bool was_running = active && L.ar.hl_run;
    uint32_t evm_tick = 0;
#ifdef HH_PROFILE_PHASES
    unsigned long long prof_t0_ = 0, prof_acc_[12] = {0};
#endif
    tick<A, B, (W >= 2)>(c, sh, tid, g, s, base, active, L.m, L.ar, act, so, evm_tick, 1, L.ar.hl_run != 0 HH_PROF_PASS);
    L.evm |= evm_tick;
    if (was_running) {
        if (agent) L.acc += so.reward;
        /* env_hier.py:133-138: surrounding event only after min_sub_steps, then s += 1, steps += 1 */
        int situ = 0;
        if (L.ar.hl_s > 10) {
#pragma unroll
            for (int i = 0; i < A; i++) {
                if (i >= c.nA || !sh_alive(sh, base + i)) continue;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    if (j < c.nA || !sh_alive(sh, base + j)) continue;
                    if (sh.p_dist[j][base + i] < 0.1 && (sh.p_foc[j][base + i] < 15.0 || sh.p_foc[i][base + j] < 15.0)) situ = 1;
                }
            }
        }
        L.ar.hl_s += 1;
        L.ar.steps += 1;
        arena_rekey(L.ar);
        L.ar.hl_run = (L.ar.hl_s <= 15 && !so.kill_event && !situ) ? 1 : 0;
        trace_append(P, A, n, s, L.m, L.ar, L.tcur);
    }
    return was_running ? 1 : 0;
}

/* HL_END / HL_REFRESH / HL_RESET: done, rewards out, episode statistics, eval counters, (auto-)reset, commander observation +
 * stored target lists (env_hier.py:49-98); the agents' rows are left staged in sh.u.obs for the caller to store */
template <int A, int B>
__device__ __forceinline__ void hl_do_end(const DevPtrs &P, const DevCfg &c, Shared<A, B> &sh, int tid, int g, int base, int s, int n, bool active,
                                          HlLaneT<(A > 8)> &L, int phase, float *__restrict__ reward_out, uint8_t *__restrict__ valid_out,
                                          uint8_t *__restrict__ done_out, const uint8_t *__restrict__ mask) {
    const bool agent = s < c.nA;
    Unit &m = L.m;
    Arena &ar = L.ar;
    const bool ending = phase == HH_HL_END && active && !ar.done; /* arena took part in this macro step */
    int ag = 0, op = 0;
#pragma unroll
    for (int j = 0; j < A; j++) { int al = sh_alive(sh, base + j); if (j < c.nA) ag += al; else op += al; }
    if (ending) ar.done = (ag <= 0 || op <= 0 || ar.steps >= c.horizon) ? 1 : 0;
    sh.rew[tid] = (ending && agent) ? L.acc : 0.0;
    hh_wg_sync<B>();
    if (ending && s == 0) {
        for (int j = 0; j < c.nA; j++) L.ep_ret += sh.rew[base + j];
        if (ar.done) {
            P.last_ret[n] = (float)L.ep_ret;
            P.last_len[n] = ar.steps;
            P.last_outcome[n] = (op <= 0 && ar.steps < c.horizon) ? 1 : ((ag <= 0 && ar.steps < c.horizon) ? -1 : 0);
        }
    }
    if (phase == HH_HL_END) {
        /* eval_info of this commander step (env_base.py:91-107): units that still exist, by assessed commander action */
        sh.res[tid] = (ending && m.alive) ? (1 | ((m.cmd_act & 7) << 1)) : 0;
        hh_wg_sync<B>();
        if (active && s == 0) {
            int e[HH_EVAL_K];
#pragma unroll
            for (int k = 0; k < HH_EVAL_K; k++) e[k] = 0;
            if (ending) {
                e[0] = (op <= 0 && ar.steps < c.horizon) ? 1 : 0;
                e[1] = (ag <= 0 && ar.steps < c.horizon) ? 1 : 0;
                e[2] = (ar.steps >= c.horizon && ag > 0 && op > 0) ? 1 : 0;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    const int w_ = sh.res[base + j];
                    if (!(w_ & 1)) continue;
                    const int v = w_ >> 1;
                    if (j < c.nA) { e[7] += 1; if (v) { e[3] += 1; e[8 + v] += 1; } else e[4] += 1; }
                    else { e[8] += 1; if (v) e[5] += 1; else e[6] += 1; }
                }
            }
#pragma unroll
            for (int k = 0; k < HH_EVAL_K; k++) {
                P.eval_last[(size_t)n * HH_EVAL_K + k] = e[k];
                if (e[k]) P.eval_tot[(size_t)n * HH_EVAL_K + k] += e[k];
            }
        }
        ar.hl_run = 0;
        if (active && agent) {
            size_t o = (size_t)n * c.nA + s;
            if (reward_out) reward_out[o] = ending ? (float)L.acc : 0.0f;
            if (valid_out) valid_out[o] = ending ? 1 : 0; /* every agent id has a reward key (env_hier.py:154,188) */
        }
        if (active && s == 0 && done_out) done_out[n] = (uint8_t)ar.done;
    }
    bool need_reset = phase == HH_HL_END ? (active && ar.done && c.auto_reset)
                                         : (phase == HH_HL_RESET && active && (mask == nullptr || mask[n]));
    const int any_reset = hh_wg_sync_or<B>(need_reset ? 1 : 0);
    if (need_reset) {
        reset_arena_scalars(ar);
        reset_unit<A>(c, s, m, ar);
        ar.hl_s = 0; ar.hl_run = 0;
        L.ep_ret = 0.0;
        L.acc = 0.0;
        trace_append(P, A, n, s, m, ar, L.tcur); /* first row of the new episode */
    }
    if (any_reset) {
        publish_obs(c, sh, tid, m);
        hh_wg_sync<B>();
        pair_tables(sh, tid, base, s, active);
        hh_wg_sync<B>();
    }
    hh_wg_sync<B>();
    if (active) { /* rows of the agents staged in LDS (opponents only refresh their stored target lists: nothing is written) */
        float *row = agent ? &sh.u.obs[(g * c.nA + s) * HH_OBS_HL] : &sh.u.obs[0];
        hl_commander_obs(c, sh, tid, base, s, m, row);
    }
    hh_wg_sync<B>();
}

template <int A, int B, int GPB = B / A>
__device__ __forceinline__ void hl_store_commander_obs(const DevCfg &c, const Shared<A, B> &sh, int tid, int phase, float *__restrict__ obs_out,
                                                       const uint8_t *__restrict__ mask) {
    if (!obs_out) return;
    const int arenas = min(GPB, c.N - (int)blockIdx.x * GPB);
    const int per = c.nA * HH_OBS_HL, cnt = arenas * per;
    float *dst = obs_out + (size_t)blockIdx.x * GPB * per;
    for (int k = tid; k < cnt; k += B) {
        const bool wr = phase != HH_HL_RESET || mask == nullptr || mask[blockIdx.x * GPB + k / per];
        if (wr) dst[k] = sh.u.obs[k];
    }
}

/* W = resident waves per SIMD the register allocation is held to (1: no spills, for up to one wave per SIMD) */
template <int A, int B, int W>
__global__ __launch_bounds__(B, W) void hh_k_hier(DevPtrs P, DevCfg c, int phase, const int8_t *__restrict__ cmd,
                                               const int8_t *__restrict__ actions, float *__restrict__ pilot_obs,
                                               uint8_t *__restrict__ pilot_mode, float *__restrict__ obs_out,
                                               float *__restrict__ reward_out, uint8_t *__restrict__ valid_out,
                                               uint8_t *__restrict__ done_out, int *__restrict__ running_count,
                                               const uint8_t *__restrict__ mask) {
    constexpr int GPB = B / A;
    __shared__ Shared<A, B> sh;
    __shared__ alignas(16) float ptile[W == 1 ? GPB * A * 30 : 4]; /* every unit's pilot row (W = 1 only, see below) */
#ifdef HH_PROFILE_PHASES /* tuning builds: where a phase launch spends its cycles (tools/phase_cost.py) */
    unsigned long long ht0_ = __builtin_readcyclecounter(), hacc_[8] = {0};
#define HH_HPROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); hacc_[k] += t_ - ht0_; ht0_ = t_; } while (0)
#else
#define HH_HPROF(k)
#endif
    const int tid = threadIdx.x;
    const int g = tid / A, s = tid % A;
    const int base = g * A;
    const int n = blockIdx.x * GPB + g;
    const bool active = g < GPB && n < c.N;
    const size_t U = (size_t)c.N * A;
    const size_t u = (size_t)n * A + s;
    const bool agent = s < c.nA;
    HlLaneT<(A > 8)> L;
    L.m = {};
    L.ar = Arena{};
    L.acc = 0.0; L.ep_ret = 0.0; L.evm = 0;
    L.tcur = (P.trace != nullptr && active && n < P.trace_K) ? P.trace_pos[n] : 0;
    Unit &m = L.m;
    Arena &ar = L.ar;
    if (active) {
        unit_load<(A > 8)>(P, U, u, m);
        arena_load(P, c, n, ar);
        L.acc = P.acc_rew[u];
        if (s == 0) L.ep_ret = P.ep_ret[n];
    } else {
        ar.done = 1;
    }
    sh.aux[tid] = 0;
    HH_HPROF(0); /* loads requested */
    publish_obs(c, sh, tid, m);
    hh_wg_sync<B>();
    HH_HPROF(1); /* state arrived, published */
    if (phase != HH_HL_TICK) { /* HL_TICK builds its table after the tick; before it only a launch test may ask for an entry */
        pair_tables(sh, tid, base, s, active);
        hh_wg_sync<B>();
    }
    HH_HPROF(2); /* pair table */
    int obs_side = -1; /* which side's pilot observations this launch emits */
    int act_fault = 0;  /* a consumed action word was out of range and ran sanitised (hh_act_unpack) */
    /* a bound policy bank (hh_bind_policy): this launch's pilot rows are binned by network here.  The selector of a row is known
     * as soon as the phase body is through (policy type from the commander's / the opponent's own choice, env_hier.py:100-112) — for
     * HL_AGENTS_ACT, which kills nobody, before it — so the list slot is REQUESTED there and used after the observation tile has
     * left: all workgroups finish together, and the same-address atomics of 820 waves arriving at once are a ~5 us tail otherwise. */
    int pslot = 0;
    HhBinTicket bt{0, 0};
    auto bin_issue = [&](int side) {
        const bool mine_ = side == 0 ? agent : !agent;
        const int sb = (active && ar.hl_run && m.alive && mine_) ? hl_selector(c, m.cmd_act != 0 ? 1 : 2, m.ac_type, agent) : 0;
        pslot = sb ? (int)P.pol_lut[sb] : 0;
        bt = hh_bin_rows_issue(P.pol_counts, pslot);
    };

    if (phase == HH_HL_BEGIN) {
        hl_do_begin<A, B>(c, sh, tid, base, s, n, active, L, cmd);
        obs_side = 0;
        if (P.pol_lut && pilot_obs) bin_issue(0);
    } else if (phase == HH_HL_AGENTS_ACT) {
        int8_t act[4];
        hl_load_act(actions, u, active, act, act_fault, ar.hl_run && m.alive && agent);
        if (P.pol_lut && pilot_obs) bin_issue(1);
        hl_do_agents_act<A, B, W>(c, sh, tid, base, s, active, L, act);
        obs_side = 1;
    } else if (phase == HH_HL_TICK) {
        int8_t act[4];
        hl_load_act(actions, u, active, act, act_fault, ar.hl_run && m.alive && !agent);
        const int ran = hl_do_tick<A, B, W, false>(P, c, sh, tid, g, base, s, n, active, L, act);
        if (s == 0 && ran && ar.hl_run && running_count) atomicAdd(running_count, 1);
        {   /* cumulative arena-ticks of this world (hh_hl_tick_count): one atomic per wave */
            const unsigned long long rn = __ballot(ran && s == 0);
            if (rn && (tid & 63) == 0 && running_count) atomicAdd(reinterpret_cast<unsigned long long *>(running_count + 2), (unsigned long long)__popcll(rn));
        }
        obs_side = 0;
        if (P.pol_lut && pilot_obs) bin_issue(0);
    } else { /* HH_HL_END, HH_HL_REFRESH, HH_HL_RESET */
        /* a bound policy bank: rows the last tick binned and nobody consumed (the macro step is over) are dropped here */
        if (P.pol_lut && blockIdx.x == 0 && tid <= 8) P.pol_counts[tid * HH_BIN_STRIDE] = 0;
        hl_do_end<A, B>(P, c, sh, tid, g, base, s, n, active, L, phase, reward_out, valid_out, done_out, mask);
        hl_store_commander_obs<A, B>(c, sh, tid, phase, obs_out, mask);
    }
    HH_HPROF(3); /* phase body */
    /* pilot observations: every unit's row is staged in LDS (the tick's exchange area is free by now) and the workgroup's
     * rows, contiguous in [N, A, 30], leave with unit-stride 16-byte stores */
    if (obs_side >= 0 && pilot_obs) {
        hh_wg_sync<B>(); /* all reads of the tick's LDS area are done */
        constexpr int HALF = A / 2; /* units per side */
        const bool mine = obs_side == 0 ? agent : !agent;
        const int arenas = min(GPB, c.N - (int)blockIdx.x * GPB);
        float *dst = pilot_obs + (size_t)blockIdx.x * GPB * A * 30; /* the workgroup's rows are contiguous in [N, A, 30] */
        if constexpr (W == 1) {
            /* a workgroup per SIMD: LDS is plentiful, every unit stages its row (zeros for the side that does not act) and
             * the tile leaves with 16-byte stores */
            if (active) {
                float *row = &ptile[tid * 30];
                int mode = 0;
                if (ar.hl_run && m.alive && mine) mode = hl_pilot_obs(c, sh, tid, base, s, m, row);
                else for (int k = 0; k < 30; k++) row[k] = 0.0f;
                if (pilot_mode) pilot_mode[u] = (uint8_t)(mode ? hl_selector(c, mode, m.ac_type, agent) : 0); /* policy type | aircraft type: selects the network */
            }
            hh_wg_sync<B>();
            const int cnt = arenas * A * 30;
            if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (cnt & 3) == 0) {
                const float4 *src4 = reinterpret_cast<const float4 *>(ptile);
                float4 *dst4 = reinterpret_cast<float4 *>(dst);
                for (int k = tid; k < (cnt >> 2); k += B) dst4[k] = src4[k];
            } else {
                for (int k = tid; k < cnt; k += B) dst[k] = ptile[k];
            }
            if (P.pol_lut) hh_bin_rows_finish(bt, P.pol_lists, P.pol_max_rows, (int)u, pslot);
        } else {
            /* two workgroups per SIMD share the CU's 160 KB: only the acting side's rows are staged (in the tick's exchange
             * area), the other side's zeros are produced by the store loop; 8-byte stores (a side's 90 floats are even) */
            if (active) {
                int mode = 0;
                if (mine) {
                    float *row = &sh.u.obs[(g * HALF + (s % HALF)) * 30];
                    if (ar.hl_run && m.alive) mode = hl_pilot_obs(c, sh, tid, base, s, m, row);
                    else for (int k = 0; k < 30; k++) row[k] = 0.0f;
                }
                if (pilot_mode) pilot_mode[u] = (uint8_t)(mode ? hl_selector(c, mode, m.ac_type, agent) : 0); /* policy type | aircraft type: selects the network */
            }
            hh_wg_sync<B>();
            const int cnt2 = arenas * A * 15; /* float2 elements */
            const float2 *src2 = reinterpret_cast<const float2 *>(sh.u.obs);
            if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
                float2 *dst2 = reinterpret_cast<float2 *>(dst);
                static_assert(B < A * 15, "index update below assumes one wrap per step");
                int ar_ = 0, r = tid; /* arena, float2 index inside its A rows; advanced without divisions */
                for (int k = tid; k < cnt2; k += B) {
                    const int side = r >= HALF * 15, q = r - side * (HALF * 15);
                    dst2[k] = side == obs_side ? src2[ar_ * (HALF * 15) + q] : make_float2(0.0f, 0.0f);
                    r += B;
                    if (r >= A * 15) { r -= A * 15; ar_ += 1; }
                }
            } else {
                for (int k = tid; k < cnt2 * 2; k += B) {
                    const int ar_ = k / (A * 30), r = k - ar_ * (A * 30);
                    const int side = r / (HALF * 30), q = r - side * (HALF * 30);
                    dst[k] = side == obs_side ? sh.u.obs[ar_ * (HALF * 30) + q] : 0.0f;
                }
            }
            if (P.pol_lut) hh_bin_rows_finish(bt, P.pol_lists, P.pol_max_rows, (int)u, pslot);
        }
    }
    HH_HPROF(4); /* pilot rows staged and stored */
    if (active) {
        unit_store<(A > 8)>(P, U, u, m);
        P.acc_rew[u] = L.acc;
        if (s == 0) {
            if (P.trace != nullptr && n < P.trace_K) P.trace_pos[n] = L.tcur;
            arena_store(P, n, ar);
            P.ep_ret[n] = L.ep_ret;
        }
    }
    HH_HPROF(5); /* state stores issued */
#ifdef HH_PROFILE_PHASES
    if ((tid & 63) == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&hh_prof_cycles[k_], hacc_[k_]); atomicAdd(&hh_prof_cycles[6 + (phase == HH_HL_TICK)], 1ULL); }
#endif
    if (phase == HH_HL_AGENTS_ACT || phase == HH_HL_TICK) {
        if (active && s == 0 && phase == HH_HL_AGENTS_ACT && ar.hl_run) P.ev_mask[n] = 0;
        __syncthreads();
        if (active && L.evm) atomicOr(&P.ev_mask[n], L.evm);
        hh_act_fault_commit(P, n, active, act_fault);
    }
}

/* ---- the persistent macro step: ONE launch per commander step when the pilots' actions are resident before it starts
 * (a tape: actions [16][N][A][4], sub-step major — the layout TapePilot hands out slice by slice).  State stays in registers
 * from _action_assess to the commander observation; the pair table a tick leaves behind serves the next sub-step's act
 * phases; a workgroup leaves the sub-step loop as soon as none of its arenas is still inside its macro step
 * (13.2 of 16 sub-steps on average).  Same device functions as the phase kernel: bit-identical. ---- */
template <int A, int B, int W, bool HLD, int APW = B / A>
__global__ __launch_bounds__(B, W) void hh_k_hier_macro(DevPtrs P, DevCfg c_in, const int8_t *__restrict__ cmd, const int8_t *__restrict__ tape,
                                                     float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                     uint8_t *__restrict__ valid_out, uint8_t *__restrict__ done_out,
                                                     int *__restrict__ counters) {
    constexpr int GPB = APW; /* arenas per wave: 10 fill 60 lanes; 8 when that still leaves every workgroup a SIMD of its own (a wave pays for
                                every branch any of its arenas takes: hh_kernels_quad.h) */
    static_assert(APW <= B / A, "arenas per wave");
    DevCfg c_hl = c_in;
    hh_cfg_set_hl_default(c_hl); /* HLD: the default HighLevelEnv configuration as literals (hh_device.h) */
    const DevCfg &c = HLD ? c_hl : c_in;
    __shared__ Shared<A, B> sh;
    const int tid = threadIdx.x;
    const int g = tid / A, s = tid % A;
    const int base = g * A;
    const int n = blockIdx.x * GPB + g;
    const bool active = g < GPB && n < c.N;
    const size_t U = (size_t)c.N * A;
    const size_t u = (size_t)n * A + s;
    HlLaneT<(A > 8)> L;
    L.m = {};
    L.ar = Arena{};
    L.acc = 0.0; L.ep_ret = 0.0; L.evm = 0;
    L.tcur = (P.trace != nullptr && active && n < P.trace_K) ? P.trace_pos[n] : 0;
    if (active) {
        unit_load<(A > 8)>(P, U, u, L.m);
        arena_load(P, c, n, L.ar);
        if (s == 0) L.ep_ret = P.ep_ret[n];
    } else {
        L.ar.done = 1;
    }
    sh.aux[tid] = 0;
    publish_obs(c, sh, tid, L.m);
    hh_wg_sync<B>();
    pair_tables(sh, tid, base, s, active);
    hh_wg_sync<B>();
    hl_do_begin<A, B>(c, sh, tid, base, s, n, active, L, cmd);
    int ticks = 0;
    uint32_t evm_last = 0;
    int act_fault = 0;
    /* the action word of the next sub-step is requested a sub-step ahead (one wave per SIMD cannot hide the round trip) */
    int act_next = active ? *reinterpret_cast<const int *>(tape + u * 4) : 0;
    for (int sub = 0; sub < 16; sub++) {
        if (!hh_wg_sync_or<B>(L.ar.hl_run)) break; /* nobody in this workgroup is inside a macro step any more */
        const int w = act_next;
        if (active && sub + 1 < 16) act_next = *reinterpret_cast<const int *>(tape + ((size_t)(sub + 1) * U + u) * 4);
        int8_t act[4];
        const bool running = active && L.ar.hl_run;
        hh_act_unpack(w, act, act_fault, running && L.m.alive);
        if (running) L.evm = 0;
        hl_do_agents_act<A, B, W>(c, sh, tid, base, s, active, L, act);
        ticks += hl_do_tick<A, B, W, true>(P, c, sh, tid, g, base, s, n, active, L, act);
        if (running) evm_last = L.evm;
    }
    hl_do_end<A, B>(P, c, sh, tid, g, base, s, n, active, L, HH_HL_END, reward_out, valid_out, done_out, nullptr);
    hl_store_commander_obs<A, B, GPB>(c, sh, tid, HH_HL_END, obs_out, nullptr);
    if (active) {
        unit_store<(A > 8)>(P, U, u, L.m);
        P.acc_rew[u] = L.acc;
        if (s == 0) {
            if (P.trace != nullptr && n < P.trace_K) P.trace_pos[n] = L.tcur;
            arena_store(P, n, L.ar);
            P.ep_ret[n] = L.ep_ret;
        }
    }
    {   /* cumulative arena-ticks (hh_hl_tick_count): one atomic per wave */
        int t = s == 0 ? ticks : 0;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((tid & 63) == 0 && t && counters) atomicAdd(reinterpret_cast<unsigned long long *>(counters + 2), (unsigned long long)t);
    }
    /* event masks of each arena's last sub-step, like the phase path leaves them */
    if (active && s == 0 && ticks) P.ev_mask[n] = 0;
    __syncthreads();
    if (active && ticks && evm_last) atomicOr(&P.ev_mask[n], evm_last);
    hh_act_fault_commit(P, n, active, act_fault);
}

#endif /* HH_KERNELS_HIER_H */
