/*
 * hh_world.hip — gfx950 kernels and C-ABI (include/hh_abi.h) of the batched air-combat world.
 *
 * Replaces, for thousands of arenas at once, the reference's per-process Python path
 *   envs/env_base.py:79-109 step  ->  envs/env_hetero.py:105-186 _take_action
 *   -> warsim/simulator/cmano_simulator.py:138-157 do_tick (ac1.py:81-133, ac2.py:68-107,
 *      rocket_unit.py:37-73)  ->  env_hetero.py:188-225 rewards  ->  env_hetero.py:65-103 state
 * and envs/env_base.py:62-77,551-585 reset.
 *
 * Kernels (all: one lane per aircraft slot, see hh_device.h):
 *   hh_k_rollout<A,B>  persistent multi-tick kernel: state is loaded into registers once, then
 *                      for each of T ticks  K1 "step" (fused action decode + scripted opponents
 *                      + turn/thrust kinematics + WGS84 Direct/Inverse + cannon/missile
 *                      envelopes + id-ordered kill resolution + rewards/done)  and
 *                      K2 "observe" (per-agent observation packing staged through LDS and
 *                      written with coalesced stores) run back to back; K3 "reset" re-samples
 *                      finished arenas in place when auto_reset is set.  T = 1 is hh_step.
 *   hh_k_observe<A,B>  K3 + K2 only (hh_reset, hh_set_state).
 * No MFMA: there is no dense contraction on this path; it is FP64 VALU + HBM streaming.
 *
 * Arithmetic is the bit-reproducible include/hh_math.h / hh_geodesic.h set, compiled with
 * -ffp-contract=off, so results are bit-identical to the CPU oracle (tests/test_gpu_parity.py).
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "hh_device.h"

/* ===================================================================== LDS exchange area */
template <int A, int B>
struct Shared {
    static constexpr int GPB = B / A;
    double lat0[B], lon0[B]; /* published positions: pre-tick, then post-tick for observe */
    double lat1[B], lon1[B]; /* post-move positions inside the tick */
    double hdg[B], uc[B], us[B], spd[B];
    double rew[B];
    int flags[B];            /* bit0 alive, bits1-2 ac_type, bit3 shot flag, bit4 burst>0 */
    int aux[B];              /* per-phase scratch: launch flag, cannon candidate mask, rocket word, oob */
    int g_alive[GPB];        /* alive mask after resolution */
    int g_nev[GPB];
    int g_ev[GPB][HH_MAX_AIRCRAFT]; /* killer | victim<<4 | rocket<<8, in reference event order */
    int g_rkdead[GPB];       /* rocket slots removed this tick */
    float obs[GPB * 3 * HH_OBS_HL];
};

#define FL_ALIVE 1
#define FL_SHOT 8
#define FL_BURST 16

template <int A, int B>
__device__ __forceinline__ int sh_alive(const Shared<A, B> &sh, int idx) { return sh.flags[idx] & FL_ALIVE; }
template <int A, int B>
__device__ __forceinline__ int sh_type(const Shared<A, B> &sh, int idx) { return (sh.flags[idx] >> 1) & 3; }

/* env_base.py:424-432 _focus_angle [deg]: angle at unit a between its heading and the line to b */
template <int A, int B>
__device__ __forceinline__ double focus_deg(const Shared<A, B> &sh, int a, int b) {
    double c = sh.uc[a], s = sh.us[a];
    double dx = sh.lon0[b] - sh.lon0[a], dy = sh.lat0[b] - sh.lat0[a];
    double dot = c * dx + s * dy;
    double n1 = hh_sqrt(c * c + s * s), n2 = hh_sqrt(dx * dx + dy * dy);
    double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
    return hh_acos(x) * (180.0 / HH_PI);
}
template <int A, int B>
__device__ __forceinline__ double focus_norm(const Shared<A, B> &sh, int a, int b) {
    return hh_clip(focus_deg(sh, a, b) / 180.0, 0.0, 1.0);
}
/* env_base.py:441-446 */
template <int A, int B>
__device__ __forceinline__ double aspect_from_focus(double fdeg) { return hh_clip((180.0 - fdeg) / 180.0, 0.0, 1.0); }
/* env_base.py:448-456 */
template <int A, int B>
__device__ __forceinline__ double heading_diff_norm(const Shared<A, B> &sh, int a, int b) {
    double c1 = sh.uc[a], s1 = sh.us[a], c2 = sh.uc[b], s2 = sh.us[b];
    double dot = c1 * c2 + s1 * s2;
    double n1 = hh_sqrt(c1 * c1 + s1 * s1), n2 = hh_sqrt(c2 * c2 + s2 * s2);
    double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
    return hh_clip((hh_acos(x) * (180.0 / HH_PI)) / 180.0, 0.0, 1.0);
}
template <int A, int B>
__device__ __forceinline__ double dist_raw(const Shared<A, B> &sh, int a, int b) {
    return hh_hypot(sh.lon0[b] - sh.lon0[a], sh.lat0[b] - sh.lat0[a]);
}

/* env_base.py:400-422 _nearby_object over the published positions: up to 3 live units of the
 * other side (or own side), stable-sorted by normalised distance.  ids are slot indices. */
struct Near3 {
    int n, i0, i1, i2;
    double d0, d1, d2, r0, r1, r2;
};
template <int A, int B>
__device__ __forceinline__ void nearby(const DevCfg &c, const Shared<A, B> &sh, int base, int s, bool friendly, Near3 &o) {
    o.n = 0; o.i0 = o.i1 = o.i2 = 0; o.d0 = o.d1 = o.d2 = 0.0; o.r0 = o.r1 = o.r2 = 0.0;
    bool me_agent = s < c.nA;
    int lo = (me_agent != friendly) ? c.nA : 0;
    int hi = (me_agent != friendly) ? A : c.nA;
#pragma unroll
    for (int j = 0; j < A; j++) {
        if (j < lo || j >= hi || j == s) continue;
        if (!sh_alive(sh, base + j)) continue;
        double dr = dist_raw(sh, base + s, base + j);
        double dn = c.inv_diag * dr;
        int p = (o.n >= 1 && o.d0 <= dn) + (o.n >= 2 && o.d1 <= dn) + (o.n >= 3 && o.d2 <= dn);
        if (p <= 1) { o.i2 = o.i1; o.d2 = o.d1; o.r2 = o.r1; }
        if (p == 0) { o.i1 = o.i0; o.d1 = o.d0; o.r1 = o.r0; o.i0 = j; o.d0 = dn; o.r0 = dr; }
        else if (p == 1) { o.i1 = j; o.d1 = dn; o.r1 = dr; }
        else if (p == 2) { o.i2 = j; o.d2 = dn; o.r2 = dr; }
        if (o.n < 3) o.n++;
    }
}

/* publish the state other lanes read (heading unit vector as in env_base.py:428) */
template <int A, int B>
__device__ __forceinline__ void publish(Shared<A, B> &sh, int tid, const Unit &m) {
    sh.lat0[tid] = m.lat;
    sh.lon0[tid] = m.lon;
    sh.hdg[tid] = m.hdg;
    sh.spd[tid] = m.spd;
    double s, c;
    hh_sincos(hh_pymod(90.0 - m.hdg, 360.0) * (HH_PI / 180.0), &s, &c);
    sh.uc[tid] = c;
    sh.us[tid] = s;
    int shot = m.burst > 0 || (m.ac_type == 1 && m.has_missile);
    sh.flags[tid] = (m.alive ? FL_ALIVE : 0) | ((m.ac_type & 3) << 1) | (shot ? FL_SHOT : 0) | (m.burst > 0 ? FL_BURST : 0);
}

/* ===================================================================== K3: reset */
/* env_base.py:489-549 / env_hier.py:226-250 _sample_state + env_base.py:551-585 _reset_scenario */
template <int A>
__device__ __forceinline__ void reset_unit(const DevCfg &c, int s, Unit &m, Arena &ar) {
    bool agent = s < c.nA;
    int i = agent ? s : s - c.nA;
    int id = s + 1;
    int r = hh_rng_randint(d_rng(ar, 0, HH_SITE_RESET_SIDE, 0), 1, 2);
    double ux = d_rng(ar, id, HH_SITE_RESET_X, 0), uy = d_rng(ar, id, HH_SITE_RESET_Y, 0), uh = d_rng(ar, id, HH_SITE_RESET_HDG, 0);
    bool near_side = agent ? (r == 1) : (r == 2);
    double x, y;
    int hd = 0;
    if (c.env_kind == HH_ENV_HIGHLEVEL) {
        double n = agent ? (double)c.nA : (double)c.nO;
        x = near_side ? hh_rng_uniform(ux, 7.07, 7.22) : hh_rng_uniform(ux, 7.28, 7.43);
        y = hh_rng_uniform(uy, 5.07 + i * (0.4 / n), 5.12 + i * (0.4 / n));
        hd = hh_rng_randint(uh, 0, 359);
    } else if (c.level == 1) {
        x = near_side ? hh_rng_uniform(ux, 7.12, 7.14) : hh_rng_uniform(ux, 7.16, 7.17);
        y = hh_rng_uniform(uy, 5.1 + i * 0.1, 5.11 + i * 0.1);
        if (agent) hd = r == 1 ? hh_rng_randint(uh, 30, 150) : hh_rng_randint(uh, 200, 330);
    } else if (c.level == 2) {
        x = near_side ? hh_rng_uniform(ux, 7.08, 7.13) : hh_rng_uniform(ux, 7.18, 7.23);
        y = hh_rng_uniform(uy, 5.08 + i * 0.1, 5.13 + i * 0.1);
        if (agent) hd = r == 1 ? hh_rng_randint(uh, 0, 180) : hh_rng_randint(uh, 180, 359);
        else hd = hh_rng_randint(uh, 0, 359);
    } else {
        x = near_side ? hh_rng_uniform(ux, 7.07, 7.12) : hh_rng_uniform(ux, 7.18, 7.23);
        y = hh_rng_uniform(uy, 5.09 + i * 0.1, 5.12 + i * 0.1);
        if (agent) hd = r == 1 ? hh_rng_randint(uh, 0, 270) : hh_rng_randint(uh, 90, 359);
        else hd = hh_rng_randint(uh, 0, 359);
    }
    int ac = i <= 1 ? i + 1 : hh_rng_randint(d_rng(ar, id, HH_SITE_RESET_TYPE, 0), 1, 2);
    m = Unit{};
    m.lat = y; m.lon = x; m.hdg = (double)hd;
    m.spd = (c.level <= 2 && !agent) ? 0.0 : 100.0;
    m.cmd_hdg = m.hdg; m.cmd_spd = m.spd;
    m.alive = 1; m.ac_type = ac;
    m.cannon_remain = m.cannon_max = HH_AC_CANNON_DEFAULT;
    m.missile_remain = m.rocket_max = ac == 1 ? HH_AC1_MISSILES_DEFAULT : 0;
    if (c.env_kind == HH_ENV_LOWLEVEL) {
        if (c.level <= 4 && !agent) {
            m.cannon_remain = m.cannon_max = 400;
            if (ac == 1) m.missile_remain = m.rocket_max = 8;
        } else if (c.level == 5) {
            m.cannon_remain = m.cannon_max = 300;
            if (ac == 1) m.missile_remain = m.rocket_max = 6;
        }
    } else {
        m.cannon_remain = m.cannon_max = 300;
        if (ac == 1) m.missile_remain = m.rocket_max = 8;
    }
}

__device__ __forceinline__ void reset_arena_scalars(Arena &ar) {
    ar.episode += 1;
    ar.steps = 0;
    ar.escaping = 0;
    ar.escaping_time = 0;
    ar.next_seq = 0;
    ar.done = 0;
}

/* ===================================================================== K2: observe */
/* env_base.py:185-212 opp_ac_values, mode 0 fight / 1 esc / 2 HighLevel; f_so = focus(self->opp) deg,
 * f_os = focus(opp->self) deg (computed once, used for focus and aspect entries) */
template <int A, int B>
__device__ __forceinline__ int opp_block(const DevCfg &c, const Shared<A, B> &sh, int mode, int me, int o, double dist,
                                         double f_so, double f_os, float *out) {
    int n = 0;
    int t = sh_type(sh, o);
    out[n++] = (float)hh_clip((sh.lat0[o] - HH_MAP_LAT0) / c.ext_lat, 0.0, 1.0);
    out[n++] = (float)hh_clip((sh.lon0[o] - HH_MAP_LON0) / c.ext_lon, 0.0, 1.0);
    out[n++] = (float)hh_clip(sh.spd[o] / HH_AC_MAX_SPEED(t), 0.0, 1.0);
    out[n++] = (float)hh_clip(hh_pymod(sh.hdg[o], 359.0) / 359.0, 0.0, 1.0);
    out[n++] = (float)heading_diff_norm(sh, o, me);
    if (mode == 0) {
        out[n++] = (float)hh_clip(f_os / 180.0, 0.0, 1.0);
        out[n++] = (float)aspect_from_focus<A, B>(f_so);
    } else {
        out[n++] = (float)hh_clip(f_so / 180.0, 0.0, 1.0);
        out[n++] = (float)hh_clip(f_os / 180.0, 0.0, 1.0);
    }
    if (mode == 2) {
        out[n++] = (float)aspect_from_focus<A, B>(f_so);
        out[n++] = (float)aspect_from_focus<A, B>(f_os);
    }
    out[n++] = (float)dist;
    if (mode != 2) out[n++] = (sh.flags[o] & FL_SHOT) ? 1.0f : 0.0f;
    return n;
}

/* env_base.py:166-183 friendly_ac_values */
template <int A, int B>
__device__ __forceinline__ void friend_block(const DevCfg &c, const Shared<A, B> &sh, int me, int f, bool have, float *out) {
    if (have && sh_alive(sh, f)) {
        out[0] = (float)hh_clip((sh.lat0[f] - HH_MAP_LAT0) / c.ext_lat, 0.0, 1.0);
        out[1] = (float)hh_clip((sh.lon0[f] - HH_MAP_LON0) / c.ext_lon, 0.0, 1.0);
        out[2] = (float)focus_norm(sh, me, f);
        out[3] = (float)focus_norm(sh, f, me);
        out[4] = (float)(c.inv_diag * dist_raw(sh, me, f));
    } else {
        out[0] = out[1] = out[2] = out[3] = out[4] = 0.0f;
    }
}

/* env_hetero.py:65-103 lowlevel_state for the lane's own unit (fight / escape), also refreshes
 * opp_to_attack (m.tgt0).  Writes D floats (zero padded) to `out` (LDS staging row). */
template <int A, int B>
__device__ __forceinline__ void lowlevel_obs(const DevCfg &c, const Shared<A, B> &sh, int base, int s, int mode, Unit &m, float *out, int D) {
    for (int k = 0; k < D; k++) out[k] = 0.0f;
    m.n_tgt = 0; m.tgt0 = 0; m.tgt_d0 = 0.0;
    if (!m.alive) return;
    Near3 nb;
    nearby(c, sh, base, s, false, nb);
    if (nb.n == 0) return;
    int me = base + s;
    m.n_tgt = 1; m.tgt0 = nb.i0 + 1; m.tgt_d0 = nb.d0;
    /* env_hetero.py:71-75 fri_ac_id */
    int fri = s < c.nA ? (s == 1 ? 0 : 1) : (s == 3 ? 2 : 3);
    int n = 0;
    out[n++] = (float)hh_clip((m.lat - HH_MAP_LAT0) / c.ext_lat, 0.0, 1.0);
    out[n++] = (float)hh_clip((m.lon - HH_MAP_LON0) / c.ext_lon, 0.0, 1.0);
    out[n++] = (float)hh_clip(m.spd / HH_AC_MAX_SPEED(m.ac_type), 0.0, 1.0);
    out[n++] = (float)hh_clip(hh_pymod(m.hdg, 359.0) / 359.0, 0.0, 1.0);
    if (mode == HH_MODE_FIGHT) {
        int o = base + nb.i0;
        double f_so = focus_deg(sh, me, o), f_os = focus_deg(sh, o, me);
        out[n++] = (float)hh_clip(f_so / 180.0, 0.0, 1.0);
        out[n++] = (float)aspect_from_focus<A, B>(f_os);
        out[n++] = (float)heading_diff_norm(sh, me, o);
        out[n++] = (float)nb.d0;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) {
            out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
            out[n++] = m.missile_wait == 0 ? 1.0f : 0.0f;
            out[n++] = (m.has_missile || m.burst > 0) ? 1.0f : 0.0f;
        } else {
            out[n++] = m.burst > 0 ? 1.0f : 0.0f;
        }
        n += opp_block(c, sh, 0, me, o, nb.d0, f_so, f_os, out + n);
    } else {
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
        out[n++] = (sh.flags[me] & FL_SHOT) ? 1.0f : 0.0f;
        {
            int o = base + nb.i0;
            opp_block(c, sh, 1, me, o, nb.d0, focus_deg(sh, me, o), focus_deg(sh, o, me), out + n);
        }
        if (nb.n >= 2) {
            int o = base + nb.i1;
            opp_block(c, sh, 1, me, o, nb.d1, focus_deg(sh, me, o), focus_deg(sh, o, me), out + n + 9);
        }
        n += 18;
    }
    friend_block(c, sh, me, base + fri, true, out + n);
}

/* ===================================================================== K1: step */
struct StepOut {
    double reward;
    int valid;
};

template <int A, int B>
__device__ __forceinline__ void tick(const DevCfg &c, Shared<A, B> &sh, int tid, int g, int s, int base, bool active,
                                     Unit &m, Arena &ar, const int8_t *act, StepOut &out, uint32_t &ev_mask_out) {
    constexpr int GPB = B / A;
    const int id = s + 1;
    const bool running = active && !ar.done;
    const bool agent = s < c.nA;
    const bool hl = c.env_kind == HH_ENV_HIGHLEVEL;
    (void)GPB; (void)hl;
    out.reward = 0.0;
    out.valid = 0;
    uint32_t evm = 0;
    if (running) ar.steps += 1;
    const bool snap = running && m.alive; /* in do_tick's start-of-tick snapshot */
    double opp_stat0 = 0.0;
    int want_launch = 0, launch_tgt = 0; /* launch_tgt: slot index */
    int wait_after = -1;                 /* scripted opponents: missile_wait value set after the attempt */
    bool base_gate = false;              /* _take_base_action missile gate passed */

    /* ---------------- phase A1: commands (env_hetero.py:160-182) ---------------- */
    if (snap) {
        if (agent || c.ext_opp) {
            int t = m.n_tgt ? m.tgt0 : 0;
            if (!agent) {
                /* env_base.py:349-398 _policy_actions -> lowlevel_state(opp_mode, i): refresh target */
                Near3 nb;
                nearby(c, sh, base, s, false, nb);
                m.n_tgt = nb.n ? 1 : 0; m.tgt0 = nb.n ? nb.i0 + 1 : 0; m.tgt_d0 = nb.n ? nb.d0 : 0.0;
                t = m.tgt0;
            } else {
                out.valid = 1;
                if (t && sh_alive(sh, base + t - 1)) opp_stat0 = focus_norm(sh, base + t - 1, base + s);
            }
            /* env_base.py:214-238 _take_base_action */
            double nh = hh_pymod(m.hdg + (double)(((int)act[0] - 6) * 15), 360.0);
            if (nh >= 360.0 || nh < 0.0) nh = 0.0;
            m.cmd_hdg = nh;
            double mx = HH_AC_MAX_SPEED(m.ac_type);
            m.cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
            bool agent_ll = agent && !hl;
            if (act[2] && m.cannon_remain > 0) {
                int b = HH_AC_BURST(m.ac_type);
                m.burst = m.cannon_remain < b ? m.cannon_remain : b;
                if (agent_ll && c.agent_mode == HH_MODE_ESCAPE && m.cannon_remain < 90) out.reward -= 0.1;
            }
            if (m.ac_type == 1 && act[3]) {
                if (t && m.missile_remain > 0 && !m.has_missile && m.missile_wait == 0) {
                    base_gate = true;
                    want_launch = 1;
                    launch_tgt = t - 1;
                }
            }
        } else if (c.level == 1) {
            /* env_hetero.py:118-123 */
            if (!m.has_missile && (ar.steps % 40) < 3 && hh_rng_randint(d_rng(ar, id, HH_SITE_L12_COIN, 0), 0, 1) &&
                m.missile_wait == 0 && m.ac_type == 1) {
                Near3 nb;
                nearby(c, sh, base, s, false, nb);
                if (nb.n) { want_launch = 1; launch_tgt = nb.i0; wait_after = 5; }
            }
        } else if (c.level == 2) {
            /* env_hetero.py:125-136 */
            { int b = HH_AC_BURST(m.ac_type); m.burst = m.cannon_remain < b ? m.cannon_remain : b; }
            bool man = ar.steps <= 5;
            if (!man) man = (ar.steps % hh_rng_randint(d_rng(ar, id, HH_SITE_L2_PERIOD, 0), 35, 45)) <= 5;
            if (man) {
                int r = hh_rng_randint(d_rng(ar, id, HH_SITE_L2_TURN, 0), 0, 1);
                m.cmd_hdg = hh_pymod(m.hdg + (r ? -90.0 : 90.0), 360.0);
                m.cmd_spd = (double)(100 + hh_rng_randint(d_rng(ar, id, HH_SITE_L2_SPEED, 0), 0, 4) * 75);
            }
            if (!m.has_missile && (ar.steps % 40) < 3 && hh_rng_randint(d_rng(ar, id, HH_SITE_L12_COIN, 0), 0, 1) &&
                m.missile_wait == 0 && m.ac_type == 1) {
                Near3 nb;
                nearby(c, sh, base, s, false, nb);
                if (nb.n) { want_launch = 1; launch_tgt = nb.i0; wait_after = 5; }
            }
        }
    }
    /* env_hetero.py:138-158 level 3: the arena-level escape flag is consumed once per live
     * opponent in id order; every lane replays the tiny integer sequence so that
     * each opponent lane sees the flag as it was at its turn and all lanes agree on the result */
    if (running && !c.ext_opp && c.level >= 3 && !hl) {
        int esc = ar.escaping, esc_t = ar.escaping_time;
        bool my_escaping = false;
#pragma unroll
        for (int j = 0; j < A; j++) {
            if (j < c.nA) continue;
            if (!sh_alive(sh, base + j)) continue;
            if (ar.steps % 60 == 0 && !esc) {
                esc = hh_rng_randint(d_rng(ar, j + 1, HH_SITE_L3_ESC_COIN, 0), 0, 1);
                if (esc) esc_t = (int)hh_rng_uniform(d_rng(ar, j + 1, HH_SITE_L3_ESC_TIME, 0), 20.0, 30.0);
            }
            if (j == s) my_escaping = esc != 0;
            if (esc) {
                esc_t -= 1;
                if (esc_t <= 0) esc = 0;
            }
        }
        ar.escaping = esc;
        ar.escaping_time = esc_t;
        if (snap && !agent) {
            int opp = -1, fire = 0, fire_m = 0;
            double heading, speed;
            if (my_escaping) {
                /* env_hetero.py:227-245 _escaping_opp */
                double y = hh_clip((m.lat - HH_MAP_LAT0) / c.ext_lat, 0.0, 1.0);
                double x = hh_clip((m.lon - HH_MAP_LON0) / c.ext_lon, 0.0, 1.0);
                double uh = d_rng(ar, id, HH_SITE_ESC_HDG, 0);
                if (y < 0.5) heading = x < 0.5 ? (double)(int)hh_rng_uniform(uh, 30.0, 60.0) : (double)(int)hh_rng_uniform(uh, 300.0, 330.0);
                else heading = x < 0.5 ? (double)(int)hh_rng_uniform(uh, 120.0, 150.0) : (double)(int)hh_rng_uniform(uh, 210.0, 240.0);
                speed = (double)(int)hh_rng_uniform(d_rng(ar, id, HH_SITE_ESC_SPEED, 0), 300.0, 600.0);
                fire = hh_rng_randint(d_rng(ar, id, HH_SITE_ESC_FIRE, 0), 0, 1);
            } else {
                /* env_hetero.py:247-271 _hardcoded_opp */
                Near3 nb;
                nearby(c, sh, base, s, false, nb);
                heading = m.hdg;
                speed = (double)(int)hh_rng_uniform(d_rng(ar, id, HH_SITE_HC_SPEED1, 0), 100.0, 400.0);
                if (nb.n) {
                    int ag = base + nb.i0;
                    /* env_base.py:464-487 _correct_angle_sign */
                    double sn, cs;
                    hh_sincos(hh_pymod(m.hdg, 360.0) * (HH_PI / 180.0), &sn, &cs);
                    double x1 = m.lon + hh_round3(sn), y1 = m.lat + hh_round3(cs);
                    double val = (x1 - m.lon) * (sh.lat0[ag] - m.lat) - (sh.lon0[ag] - m.lon) * (y1 - m.lat);
                    double sign = val < 0.0 ? 1.0 : -1.0;
                    double r = hh_rng_uniform(d_rng(ar, id, HH_SITE_HC_R, 0), 0.7, 1.3);
                    double focus = focus_deg(sh, base + s, ag);
                    if (nb.d0 > 0.008 && focus > 4.0) heading = hh_pymod(heading + r * sign * focus, 360.0);
                    if (nb.d0 > 0.05) {
                        double us = d_rng(ar, id, HH_SITE_HC_SPEED2, 0);
                        speed = focus < 30.0 ? (double)(int)hh_rng_uniform(us, 500.0, 800.0) : (double)(int)hh_rng_uniform(us, 100.0, 500.0);
                    }
                    fire = nb.d0 < 0.03 && focus < 10.0;
                    fire_m = nb.d0 < 0.09 && focus < 5.0;
                    opp = nb.i0;
                }
                if (m.ac_type == 2) speed = hh_clip(speed, 0.0, 600.0);
            }
            if (heading >= 360.0 || heading < 0.0) heading = 0.0;
            m.cmd_hdg = heading;
            m.cmd_spd = speed;
            if (fire) { int b = HH_AC_BURST(m.ac_type); m.burst = m.cannon_remain < b ? m.cannon_remain : b; }
            if (fire_m && opp >= 0 && !m.has_missile && m.missile_wait == 0 && m.ac_type == 1) {
                want_launch = 1; launch_tgt = opp; wait_after = 10;
            }
        }
    }

    /* ---------------- phase A2: missile envelope (ac1.py:72-79,144-146) ---------------- */
    int launched = 0;
    if (want_launch && !m.has_missile && m.missile_remain > 0) {
        double km, brg;
        d_dist_bearing(m.lat, m.lon, sh.lat0[base + launch_tgt], sh.lon0[base + launch_tgt], km, brg);
        if (km <= HH_MISSILE_RANGE_KM) {
            double delta = hh_fabs(d_signed_heading_diff(d_normalize_angle(m.hdg + HH_MISSILE_HALF_DEG), brg));
            if ((int)delta <= (int)HH_MISSILE_HALF_DEG) {
                launched = 1;
                m.rk_alive = 1; m.rk_lat = m.lat; m.rk_lon = m.lon; m.rk_hdg = m.hdg; m.rk_cmd = m.hdg;
                m.rk_target = launch_tgt + 1; m.rk_life = 0;
                m.has_missile = 1;
                m.missile_remain = m.missile_remain - 1 > 0 ? m.missile_remain - 1 : 0;
                evm |= 1u << (24 + s);
            }
        }
    }
    if (base_gate) {
        double uu = d_rng(ar, id, HH_SITE_MISSILE_WAIT, 0);
        m.missile_wait = hl ? hh_rng_randint(uu, 8, 12) : hh_rng_randint(uu, 7, 17);
        if (agent && !hl && c.agent_mode == HH_MODE_ESCAPE && m.missile_remain < 3) out.reward -= 0.1;
    }
    if (snap && (agent || c.ext_opp)) {
        if (m.missile_wait > 0 && !m.has_missile) m.missile_wait -= 1;
    }
    if (want_launch && wait_after >= 0) m.missile_wait = wait_after;
    /* launch order = unit id order (cmano_simulator.py:104-108): seq = running id counter */
    sh.aux[tid] = launched;
    __syncthreads();
    {
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            int l = active ? sh.aux[base + j] : 0;
            total += l;
            if (j < s) before += l;
        }
        if (launched) m.rk_seq = ar.next_seq + before + 1;
        ar.next_seq += total;
    }
    __syncthreads();

    /* ---------------- phase B: aircraft kinematics + move (ac1.py:81-133) ---------------- */
    const double lat_old = m.lat, lon_old = m.lon;
    bool fired = false;
    const int rk_at_start = m.rk_alive; /* rockets alive at tick start incl. just launched */
    if (snap) {
        int t = m.ac_type;
        if (m.hdg != m.cmd_hdg) {
            double delta = d_signed_heading_diff(m.hdg, m.cmd_hdg);
            double max_deg = HH_AC_TURN_RATE(t) * 1.0;
            if (hh_fabs(delta) <= max_deg) m.hdg = m.cmd_hdg;
            else { m.hdg += delta >= 0.0 ? max_deg : -max_deg; m.hdg = hh_pymod(m.hdg, 360.0); }
        }
        if (m.spd != m.cmd_spd) {
            double delta = m.cmd_spd - m.spd;
            double max_delta = HH_AC_ACCEL(t) * 1.0;
            if (hh_fabs(delta) <= max_delta) m.spd = m.cmd_spd;
            else m.spd += delta >= 0.0 ? max_delta : -max_delta;
        }
        if (m.burst > 0) {
            fired = true;
            m.burst = m.burst - 1 > 0 ? m.burst - 1 : 0;
            m.cannon_remain = m.cannon_remain - 1 > 0 ? m.cannon_remain - 1 : 0;
        }
        if (m.has_missile) { /* ac1.py:117-128 */
            if (!m.rk_alive) m.has_missile = 0;
            else m.rk_cmd = hh_clip(m.rk_hdg * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
        }
        if (m.spd > 0.0) hh_geo_direct(m.lat, m.lon, m.hdg, m.spd * HH_KNOTS_TO_MS * 1.0, &m.lat, &m.lon);
    }
    sh.lat1[tid] = m.lat;
    sh.lon1[tid] = m.lon;
    sh.aux[tid] = snap ? 1 : 0;
    __syncthreads();

    /* ---------------- phase B2: cannon candidates (ac1.py:106-115,135-142) ---------------- */
    int cand = 0;
    if (fired) {
        int t = m.ac_type;
#pragma unroll
        for (int j = 0; j < A; j++) {
            if (j == s) continue;
            if (!sh.aux[base + j]) continue; /* not alive at tick start -> can never be "currently alive" */
            bool enemy = agent ? (j >= c.nA) : (j < c.nA);
            if (!(c.friendly_kill || enemy)) continue;
            /* target already moved iff its id is lower (cmano_simulator.py:142) */
            double tl = j < s ? sh.lat1[base + j] : sh.lat0[base + j];
            double to = j < s ? sh.lon1[base + j] : sh.lon0[base + j];
            if (!d_maybe_within_km(lat_old, lon_old, tl, to, HH_AC_CANNON_KM(t))) continue;
            double km, brg;
            d_dist_bearing(lat_old, lon_old, tl, to, km, brg);
            if (km < HH_AC_CANNON_KM(t)) {
                double d = hh_fabs(d_signed_heading_diff(m.hdg, brg));
                if (d <= HH_AC_CANNON_HALF(t)) {
                    if (d_rng(ar, id, HH_SITE_CANNON, j + 1) < HH_AC_HIT_PROB(t)) cand |= 1 << j;
                }
            }
        }
    }
    __syncthreads(); /* everyone has read aux (snapshot flags) */
    sh.aux[tid] = cand;
    __syncthreads();

    /* ---------------- phase C: id-ordered cannon kill resolution (one lane per arena) ---------------- */
    if (s == 0 && active) {
        int alive = 0, nev = 0;
#pragma unroll
        for (int j = 0; j < A; j++) alive |= (sh_alive(sh, base + j) ? 1 : 0) << j;
        if (running) {
#pragma unroll
            for (int i = 0; i < A; i++) {
                int ci = sh.aux[base + i];
#pragma unroll
                for (int j = 0; j < A; j++) {
                    if (((ci >> j) & 1) && ((alive >> j) & 1)) {
                        alive &= ~(1 << j);
                        sh.g_ev[g][nev++] = i | (j << 4);
                    }
                }
            }
        }
        sh.g_alive[g] = alive;
        sh.g_nev[g] = nev;
    }
    __syncthreads();

    /* ---------------- phase D: rockets (rocket_unit.py:37-73) ---------------- */
    int rkw = 0; /* bit0 present, bit1 fuse on target, bit2 fuse on "friendly", bit3 end of life, bits4-6 target, bits 8.. seq */
    double rk_nlat = m.rk_lat, rk_nlon = m.rk_lon, rk_nhdg = m.rk_hdg;
    if (running && rk_at_start) {
        int tg = m.rk_target - 1;
        int hit_t = 0, hit_f = 0;
        double tl = sh.lat1[base + tg], to = sh.lon1[base + tg];
        if (d_maybe_within_km(m.rk_lat, m.rk_lon, tl, to, HH_ROCKET_FUSE_KM)) {
            double km, brg;
            d_dist_bearing(m.rk_lat, m.rk_lon, tl, to, km, brg);
            hit_t = km < HH_ROCKET_FUSE_KM;
        }
        if (c.friendly_kill) {
            int fid = s == 1 ? 0 : 1; /* rocket_unit.py:46: 1 if source.id == 2 else 2 */
            double fl = sh.lat1[base + fid], fo = sh.lon1[base + fid];
            if (d_maybe_within_km(m.rk_lat, m.rk_lon, fl, fo, HH_ROCKET_FUSE_KM)) {
                double km, brg;
                d_dist_bearing(m.rk_lat, m.rk_lon, fl, fo, km, brg);
                hit_f = km < HH_ROCKET_FUSE_KM;
            }
        }
        int eol = m.rk_life > HH_ROCKET_MAX_LIFE;
        rkw = 1 | (hit_t << 1) | (hit_f << 2) | (eol << 3) | (tg << 4) | (m.rk_seq << 8);
        if (!eol) { /* speculative turn + move; committed only if the rocket survives resolution */
            const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
            if (rk_nhdg != m.rk_cmd) {
                double delta = d_signed_heading_diff(rk_nhdg, m.rk_cmd);
                if (hh_fabs(delta) <= HH_ROCKET_TURN_RATE) rk_nhdg = m.rk_cmd;
                else rk_nhdg += delta >= 0.0 ? HH_ROCKET_TURN_RATE : -HH_ROCKET_TURN_RATE;
            }
            double spd = speed_table[m.rk_life];
            if (spd > 0.0) hh_geo_direct(m.rk_lat, m.rk_lon, rk_nhdg, spd * HH_KNOTS_TO_MS * 1.0, &rk_nlat, &rk_nlon);
        }
    }
    sh.aux[tid] = rkw;
    __syncthreads();
    if (s == 0 && active) {
        int alive = sh.g_alive[g], nev = sh.g_nev[g], dead = 0, done_mask = 0;
        if (running) {
            for (int k = 0; k < A; k++) { /* launch (= id) order: pick the smallest unprocessed seq */
                int best = -1, best_seq = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    int w = sh.aux[base + j];
                    if ((w & 1) && !((done_mask >> j) & 1) && (w >> 8) < best_seq) { best = j; best_seq = w >> 8; }
                }
                if (best < 0) break;
                done_mask |= 1 << best;
                int w = sh.aux[base + best];
                int tg = (w >> 4) & 7;
                int fid = best == 1 ? 0 : 1;
                if (((w >> 1) & 1) && ((alive >> tg) & 1)) {
                    alive &= ~(1 << tg); dead |= 1 << best;
                    sh.g_ev[g][nev++] = best | (tg << 4) | (1 << 8);
                } else if (c.friendly_kill && ((alive >> fid) & 1) && ((w >> 2) & 1)) {
                    alive &= ~(1 << fid); dead |= 1 << best;
                    sh.g_ev[g][nev++] = best | (fid << 4) | (1 << 8);
                } else if ((w >> 3) & 1) {
                    dead |= 1 << best;
                }
            }
        }
        sh.g_alive[g] = alive;
        sh.g_nev[g] = nev;
        sh.g_rkdead[g] = dead;
    }
    __syncthreads();
    if (running && rk_at_start) {
        if ((sh.g_rkdead[g] >> s) & 1) {
            m.rk_alive = 0; m.rk_target = 0; m.rk_life = 0; m.rk_seq = 0;
            m.rk_lat = m.rk_lon = m.rk_hdg = m.rk_cmd = 0.0;
        } else {
            m.rk_lat = rk_nlat; m.rk_lon = rk_nlon; m.rk_hdg = rk_nhdg; m.rk_life += 1;
        }
    }

    /* ---------------- phase E: out of bounds, rewards, done (env_base.py:240-310, env_hetero.py:188-225) ---------------- */
    int oob = 0;
    if (active) {
        m.alive = (sh.g_alive[g] >> s) & 1;
        if (running && m.alive) {
            bool inb = HH_MAP_LON0 <= m.lon && m.lon <= c.lon_hi && HH_MAP_LAT0 <= m.lat && m.lat <= c.lat_hi;
            if (!inb) { m.alive = 0; oob = 1; }
        }
    }
    double rews = 0.0;
    int destroyed = 0;
    if (running && agent) {
        double sc = c.rew_scale;
        if (oob) { rews += (hl ? -2.0 : -5.0) * sc; destroyed = 1; }
        int nev = sh.g_nev[g];
        for (int e = 0; e < nev; e++) {
            int w = sh.g_ev[g][e];
            int k = w & 15, d = (w >> 4) & 15, rocket = (w >> 8) & 1;
            if (k < c.nA) {
                if (d >= c.nA) {
                    if (k == s) {
                        if (!hl) {
                            if (c.agent_mode == HH_MODE_FIGHT) {
                                if (rocket) {
                                    rews += (1.0 + ((1.5 - 1.0) / (1.0 - 0.0)) * ((double)m.missile_remain / (double)m.rocket_max - 0.0)) * sc;
                                } else {
                                    double r1 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * ((double)m.cannon_remain / (double)m.cannon_max - 0.0);
                                    double r2 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * (opp_stat0 - 0.0);
                                    rews += (r1 + r2) * sc;
                                }
                            }
                        } else {
                            rews += 1.0;
                        }
                    }
                } else if (!hl) {
                    if (k == s) rews += -2.0 * sc;
                    if (c.friendly_punish && d == s) { rews += -2.0 * sc; destroyed = 1; }
                }
            } else if (d < c.nA) {
                if (d == s) { rews += (hl ? -1.0 : -2.0) * sc; destroyed = 1; }
            }
        }
    }
    sh.aux[tid] = oob;
    sh.rew[tid] = rews;
    /* publish the post-tick state for escape shaping and the observation */
    publish(sh, tid, m);
    __syncthreads();
    if (running && agent && !hl && c.agent_mode == HH_MODE_ESCAPE && c.esc_dist_rew && m.alive) {
        /* env_hetero.py:198-214 */
        Near3 nb;
        nearby(c, sh, base, s, false, nb);
        double dr[3] = {nb.r0, nb.r1, nb.r2};
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            if (j > nb.n) break;
            if (dr[j - 1] < 0.06) { rews += -0.02 / j; if (m.spd < 200.0) rews += -0.02 / j; }
            else if (dr[j - 1] > 0.13) { rews += 0.02 / j; if (m.spd > 500.0) rews += 0.02 / j; }
        }
    }
    if (running && agent) {
        if (m.alive || destroyed) {
            if (c.glob_frac > 0.0 && !hl && c.agent_mode == HH_MODE_FIGHT) {
                out.reward += rews + c.glob_frac * sh.rew[base + ((s + 1) % 2)];
            } else if (c.glob_frac > 0.0 && hl) {
                double other = 0.0;
#pragma unroll
                for (int j = 0; j < A; j++) if (j < c.nA && j != s) other += sh.rew[base + j];
                out.reward += rews + c.glob_frac * other;
            } else {
                out.reward += rews;
            }
        }
    }
    /* event masks for parity checks */
    if (active) {
        int nev = sh.g_nev[g];
        for (int e = 0; e < nev; e++) {
            int w = sh.g_ev[g][e];
            evm |= ((w >> 8) & 1) ? (1u << (8 + ((w >> 4) & 15))) : (1u << ((w >> 4) & 15));
        }
        if (oob) evm |= 1u << (16 + s);
    }
    ev_mask_out = evm;
    if (running) {
        int ag = 0, op = 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            int al = ((sh.g_alive[g] >> j) & 1) && !sh.aux[base + j];
            if (j < c.nA) ag += al; else op += al;
        }
        ar.done = (ag <= 0 || op <= 0 || ar.steps >= c.horizon) ? 1 : 0;
    }
    __syncthreads(); /* all reads of rew/aux/g_* done before the caller reuses them */
}

/* ===================================================================== kernels */
enum { HH_RUN_ROLLOUT = 0, HH_RUN_RESET = 1, HH_RUN_OBSERVE = 2 };

template <int A, int B>
__global__ __launch_bounds__(B) void hh_k_world(DevPtrs P, DevCfg c, int run, int T, const int8_t *__restrict__ actions,
                                                const uint8_t *__restrict__ mask, float *__restrict__ obs_out,
                                                float *__restrict__ reward_out, uint8_t *__restrict__ valid_out,
                                                uint8_t *__restrict__ done_out) {
    constexpr int GPB = B / A;
    __shared__ Shared<A, B> sh;
    const int tid = threadIdx.x;
    const int g = tid / A, s = tid % A;
    const int base = g * A;
    const int n = blockIdx.x * GPB + g;
    const bool active = g < GPB && n < c.N;
    const size_t U = (size_t)c.N * A;
    const size_t u = (size_t)n * A + s;
    const int D = c.D;
    Unit m = Unit{};
    Arena ar = Arena{};
    double ep_ret = 0.0;
    if (active) {
        unit_load(P, U, u, m);
        arena_load(P, c, n, ar);
        if (s == 0) ep_ret = P.ep_ret[n];
    } else {
        ar.done = 1;
    }
    sh.flags[tid] = 0;
    sh.aux[tid] = 0;
    bool need_reset = run == HH_RUN_RESET && active && (mask == nullptr || mask[n]);
    uint32_t evm_last = 0;
    for (int t = 0; t < T; t++) {
        StepOut so;
        so.reward = 0.0;
        so.valid = 0;
        int done_flag = ar.done;
        if (run == HH_RUN_ROLLOUT) {
            if (t == 0) { publish(sh, tid, m); __syncthreads(); }
            int8_t act[4] = {0, 0, 0, 0};
            if (active && s < c.n_ctrl) {
                const int8_t *ap = actions + (((size_t)t * c.N + n) * c.n_ctrl + s) * 4;
                int w = *reinterpret_cast<const int *>(ap);
                act[0] = (int8_t)(w & 0xff); act[1] = (int8_t)((w >> 8) & 0xff); act[2] = (int8_t)((w >> 16) & 0xff); act[3] = (int8_t)((w >> 24) & 0xff);
            }
            const bool was_running = active && !ar.done;
            tick<A, B>(c, sh, tid, g, s, base, active, m, ar, act, so, evm_last);
            done_flag = ar.done;
            /* outputs of this tick */
            if (active && s < c.nA) {
                size_t o = ((size_t)t * c.N + n) * c.nA + s;
                if (reward_out) reward_out[o] = (float)so.reward;
                if (valid_out) valid_out[o] = (uint8_t)so.valid;
            }
            /* episode statistics: one lane per arena, agent order */
            sh.rew[tid] = so.valid ? so.reward : 0.0;
            __syncthreads();
            if (active && s == 0 && was_running) {
                for (int j = 0; j < c.nA; j++) ep_ret += sh.rew[base + j];
                if (ar.done) {
                    int ag = 0, op = 0;
                    for (int j = 0; j < A; j++) { int al = sh_alive(sh, base + j); if (j < c.nA) ag += al; else op += al; }
                    P.last_ret[n] = (float)ep_ret;
                    P.last_len[n] = ar.steps;
                    P.last_outcome[n] = (op <= 0 && ar.steps < c.horizon) ? 1 : ((ag <= 0 && ar.steps < c.horizon) ? -1 : 0);
                }
            }
            if (active && s == 0 && done_out) done_out[(size_t)t * c.N + n] = (uint8_t)done_flag;
            need_reset = active && ar.done && c.auto_reset;
        }
        const int any_reset = __syncthreads_or(need_reset ? 1 : 0);
        if (need_reset) { /* K3 */
            reset_arena_scalars(ar);
            reset_unit<A>(c, s, m, ar);
            ep_ret = 0.0;
        }
        if (run != HH_RUN_ROLLOUT || any_reset) {
            /* state changed (or was never published): publish for the observation */
            publish(sh, tid, m);
            __syncthreads();
        }
        need_reset = false;
        /* K2: observation rows staged in LDS, then written with unit-stride stores */
        if (active && s < c.nA) lowlevel_obs<A, B>(c, sh, base, s, c.agent_mode, m, &sh.obs[(g * c.nA + s) * D], D);
        __syncthreads();
        if (obs_out) {
            const int rows = min(GPB, c.N - (int)blockIdx.x * GPB);
            const int cnt = rows * c.nA * D;
            float *dst = obs_out + ((size_t)t * c.N + (size_t)blockIdx.x * GPB) * c.nA * D;
            if (run == HH_RUN_RESET && mask != nullptr) {
                for (int k = tid; k < cnt; k += B) if (mask[blockIdx.x * GPB + k / (c.nA * D)]) dst[k] = sh.obs[k];
            } else {
                for (int k = tid; k < cnt; k += B) dst[k] = sh.obs[k];
            }
        }
        __syncthreads();
    }
    if (active) {
        unit_store(P, U, u, m);
        if (s == 0) {
            arena_store(P, n, ar);
            P.ep_ret[n] = ep_ret;
            if (run == HH_RUN_ROLLOUT) P.ev_mask[n] = 0;
        }
    }
    if (run == HH_RUN_ROLLOUT) {
        /* OR-reduce the per-lane event bits of the last tick into the arena word */
        __syncthreads();
        if (active && evm_last) atomicOr(&P.ev_mask[n], evm_last);
    }
}

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

struct hh_world {
    hh_config cfg;
    DevCfg dc;
    DevPtrs P;
    int device;
    int block; /* threads per workgroup */
    void *slab;
    size_t slab_bytes;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int hh_world_create(const hh_config *cfg, int device, hh_world **out) {
    if (!cfg || !out) { g_err = "null argument"; return HH_E_ARG; }
    int A = cfg->n_agents + cfg->n_opps;
    if (cfg->n_arenas <= 0 || cfg->n_agents < 1 || cfg->n_opps < 1 || (A != 4 && A != 6)) {
        g_err = "unsupported configuration: need n_arenas > 0 and 2-vs-2 or 3-vs-3";
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
    HIPCHK(hipSetDevice(device));
    hh_world *w = new hh_world();
    w->cfg = *cfg;
    w->device = device;
    DevCfg &d = w->dc;
    d.N = cfg->n_arenas; d.env_kind = cfg->env_kind; d.nA = cfg->n_agents; d.nO = cfg->n_opps; d.A = A;
    d.level = cfg->level; d.agent_mode = cfg->agent_mode; d.horizon = cfg->horizon;
    d.friendly_kill = cfg->friendly_kill; d.friendly_punish = cfg->friendly_punish; d.esc_dist_rew = cfg->esc_dist_rew;
    d.hier_action_assess = cfg->hier_action_assess; d.hier_opp_fight_ratio = cfg->hier_opp_fight_ratio;
    d.auto_reset = cfg->auto_reset; d.ext_opp = cfg->ext_opp_actions;
    d.D = cfg->env_kind == HH_ENV_HIGHLEVEL ? HH_OBS_HL : (cfg->agent_mode == HH_MODE_FIGHT ? HH_OBS_FIGHT_AC1 : HH_OBS_ESC_AC1);
    d.n_ctrl = cfg->ext_opp_actions ? A : cfg->n_agents;
    d.glob_frac = cfg->glob_frac; d.rew_scale = cfg->rew_scale;
    double ms = cfg->map_size;
    d.lat_hi = HH_MAP_LAT0 + ms; d.lon_hi = HH_MAP_LON0 + ms;
    d.ext_lat = d.lat_hi - HH_MAP_LAT0; d.ext_lon = d.lon_hi - HH_MAP_LON0;
    d.inv_diag = (1.0 - 0.0) / (__builtin_sqrt(2.0 * (ms * ms)) - 0.0);
    d.seed = cfg->seed; d.arena_offset = cfg->arena_offset;
    w->block = HH_BLOCK;
    /* one slab, 256-byte aligned sub-arrays */
    size_t U = (size_t)d.N * A, N = (size_t)d.N;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_f[6]; for (int k = 0; k < 6; k++) o_f[k] = take(U * 8);
    size_t o_pack = take(U * 16), o_tgt = take(3 * U * 8);
    size_t o_rk[4]; for (int k = 0; k < 4; k++) o_rk[k] = take(U * 8);
    size_t o_rkp = take(U * 8), o_ar = take(N * 16), o_ep = take(N * 8), o_lr = take(N * 4), o_ll = take(N * 4);
    size_t o_lo = take(N), o_ev = take(N * 4);
    w->slab_bytes = off;
    HIPCHK(hipMalloc(&w->slab, off));
    HIPCHK(hipMemset(w->slab, 0, off));
    char *b = (char *)w->slab;
    DevPtrs &P = w->P;
    P.lat = (double *)(b + o_f[0]); P.lon = (double *)(b + o_f[1]); P.hdg = (double *)(b + o_f[2]);
    P.spd = (double *)(b + o_f[3]); P.cmd_hdg = (double *)(b + o_f[4]); P.cmd_spd = (double *)(b + o_f[5]);
    P.pack = (int4 *)(b + o_pack); P.tgt_d = (double *)(b + o_tgt);
    P.rk_lat = (double *)(b + o_rk[0]); P.rk_lon = (double *)(b + o_rk[1]); P.rk_hdg = (double *)(b + o_rk[2]); P.rk_cmd = (double *)(b + o_rk[3]);
    P.rk_pack = (int2 *)(b + o_rkp); P.ar_pack = (int4 *)(b + o_ar); P.ep_ret = (double *)(b + o_ep);
    P.last_ret = (float *)(b + o_lr); P.last_len = (int *)(b + o_ll); P.last_outcome = (int8_t *)(b + o_lo);
    P.ev_mask = (uint32_t *)(b + o_ev);
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

extern "C" int hh_world_destroy(hh_world *w) {
    if (!w) return HH_E_ARG;
    hipSetDevice(w->device);
    hipFree(w->slab);
    delete w;
    return HH_OK;
}

extern "C" int hh_obs_dim(const hh_world *w) { return w ? w->dc.D : HH_E_ARG; }
extern "C" int hh_n_ctrl(const hh_world *w) { return w ? w->dc.n_ctrl : HH_E_ARG; }

static int launch(hh_world *w, int run, int T, const int8_t *actions, const uint8_t *mask, float *obs, float *reward,
                  uint8_t *valid, uint8_t *done, hipStream_t st) {
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) { g_err = "HighLevelEnv stepping goes through hh_hl_* (not built in this round)"; return HH_E_ARG; }
    const DevCfg &c = w->dc;
    if (c.A == 4) {
        constexpr int B = HH_BLOCK, GPB = B / 4;
        int grid = (c.N + GPB - 1) / GPB;
        hipLaunchKernelGGL((hh_k_world<4, B>), dim3(grid), dim3(B), 0, st, w->P, c, run, T, actions, mask, obs, reward, valid, done);
    } else {
        constexpr int B = HH_BLOCK, GPB = B / 6;
        int grid = (c.N + GPB - 1) / GPB;
        hipLaunchKernelGGL((hh_k_world<6, B>), dim3(grid), dim3(B), 0, st, w->P, c, run, T, actions, mask, obs, reward, valid, done);
    }
    HIPCHK(hipGetLastError());
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
    hipStream_t st = (hipStream_t)stream;
    size_t N = (size_t)w->dc.N;
    if (ret) HIPCHK(hipMemcpyAsync(ret, w->P.last_ret, N * 4, hipMemcpyDeviceToDevice, st));
    if (len) HIPCHK(hipMemcpyAsync(len, w->P.last_len, N * 4, hipMemcpyDeviceToDevice, st));
    if (outcome) HIPCHK(hipMemcpyAsync(outcome, w->P.last_outcome, N, hipMemcpyDeviceToDevice, st));
    return HH_OK;
}

extern "C" int hh_get_event_masks(hh_world *w, uint32_t *masks) {
    if (!w || !masks) return HH_E_ARG;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(masks, w->P.ev_mask, (size_t)w->dc.N * 4, hipMemcpyDeviceToHost));
    return HH_OK;
}

extern "C" int hh_get_state(hh_world *w, hh_state_view *v) {
    if (!w || !v) return HH_E_ARG;
    HIPCHK(hipSetDevice(w->device));
    HIPCHK(hipDeviceSynchronize());
    const DevCfg &c = w->dc;
    size_t U = (size_t)c.N * c.A, N = (size_t)c.N;
    std::vector<double> f[6], tg(3 * U), rk[4];
    std::vector<int4> pack(U), ar(N);
    std::vector<int2> rkp(U);
    const double *src[6] = {w->P.lat, w->P.lon, w->P.hdg, w->P.spd, w->P.cmd_hdg, w->P.cmd_spd};
    const double *rsrc[4] = {w->P.rk_lat, w->P.rk_lon, w->P.rk_hdg, w->P.rk_cmd};
    for (int k = 0; k < 6; k++) { f[k].resize(U); HIPCHK(hipMemcpy(f[k].data(), src[k], U * 8, hipMemcpyDeviceToHost)); }
    for (int k = 0; k < 4; k++) { rk[k].resize(U); HIPCHK(hipMemcpy(rk[k].data(), rsrc[k], U * 8, hipMemcpyDeviceToHost)); }
    HIPCHK(hipMemcpy(pack.data(), w->P.pack, U * 16, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tg.data(), w->P.tgt_d, 3 * U * 8, hipMemcpyDeviceToHost));
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
        int ids[3] = {p.w & 0xff, (p.w >> 8) & 0xff, (p.w >> 16) & 0xff};
        for (int k = 0; k < HH_TGT_K; k++) {
            v->tgt_id[u * HH_TGT_K + k] = k < n_tgt ? ids[k] : 0;
            v->tgt_d[u * HH_TGT_K + k] = k < n_tgt ? tg[(size_t)k * U + u] : 0.0;
        }
        int2 r = rkp[u];
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
    HIPCHK(hipSetDevice(w->device));
    HIPCHK(hipDeviceSynchronize());
    const DevCfg &c = w->dc;
    size_t U = (size_t)c.N * c.A, N = (size_t)c.N;
    std::vector<double> f[6], tg(3 * U), rk[4];
    std::vector<int4> pack(U), ar(N);
    std::vector<int2> rkp(U);
    for (int k = 0; k < 6; k++) f[k].resize(U);
    for (int k = 0; k < 4; k++) rk[k].resize(U);
    for (size_t u = 0; u < U; u++) {
        for (int k = 0; k < 6; k++) f[k][u] = v->ac_f[u * HH_ACF_K + k];
        const int32_t *q = v->ac_i + u * HH_ACI_K;
        int n_tgt = 0, ids[3];
        for (int k = 0; k < HH_TGT_K; k++) {
            ids[k] = v->tgt_id[u * HH_TGT_K + k];
            tg[(size_t)k * U + u] = v->tgt_d[u * HH_TGT_K + k];
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
        r.x = (rp[0] & 0xff) | ((rp[1] & 0xff) << 8) | ((rp[2] & 0xff) << 16);
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
    HIPCHK(hipMemcpy(w->P.tgt_d, tg.data(), 3 * U * 8, hipMemcpyHostToDevice));
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
