/*
 * hh_oracle.c — TEST INFRASTRUCTURE.  Sequential CPU restatement (plain C, one arena at a time,
 * IEEE double, reference statement order) of the reference's env-step path:
 *
 *   warsim/simulator/{cmano_simulator,ac1,ac2,rocket_unit}.py, warsim/utils/{angles,map_limits}.py,
 *   envs/env_base.py, envs/env_hetero.py (LowLevelEnv), envs/env_hier.py (HighLevelEnv).
 *
 * Every function cites the reference lines it follows.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libhh_oracle.so,
 * and only as the checker.  The product (hhmarl_2d_amd/csrc) is an independent, parallel
 * formulation and never calls into this file.
 *
 * How this oracle is pinned (parity is NOT pinned by the reference itself: it ships no tests
 * and its geodesic arithmetic lives in un-vendored geographiclib==2.0):
 *   - env/simulator logic: golden step traces recorded from the REAL reference, imported
 *     unchanged in the build container behind import stubs (oracle/ref_harness.py,
 *     oracle/gen_env_golden.py -> tests/golden/env_*.npz); tests/test_oracle_golden.py replays
 *     them here: integer state/masks bit-exact, floats <= 1e-9.
 *   - geodesic: include/hh_geodesic.h (Karney 2013) against an independent mpmath ODE
 *     integration and the paper's worked examples (tests/test_geodesic.py).
 *   - randomness: the keyed tape of include/hh_rng.h, patched into the reference by the harness.
 *
 * Shared with the product on purpose: include/hh_{spec,math,rng,geodesic}.h (constants, the
 * bit-reproducible FP64 primitives and the geodesic series) so that oracle and kernels execute
 * the same IEEE operation sequence and can be compared bit-for-bit.
 *
 * Build: oracle/Makefile  (gcc -O2 -mfma -ffp-contract=off -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "hh_abi.h"
#include "hh_envelope.h"
#include "hh_geodesic.h"
#include "hh_math.h"
#include "hh_rng.h"
#include "hh_spec.h"

#define MAXA HH_MAX_AIRCRAFT
#define O_TGT_K HH_TGT_K_WIDE /* internal capacity of a stored target list; the views carry w->tgt_k of them */
#define OBS_MAX 34

typedef struct {
    /* Position + Unit fields, cmano_simulator.py:25-32,55-63; per-type fields ac1.py:38-56, ac2.py:34-52 */
    double lat, lon, hdg, spd, cmd_hdg, cmd_spd;
    int alive, ac_type, cannon_remain, cannon_burst, cannon_max, missile_remain, rocket_max;
    int has_missile;  /* actual_missile is not None (may refer to an already removed rocket) */
    int missile_wait; /* env_base.py:72 self.missile_wait[i] */
} o_ac;

typedef struct {
    /* rocket_unit.py:23-30; slot = launcher slot */
    int alive, target, life, seq;
    double lat, lon, hdg, cmd_hdg;
} o_rk;

typedef struct {
    o_ac ac[MAXA];
    o_rk rk[MAXA];
    int steps, episode, escaping, escaping_time, next_seq, done;
    int tgt_n[MAXA], tgt_id[MAXA][O_TGT_K]; /* opp_to_attack (low level: first entry only) */
    double tgt_d[MAXA][O_TGT_K];
    uint64_t akey;
    /* outputs of the last step */
    double reward[MAXA];
    int reward_valid[MAXA];
    float obs[MAXA][OBS_MAX];
    uint32_t ev_mask;
    int act_fault; /* sticky: some consumed action word of this arena was out of range and ran sanitised */
    /* HighLevelEnv macro step (env_hier.py:114-140) */
    int hl_s, hl_running, hl_kill, hl_situ;
    int cmd_act[MAXA]; /* self.commander_actions[i]: 0 escape, k>0 fight stored target k (agents and opponents) */
    double opp_stat0[MAXA]; /* env_hetero.py:169-170 opp_stats[i][0], kept between the two halves of a split step */
    int eval_last[HH_EVAL_K], eval_tot[HH_EVAL_K]; /* env_base.py:91-107 info dict of the last commander step / summed */
    /* episode statistics */
    double ep_ret;
    float last_ret;
    int last_len, last_outcome;
} o_arena;

typedef struct {
    hh_config cfg;
    int A, D, n_ctrl;
    int tgt_k;  /* entries of a stored target list in the state views (hh_spec.h: HH_TGT_K_OF) */
    int slots;  /* unit slots of the device world's arenas (event-mask layout, hh_spec.h: HH_EV_BIT) */
    double ext_lat, ext_lon, inv_ext_lat, inv_ext_lon, lat_hi, lon_hi, inv_diag;
    o_arena *ar;
} o_world;

typedef struct { int origin_rocket, killer, destroyed; } o_event;

/* ------------------------------------------------------------------ small helpers */
static double rng_u(const o_arena *a, int unit, int site, int sub) {
    return hh_rng_u01(hh_rng_tick_key(a->akey, (uint32_t)a->episode, (uint32_t)a->steps), (uint32_t)unit,
                      (uint32_t)site, (uint32_t)sub);
}

/* warsim/utils/angles.py:10-15 */
static double normalize_angle(double a) {
    while (a >= 360.0) a -= 360.0;
    while (a < 0.0) a += 360.0;
    return a;
}
/* angles.py:22-29 */
static double signed_heading_diff(double actual, double desired) {
    double delta = desired - actual;
    if (delta < -180.0) delta = 360.0 + delta;
    if (delta > 180.0) delta = -360.0 + delta;
    return delta;
}

/* cmano_simulator.py:167-174 + geodesics.py:12-19: one Inverse solution gives both */
static void dist_bearing(double lat1, double lon1, double lat2, double lon2, double *km, double *brg) {
    double s12, azi1;
    hh_geo_inverse(lat1, lon1, lat2, lon2, &s12, &azi1);
    *km = s12 / 1000.0;
    *brg = normalize_angle(azi1);
}

/* env_base.py:424-432 _focus_angle (degrees) */
static double focus_deg(const o_ac *a, const o_ac *b) {
    double ang = hh_pymod(90.0 - a->hdg, 360.0) * (HH_PI / 180.0);
    double s, c;
    hh_sincos(ang, &s, &c);
    double dx = b->lon - a->lon, dy = b->lat - a->lat;
    double dot = c * dx + s * dy;
    double n1 = hh_sqrt(c * c + s * s), n2 = hh_sqrt(dx * dx + dy * dy);
    double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
    return hh_acos(x) * (180.0 / HH_PI);
}
static double focus_norm(const o_ac *a, const o_ac *b) { return hh_clip(HH_DIVC(focus_deg(a, b), 180.0), 0.0, 1.0); }
/* env_base.py:441-446 _aspect_angle(norm=True) */
static double aspect_norm(const o_ac *a, const o_ac *b) { return hh_clip(HH_DIVC(180.0 - focus_deg(a, b), 180.0), 0.0, 1.0); }
/* env_base.py:448-456 _heading_diff(norm=True) */
static double heading_diff_norm(const o_ac *a, const o_ac *b) {
    double s1, c1, s2, c2;
    hh_sincos(hh_pymod(90.0 - a->hdg, 360.0) * (HH_PI / 180.0), &s1, &c1);
    hh_sincos(hh_pymod(90.0 - b->hdg, 360.0) * (HH_PI / 180.0), &s2, &c2);
    double dot = c1 * c2 + s1 * s2;
    double n1 = hh_sqrt(c1 * c1 + s1 * s1), n2 = hh_sqrt(c2 * c2 + s2 * s2);
    double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
    return hh_clip(HH_DIVC(hh_acos(x) * (180.0 / HH_PI), 180.0), 0.0, 1.0);
}
/* env_base.py:434-439 _distance */
static double dist_raw(const o_ac *a, const o_ac *b) { return hh_hypot(b->lon - a->lon, b->lat - a->lat); }
static double dist_norm(const o_world *w, const o_ac *a, const o_ac *b) { return w->inv_diag * dist_raw(a, b); }

static int is_agent(const o_world *w, int id) { return id <= w->cfg.n_agents; }

/* env_base.py:400-422 _nearby_object: ids (1-based) of live enemies (or friends) sorted by
 * normalised distance, stable */
static int nearby(const o_world *w, const o_arena *a, int id, int friendly, int *ids, double *dn, double *dr) {
    int n = 0, lo, hi;
    if (friendly) {
        lo = is_agent(w, id) ? 1 : w->cfg.n_agents + 1;
        hi = is_agent(w, id) ? w->cfg.n_agents : w->A;
    } else {
        lo = is_agent(w, id) ? w->cfg.n_agents + 1 : 1;
        hi = is_agent(w, id) ? w->A : w->cfg.n_agents;
    }
    for (int j = lo; j <= hi; j++) {
        if (j == id || !a->ac[j - 1].alive) continue;
        ids[n] = j;
        dn[n] = dist_norm(w, &a->ac[id - 1], &a->ac[j - 1]);
        dr[n] = dist_raw(&a->ac[id - 1], &a->ac[j - 1]);
        n++;
    }
    for (int i = 1; i < n; i++) { /* insertion sort = stable */
        int ti = ids[i]; double tn = dn[i], tr = dr[i];
        int k = i - 1;
        while (k >= 0 && dn[k] > tn) { ids[k + 1] = ids[k]; dn[k + 1] = dn[k]; dr[k + 1] = dr[k]; k--; }
        ids[k + 1] = ti; dn[k + 1] = tn; dr[k + 1] = tr;
    }
    return n;
}

/* map_limits.py:37-40 relative_position -> (lat_rel, lon_rel) */
static void rel_pos(const o_world *w, const o_ac *u, double *lat_rel, double *lon_rel) {
    *lat_rel = hh_clip(hh_div_known(u->lat - HH_MAP_LAT0, w->ext_lat, w->inv_ext_lat), 0.0, 1.0);
    *lon_rel = hh_clip(hh_div_known(u->lon - HH_MAP_LON0, w->ext_lon, w->inv_ext_lon), 0.0, 1.0);
}
/* map_limits.py:47-48 */
static int in_boundary(const o_world *w, const o_ac *u) {
    return HH_MAP_LON0 <= u->lon && u->lon <= w->lon_hi && HH_MAP_LAT0 <= u->lat && u->lat <= w->lat_hi;
}

static int shot_flag(const o_ac *u) { /* env_base.py:151-154,208-211 */
    int shot = u->cannon_burst > 0;
    if (u->ac_type == 1) shot = shot || u->has_missile;
    return shot;
}

/* ------------------------------------------------------------------ observations */
/* env_base.py:185-212 opp_ac_values; mode 0 fight, 1 esc, 2 HighLevel */
static int opp_ac_values(const o_world *w, const o_arena *a, int mode, int opp_id, int agent_id, double dist, double *st) {
    const o_ac *o = &a->ac[opp_id - 1], *s = &a->ac[agent_id - 1];
    int n = 0;
    double x, y;
    rel_pos(w, o, &x, &y);
    st[n++] = x;
    st[n++] = y;
    st[n++] = hh_clip(hh_div_known(o->spd, HH_AC_MAX_SPEED(o->ac_type), HH_AC_INV_MAX_SPEED(o->ac_type)), 0.0, 1.0);
    st[n++] = hh_clip(HH_DIVC(hh_pymod(o->hdg, 359.0), 359.0), 0.0, 1.0);
    st[n++] = heading_diff_norm(o, s);
    if (mode == 0) {
        st[n++] = focus_norm(o, s);
        st[n++] = aspect_norm(s, o);
    } else {
        st[n++] = focus_norm(s, o);
        st[n++] = focus_norm(o, s);
    }
    if (mode == 2) {
        st[n++] = aspect_norm(s, o);
        st[n++] = aspect_norm(o, s);
    }
    st[n++] = dist;
    if (mode != 2) st[n++] = (double)shot_flag(o);
    return n;
}

/* env_base.py:166-183 friendly_ac_values */
static int friendly_ac_values(const o_world *w, const o_arena *a, int agent_id, int fri_id, double *st) {
    for (int k = 0; k < 5; k++) st[k] = 0.0;
    if (fri_id && a->ac[fri_id - 1].alive) {
        const o_ac *f = &a->ac[fri_id - 1], *s = &a->ac[agent_id - 1];
        double x, y;
        rel_pos(w, f, &x, &y);
        st[0] = x;
        st[1] = y;
        st[2] = focus_norm(s, f);
        st[3] = focus_norm(f, s);
        st[4] = dist_norm(w, s, f);
    }
    return 5;
}

/* env_base.py:111-135 fight_state_values */
static int fight_state_values(const o_world *w, const o_arena *a, int id, int opp_id, double opp_dist, int fri_id, double *st) {
    const o_ac *u = &a->ac[id - 1], *o = &a->ac[opp_id - 1];
    int n = 0;
    double x, y;
    rel_pos(w, u, &x, &y);
    st[n++] = x;
    st[n++] = y;
    st[n++] = hh_clip(hh_div_known(u->spd, HH_AC_MAX_SPEED(u->ac_type), HH_AC_INV_MAX_SPEED(u->ac_type)), 0.0, 1.0);
    st[n++] = hh_clip(HH_DIVC(hh_pymod(u->hdg, 359.0), 359.0), 0.0, 1.0);
    st[n++] = focus_norm(u, o);
    st[n++] = aspect_norm(o, u);
    st[n++] = heading_diff_norm(u, o);
    st[n++] = opp_dist;
    st[n++] = hh_clip((double)u->cannon_remain / (double)u->cannon_max, 0.0, 1.0);
    if (u->ac_type == 1) {
        st[n++] = hh_clip((double)u->missile_remain / (double)u->rocket_max, 0.0, 1.0);
        st[n++] = (double)(u->missile_wait == 0);
        st[n++] = (double)(u->has_missile || u->cannon_burst > 0);
    } else {
        st[n++] = (double)(u->cannon_burst > 0);
    }
    n += opp_ac_values(w, a, 0, opp_id, id, opp_dist, st + n);
    n += friendly_ac_values(w, a, id, fri_id, st + n);
    return n;
}

/* env_base.py:137-164 esc_state_values */
static int esc_state_values(const o_world *w, const o_arena *a, int id, int n_opps, const int *opp_ids, const double *opp_d,
                            int fri_id, double *st) {
    const o_ac *u = &a->ac[id - 1];
    int n = 0;
    double x, y;
    rel_pos(w, u, &x, &y);
    st[n++] = x;
    st[n++] = y;
    st[n++] = hh_clip(hh_div_known(u->spd, HH_AC_MAX_SPEED(u->ac_type), HH_AC_INV_MAX_SPEED(u->ac_type)), 0.0, 1.0);
    st[n++] = hh_clip(HH_DIVC(hh_pymod(u->hdg, 359.0), 359.0), 0.0, 1.0);
    st[n++] = hh_clip((double)u->cannon_remain / (double)u->cannon_max, 0.0, 1.0);
    if (u->ac_type == 1) st[n++] = hh_clip((double)u->missile_remain / (double)u->rocket_max, 0.0, 1.0);
    st[n++] = (double)shot_flag(u);
    int m = 0;
    double os[18];
    for (int k = 0; k < 18; k++) os[k] = 0.0;
    for (int k = 0; k < n_opps && m < 18; k++) m += opp_ac_values(w, a, 1, opp_ids[k], id, opp_d[k], os + m);
    for (int k = 0; k < 18; k++) st[n++] = os[k];
    n += friendly_ac_values(w, a, id, fri_id, st + n);
    return n;
}

static int fri_ac_id(const o_world *w, int id) { /* env_hetero.py:71-75 */
    if (id <= w->cfg.n_agents) return id == 2 ? 1 : 2;
    return id == 4 ? 3 : 4;
}

static void obs_store(o_arena *a, int slot, const double *st, int n) {
    for (int k = 0; k < OBS_MAX; k++) a->obs[slot][k] = k < n ? (float)st[k] : 0.0f;
}

/* env_hetero.py:65-103 lowlevel_state for one unit id (also refreshes opp_to_attack[id]) */
static void lowlevel_state_one(const o_world *w, o_arena *a, int id, int mode, double *st, int *n_out) {
    int ids[MAXA];
    double dn[MAXA], dr[MAXA];
    *n_out = 0;
    a->tgt_n[id - 1] = 0;
    a->tgt_id[id - 1][0] = 0;
    if (a->ac[id - 1].alive) {
        int n = nearby(w, a, id, 0, ids, dn, dr);
        if (n > 0) {
            if (mode == HH_MODE_FIGHT)
                *n_out = fight_state_values(w, a, id, ids[0], dn[0], fri_ac_id(w, id), st);
            else
                *n_out = esc_state_values(w, a, id, n, ids, dn, fri_ac_id(w, id), st);
            a->tgt_n[id - 1] = 1;
            a->tgt_id[id - 1][0] = ids[0];
            a->tgt_d[id - 1][0] = dn[0];
        }
    }
}

/* env_hetero.py:62-63 state() */
static void ll_state(const o_world *w, o_arena *a) {
    double st[OBS_MAX];
    for (int id = 1; id <= w->cfg.n_agents; id++) {
        int n;
        lowlevel_state_one(w, a, id, w->cfg.agent_mode, st, &n);
        obs_store(a, id - 1, st, n);
    }
}

/* ------------------------------------------------------------------ simulator */
/* ac1.py:72-79 fire_missile + 144-146 _angle_in_radar_range */
static void fire_missile(const o_world *w, o_arena *a, int id, int opp_id) {
    o_ac *u = &a->ac[id - 1];
    (void)w;
    if (!u->has_missile && u->missile_remain > 0) {
        double km, brg;
        const o_ac *o = &a->ac[opp_id - 1];
        dist_bearing(u->lat, u->lon, o->lat, o->lon, &km, &brg);
        if (km <= HH_MISSILE_RANGE_KM) {
            double delta = hh_fabs(signed_heading_diff(normalize_angle(u->hdg + HH_MISSILE_HALF_DEG), brg));
            if ((int)delta <= (int)HH_MISSILE_HALF_DEG) {
                o_rk *r = &a->rk[id - 1];
                r->alive = 1;
                r->lat = u->lat;
                r->lon = u->lon;
                r->hdg = u->hdg;
                r->cmd_hdg = u->hdg;
                r->target = opp_id;
                r->life = 0;
                r->seq = ++a->next_seq;
                u->has_missile = 1;
                u->missile_remain = u->missile_remain - 1 > 0 ? u->missile_remain - 1 : 0;
                a->ev_mask |= HH_EV_BIT(w->slots, 3, id - 1, is_agent(w, id));
            }
        }
    }
}

/* ac1.py:69-70 / ac2.py:65-66 */
static void fire_cannon(o_ac *u) {
    int b = HH_AC_BURST(u->ac_type);
    u->cannon_burst = u->cannon_remain < b ? u->cannon_remain : b;
}

/* ac1.py:81-133 / ac2.py:68-107 update() of aircraft `id`; appends events */
static void aircraft_update(const o_world *w, o_arena *a, int id, o_event *ev, int *nev) {
    o_ac *u = &a->ac[id - 1];
    int t = u->ac_type;
    /* heading, ac1.py:83-90 */
    if (u->hdg != u->cmd_hdg) {
        double delta = signed_heading_diff(u->hdg, u->cmd_hdg);
        double max_deg = HH_AC_TURN_RATE(t) * 1.0;
        if (hh_fabs(delta) <= max_deg) {
            u->hdg = u->cmd_hdg;
        } else {
            u->hdg += delta >= 0.0 ? max_deg : -max_deg;
            u->hdg = hh_pymod(u->hdg, 360.0);
        }
    }
    /* speed, ac1.py:93-99 */
    if (u->spd != u->cmd_spd) {
        double delta = u->cmd_spd - u->spd;
        double max_delta = HH_AC_ACCEL(t) * 1.0;
        if (hh_fabs(delta) <= max_delta)
            u->spd = u->cmd_spd;
        else
            u->spd += delta >= 0.0 ? max_delta : -max_delta;
    }
    /* cannon, ac1.py:101-115 */
    if (u->cannon_burst > 0) {
        u->cannon_burst = u->cannon_burst - 1 > 0 ? u->cannon_burst - 1 : 0;
        u->cannon_remain = u->cannon_remain - 1 > 0 ? u->cannon_remain - 1 : 0;
        for (int j = 1; j <= w->A; j++) { /* list(sim.active_units.values()): currently alive, id order */
            if (j == id || !a->ac[j - 1].alive) continue;
            int enemy = is_agent(w, id) ? (j >= w->cfg.n_agents + 1) : (j <= w->cfg.n_agents);
            if (!(w->cfg.friendly_kill || enemy)) continue;
            /* _unit_in_cannon_range, ac1.py:135-142 */
            double km, brg;
            dist_bearing(u->lat, u->lon, a->ac[j - 1].lat, a->ac[j - 1].lon, &km, &brg);
            int in_range = 0;
            if (km < HH_AC_CANNON_KM(t)) {
                double d = hh_fabs(signed_heading_diff(u->hdg, brg));
                in_range = d <= HH_AC_CANNON_HALF(t);
            }
            if (in_range) {
                if (rng_u(a, id, HH_SITE_CANNON, j) < HH_AC_HIT_PROB(t)) {
                    a->ac[j - 1].alive = 0;
                    ev[*nev].origin_rocket = 0;
                    ev[*nev].killer = id;
                    ev[*nev].destroyed = j;
                    (*nev)++;
                    a->ev_mask |= HH_EV_BIT(w->slots, 0, j - 1, 0);
                }
            }
        }
    }
    /* missile bookkeeping, ac1.py:117-128 (type 2 never has actual_missile) */
    if (u->has_missile) {
        o_rk *r = &a->rk[id - 1];
        if (!r->alive) {
            u->has_missile = 0;
        } else {
            double h = r->hdg * hh_rng_uniform(rng_u(a, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05);
            r->cmd_hdg = hh_clip(h, 0.0, 359.0);
        }
    }
    /* Unit.update, cmano_simulator.py:65-72 */
    if (u->spd > 0.0) hh_geo_move(u->lat, u->lon, u->hdg, u->spd * HH_KNOTS_TO_MS * 1.0, &u->lat, &u->lon);
}

/* rocket_unit.py:37-73 */
static void rocket_update(const o_world *w, o_arena *a, int slot, o_event *ev, int *nev) {
    static const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
    o_rk *r = &a->rk[slot];
    int source = slot + 1;
    double km, brg;
    o_ac *tg = &a->ac[r->target - 1];
    dist_bearing(r->lat, r->lon, tg->lat, tg->lon, &km, &brg);
    if (km < HH_ROCKET_FUSE_KM && tg->alive) {
        r->alive = 0;
        tg->alive = 0;
        ev[*nev].origin_rocket = 1; ev[*nev].killer = source; ev[*nev].destroyed = r->target; (*nev)++;
        a->ev_mask |= HH_EV_BIT(w->slots, 1, r->target - 1, 0);
        return;
    }
    if (w->cfg.friendly_kill) {
        int fid = source == 2 ? 1 : 2; /* rocket_unit.py:46 */
        o_ac *f = &a->ac[fid - 1];
        if (f->alive) {
            dist_bearing(r->lat, r->lon, f->lat, f->lon, &km, &brg);
            if (km < HH_ROCKET_FUSE_KM) {
                r->alive = 0;
                f->alive = 0;
                ev[*nev].origin_rocket = 1; ev[*nev].killer = source; ev[*nev].destroyed = fid; (*nev)++;
                a->ev_mask |= HH_EV_BIT(w->slots, 1, fid - 1, 0);
                return;
            }
        }
    }
    if (r->life > HH_ROCKET_MAX_LIFE) { r->alive = 0; return; }
    if (r->hdg != r->cmd_hdg) {
        double delta = signed_heading_diff(r->hdg, r->cmd_hdg);
        if (hh_fabs(delta) <= HH_ROCKET_TURN_RATE) r->hdg = r->cmd_hdg;
        else r->hdg += delta >= 0.0 ? HH_ROCKET_TURN_RATE : -HH_ROCKET_TURN_RATE;
    }
    double spd = speed_table[r->life];
    if (spd > 0.0) hh_geo_move(r->lat, r->lon, r->hdg, spd * HH_KNOTS_TO_MS * 1.0, &r->lat, &r->lon);
    r->life++; /* sim.utc_time advances after the tick (cmano_simulator.py:146) */
}

/* cmano_simulator.py:138-157 do_tick */
static int do_tick(const o_world *w, o_arena *a, o_event *ev) {
    int nev = 0;
    int ac_snap[MAXA], rk_order[MAXA], nrk = 0;
    for (int i = 0; i < w->A; i++) ac_snap[i] = a->ac[i].alive;
    for (int s = 0; s < w->A; s++) if (a->rk[s].alive) rk_order[nrk++] = s;
    for (int i = 1; i < nrk; i++) { /* launch order = unit id order */
        int t = rk_order[i], k = i - 1;
        while (k >= 0 && a->rk[rk_order[k]].seq > a->rk[t].seq) { rk_order[k + 1] = rk_order[k]; k--; }
        rk_order[k + 1] = t;
    }
    for (int i = 1; i <= w->A; i++) if (ac_snap[i - 1]) aircraft_update(w, a, i, ev, &nev);
    for (int k = 0; k < nrk; k++) rocket_update(w, a, rk_order[k], ev, &nev);
    return nev;
}

/* ------------------------------------------------------------------ actions */
/* env_base.py:214-238 _take_base_action; returns 0 or error */
static void take_base_action(const o_world *w, o_arena *a, int hl, int id, int opp_id, const int8_t *act_in) {
    o_ac *u = &a->ac[id - 1];
    /* an action outside MultiDiscrete([13,9,2,2]) (the reference would raise at ac1.py:62-66 for the speed component): run on the
     * sanitised word and remember it per arena (hh_spec.h: hh_action_sanitize, hh_abi.h: hh_action_faults) */
    int bad = 0;
    const uint32_t aw = hh_action_sanitize((uint32_t)(uint8_t)act_in[0] | ((uint32_t)(uint8_t)act_in[1] << 8) | ((uint32_t)(uint8_t)act_in[2] << 16) |
                                           ((uint32_t)(uint8_t)act_in[3] << 24), &bad);
    const int8_t act[4] = {(int8_t)(aw & 0xff), (int8_t)((aw >> 8) & 0xff), (int8_t)((aw >> 16) & 0xff), (int8_t)((aw >> 24) & 0xff)};
    a->act_fault |= bad;
    double nh = hh_pymod(u->hdg + (double)((act[0] - 6) * 15), 360.0);
    if (nh >= 360.0 || nh < 0.0) nh = 0.0; /* ac1.py:59-60 would raise (unreachable) */
    u->cmd_hdg = nh;
    double mx = HH_AC_MAX_SPEED(u->ac_type);
    u->cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
    int agent_ll = !hl && id <= w->cfg.n_agents;
    if (act[2] && u->cannon_remain > 0) {
        fire_cannon(u);
        if (agent_ll && w->cfg.agent_mode == HH_MODE_ESCAPE && u->cannon_remain < 90) a->reward[id - 1] -= 0.1;
    }
    if (u->ac_type == 1 && act[3]) {
        if (opp_id && u->missile_remain > 0 && !u->has_missile && u->missile_wait == 0) {
            fire_missile(w, a, id, opp_id);
            double uu = rng_u(a, id, HH_SITE_MISSILE_WAIT, 0);
            u->missile_wait = hl ? hh_rng_randint(uu, 8, 12) : hh_rng_randint(uu, 7, 17);
            if (agent_ll && w->cfg.agent_mode == HH_MODE_ESCAPE && u->missile_remain < 3) a->reward[id - 1] -= 0.1;
        }
    }
    if (u->missile_wait > 0 && !u->has_missile) u->missile_wait -= 1;
}

/* env_base.py:464-487 _correct_angle_sign */
static double correct_angle_sign(const o_ac *opp, const o_ac *ag) {
    double x = opp->lon, y = opp->lat, h = opp->hdg;
    double s, c;
    hh_sincos(hh_pymod(h, 360.0) * (HH_PI / 180.0), &s, &c);
    double x1 = x + hh_round3(s), y1 = y + hh_round3(c);
    double xc = ag->lon, yc = ag->lat;
    double val = (x1 - x) * (yc - y) - (xc - x) * (y1 - y);
    return val < 0.0 ? 1.0 : -1.0;
}

static void set_speed_checked(o_ac *u, double s) { u->cmd_spd = s; }

/* env_hetero.py:118-123 */
static void opp_level1(const o_world *w, o_arena *a, int id) {
    o_ac *u = &a->ac[id - 1];
    if (!u->has_missile && (a->steps % 40) < 3 && hh_rng_randint(rng_u(a, id, HH_SITE_L12_COIN, 0), 0, 1) &&
        u->missile_wait == 0 && u->ac_type == 1) {
        int ids[MAXA]; double dn[MAXA], dr[MAXA];
        if (nearby(w, a, id, 0, ids, dn, dr) > 0) {
            fire_missile(w, a, id, ids[0]);
            u->missile_wait = 5;
        }
    }
}

/* env_hetero.py:125-136 */
static void opp_level2(const o_world *w, o_arena *a, int id) {
    o_ac *u = &a->ac[id - 1];
    fire_cannon(u);
    int man = a->steps <= 5;
    if (!man) man = (a->steps % hh_rng_randint(rng_u(a, id, HH_SITE_L2_PERIOD, 0), 35, 45)) <= 5;
    if (man) {
        int r = hh_rng_randint(rng_u(a, id, HH_SITE_L2_TURN, 0), 0, 1);
        u->cmd_hdg = hh_pymod(u->hdg + (r ? -90.0 : 90.0), 360.0);
        u->cmd_spd = (double)(100 + hh_rng_randint(rng_u(a, id, HH_SITE_L2_SPEED, 0), 0, 4) * 75);
    }
    if (!u->has_missile && (a->steps % 40) < 3 && hh_rng_randint(rng_u(a, id, HH_SITE_L12_COIN, 0), 0, 1) &&
        u->missile_wait == 0 && u->ac_type == 1) {
        int ids[MAXA]; double dn[MAXA], dr[MAXA];
        if (nearby(w, a, id, 0, ids, dn, dr) > 0) {
            fire_missile(w, a, id, ids[0]);
            u->missile_wait = 5;
        }
    }
}

/* env_hetero.py:138-158 (+ _escaping_opp 227-245, _hardcoded_opp 247-271) */
static void opp_level3(const o_world *w, o_arena *a, int id) {
    o_ac *u = &a->ac[id - 1];
    if (a->steps % 60 == 0 && !a->escaping) {
        a->escaping = hh_rng_randint(rng_u(a, id, HH_SITE_L3_ESC_COIN, 0), 0, 1);
        if (a->escaping) a->escaping_time = (int)hh_rng_uniform(rng_u(a, id, HH_SITE_L3_ESC_TIME, 0), 20.0, 30.0);
    }
    int opp = 0, fire = 0, fire_m = 0;
    double heading, speed;
    if (a->escaping) {
        double y, x;
        rel_pos(w, u, &y, &x);
        double uh = rng_u(a, id, HH_SITE_ESC_HDG, 0);
        if (y < 0.5) heading = x < 0.5 ? (double)(int)hh_rng_uniform(uh, 30.0, 60.0) : (double)(int)hh_rng_uniform(uh, 300.0, 330.0);
        else heading = x < 0.5 ? (double)(int)hh_rng_uniform(uh, 120.0, 150.0) : (double)(int)hh_rng_uniform(uh, 210.0, 240.0);
        speed = (double)(int)hh_rng_uniform(rng_u(a, id, HH_SITE_ESC_SPEED, 0), 300.0, 600.0);
        fire = hh_rng_randint(rng_u(a, id, HH_SITE_ESC_FIRE, 0), 0, 1);
        a->escaping_time -= 1;
        if (a->escaping_time <= 0) a->escaping = 0;
    } else {
        int ids[MAXA]; double dn[MAXA], dr[MAXA];
        int n = nearby(w, a, id, 0, ids, dn, dr);
        heading = u->hdg;
        speed = (double)(int)hh_rng_uniform(rng_u(a, id, HH_SITE_HC_SPEED1, 0), 100.0, 400.0);
        if (n > 0) {
            const o_ac *ag = &a->ac[ids[0] - 1];
            double sign = correct_angle_sign(u, ag);
            double r = hh_rng_uniform(rng_u(a, id, HH_SITE_HC_R, 0), 0.7, 1.3);
            double focus = focus_deg(u, ag);
            if (dn[0] > 0.008 && focus > 4.0) heading = hh_pymod(heading + r * sign * focus, 360.0);
            if (dn[0] > 0.05) {
                double us = rng_u(a, id, HH_SITE_HC_SPEED2, 0);
                speed = focus < 30.0 ? (double)(int)hh_rng_uniform(us, 500.0, 800.0) : (double)(int)hh_rng_uniform(us, 100.0, 500.0);
            }
            fire = dn[0] < 0.03 && focus < 10.0;
            fire_m = dn[0] < 0.09 && focus < 5.0;
            opp = ids[0];
        }
        if (u->ac_type == 2) speed = hh_clip(speed, 0.0, 600.0);
    }
    if (heading >= 360.0 || heading < 0.0) heading = 0.0; /* set_heading would raise (Q20) */
    u->cmd_hdg = heading;
    set_speed_checked(u, speed);
    if (fire) fire_cannon(u);
    if (fire_m && opp && !u->has_missile && u->missile_wait == 0 && u->ac_type == 1) {
        fire_missile(w, a, id, opp);
        u->missile_wait = 10;
    }
}

/* ------------------------------------------------------------------ rewards */
/* env_base.py:240-310 _combat_rewards; rews[] are running sums in append order */
static int combat_rewards(const o_world *w, o_arena *a, int hl, const o_event *ev, int nev, const double *opp_stat0,
                          double *rews, int *destroyed) {
    double s = w->cfg.rew_scale;
    int kill_event = 0;
    int nA = w->cfg.n_agents;
    for (int i = 0; i < nA; i++) { rews[i] = 0.0; destroyed[i] = 0; }
    for (int i = 1; i <= w->A; i++) {
        o_ac *u = &a->ac[i - 1];
        if (u->alive && !in_boundary(w, u)) {
            u->alive = 0;
            kill_event = 1;
            a->ev_mask |= HH_EV_BIT(w->slots, 2, i - 1, 0);
            if (i <= nA) {
                rews[i - 1] += (hl ? -2.0 : -5.0) * s;
                destroyed[i - 1] = 1;
            }
        }
    }
    for (int e = 0; e < nev; e++) {
        int k = ev[e].killer, d = ev[e].destroyed;
        if (k <= nA) {
            if (d > nA) {
                if (!hl) {
                    if (w->cfg.agent_mode == HH_MODE_FIGHT) {
                        const o_ac *ku = &a->ac[k - 1];
                        if (ev[e].origin_rocket) {
                            rews[k - 1] += (1.0 + ((1.5 - 1.0) / (1.0 - 0.0)) * ((double)ku->missile_remain / (double)ku->rocket_max - 0.0)) * s;
                        } else {
                            double r1 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * ((double)ku->cannon_remain / (double)ku->cannon_max - 0.0);
                            double r2 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * (opp_stat0[k - 1] - 0.0);
                            rews[k - 1] += (r1 + r2) * s;
                        }
                    }
                } else {
                    rews[k - 1] += 1.0;
                }
            } else {
                if (!hl) {
                    rews[k - 1] += -2.0 * s;
                    if (w->cfg.friendly_punish) {
                        rews[d - 1] += -2.0 * s;
                        destroyed[d - 1] = 1;
                    }
                }
            }
        } else {
            if (d <= nA) {
                rews[d - 1] += (hl ? -1.0 : -2.0) * s;
                destroyed[d - 1] = 1;
            }
        }
        kill_event = 1;
    }
    return kill_event;
}

static void count_alive(const o_world *w, const o_arena *a, int *ag, int *op) {
    *ag = *op = 0;
    for (int i = 1; i <= w->A; i++)
        if (a->ac[i - 1].alive) { if (i <= w->cfg.n_agents) (*ag)++; else (*op)++; }
}

/* ------------------------------------------------------------------ reset */
/* env_base.py:489-549 _sample_state (low level) / env_hier.py:226-250 (high level) */
static void sample_state(const o_world *w, o_arena *a, int agent, int i, int r, double *x, double *y, int *hd) {
    int id = agent ? i + 1 : w->cfg.n_agents + i + 1;
    double ux = rng_u(a, id, HH_SITE_RESET_X, 0), uy = rng_u(a, id, HH_SITE_RESET_Y, 0);
    double uh = rng_u(a, id, HH_SITE_RESET_HDG, 0);
    int near_side = agent ? (r == 1) : (r == 2); /* which x-band this group spawns in */
    *hd = 0;
    if (w->cfg.env_kind == HH_ENV_HIGHLEVEL) {
        double n = agent ? (double)w->cfg.n_agents : (double)w->cfg.n_opps;
        *x = near_side ? hh_rng_uniform(ux, 7.07, 7.22) : hh_rng_uniform(ux, 7.28, 7.43);
        *y = hh_rng_uniform(uy, 5.07 + i * (0.4 / n), 5.12 + i * (0.4 / n));
        *hd = hh_rng_randint(uh, 0, 359);
        return;
    }
    int lvl = w->cfg.level;
    if (lvl == 1) {
        *x = near_side ? hh_rng_uniform(ux, 7.12, 7.14) : hh_rng_uniform(ux, 7.16, 7.17);
        *y = hh_rng_uniform(uy, 5.1 + i * 0.1, 5.11 + i * 0.1);
        if (agent) *hd = r == 1 ? hh_rng_randint(uh, 30, 150) : hh_rng_randint(uh, 200, 330);
    } else if (lvl == 2) {
        *x = near_side ? hh_rng_uniform(ux, 7.08, 7.13) : hh_rng_uniform(ux, 7.18, 7.23);
        *y = hh_rng_uniform(uy, 5.08 + i * 0.1, 5.13 + i * 0.1);
        if (agent) *hd = r == 1 ? hh_rng_randint(uh, 0, 180) : hh_rng_randint(uh, 180, 359);
        else *hd = hh_rng_randint(uh, 0, 359);
    } else {
        *x = near_side ? hh_rng_uniform(ux, 7.07, 7.12) : hh_rng_uniform(ux, 7.18, 7.23);
        *y = hh_rng_uniform(uy, 5.09 + i * 0.1, 5.12 + i * 0.1);
        if (agent) *hd = r == 1 ? hh_rng_randint(uh, 0, 270) : hh_rng_randint(uh, 90, 359);
        else *hd = hh_rng_randint(uh, 0, 359);
    }
}

static void hl_state(const o_world *w, o_arena *a);

/* env_base.py:62-77 reset + 551-585 _reset_scenario */
static void arena_reset(const o_world *w, o_arena *a) {
    a->episode += 1;
    a->steps = 0;
    a->escaping = 0;
    a->escaping_time = 0;
    a->next_seq = 0;
    a->done = 0;
    a->ep_ret = 0.0;
    a->ev_mask = 0;
    int hl = w->cfg.env_kind == HH_ENV_HIGHLEVEL;
    int r = hh_rng_randint(rng_u(a, 0, HH_SITE_RESET_SIDE, 0), 1, 2);
    for (int g = 0; g < 2; g++) {
        int agent = g == 0;
        int count = agent ? w->cfg.n_agents : w->cfg.n_opps;
        for (int i = 0; i < count; i++) {
            int id = agent ? i + 1 : w->cfg.n_agents + i + 1;
            double x, y;
            int hd;
            sample_state(w, a, agent, i, r, &x, &y, &hd);
            int ac = i <= 1 ? i + 1 : hh_rng_randint(rng_u(a, id, HH_SITE_RESET_TYPE, 0), 1, 2);
            o_ac *u = &a->ac[id - 1];
            memset(u, 0, sizeof(*u));
            u->lat = y;
            u->lon = x;
            u->hdg = (double)hd;
            u->spd = (w->cfg.level <= 2 && !agent) ? 0.0 : 100.0;
            u->cmd_hdg = u->hdg;
            u->cmd_spd = u->spd;
            u->alive = 1;
            u->ac_type = ac;
            u->cannon_remain = u->cannon_max = HH_AC_CANNON_DEFAULT;
            u->missile_remain = u->rocket_max = ac == 1 ? HH_AC1_MISSILES_DEFAULT : 0;
            if (!hl) {
                if (w->cfg.level <= 4 && !agent) {
                    u->cannon_remain = u->cannon_max = 400;
                    if (ac == 1) u->missile_remain = u->rocket_max = 8;
                } else if (w->cfg.level == 5) {
                    u->cannon_remain = u->cannon_max = 300;
                    if (ac == 1) u->missile_remain = u->rocket_max = 6;
                }
            } else {
                u->cannon_remain = u->cannon_max = 300;
                if (ac == 1) u->missile_remain = u->rocket_max = 8;
            }
            memset(&a->rk[id - 1], 0, sizeof(o_rk));
            a->tgt_n[id - 1] = 0;
            for (int k = 0; k < O_TGT_K; k++) { a->tgt_id[id - 1][k] = 0; a->tgt_d[id - 1][k] = 0.0; }
        }
    }
    for (int i = 0; i < MAXA; i++) { a->reward[i] = 0.0; a->reward_valid[i] = 0; }
    if (hl) hl_state(w, a); else ll_state(w, a);
}

/* ------------------------------------------------------------------ LowLevelEnv.step */
static void finish_episode(o_arena *a, int ag, int op, int horizon) {
    a->last_ret = (float)a->ep_ret;
    a->last_len = a->steps;
    a->last_outcome = (op <= 0 && a->steps < horizon) ? 1 : ((ag <= 0 && a->steps < horizon) ? -1 : 0);
}

/* env_hetero.py:160-172: the units flown by actions (agents; opponents at levels 4-5) in id order.
 * first = 1 / last = n_agents for the agents' half, n_agents+1 / A for the frozen-policy opponents. */
static void ll_act_range(const o_world *w, o_arena *a, int first, int last, const int8_t *actions /* rows by unit id */) {
    int nA = w->cfg.n_agents;
    for (int i = first; i <= last; i++) {
        if (!a->ac[i - 1].alive) continue;
        if (i <= nA) {
            a->reward_valid[i - 1] = 1;
            int t = a->tgt_n[i - 1] ? a->tgt_id[i - 1][0] : 0;
            if (t && a->ac[t - 1].alive) a->opp_stat0[i - 1] = focus_norm(&a->ac[t - 1], &a->ac[i - 1]);
        }
        int t = a->tgt_n[i - 1] ? a->tgt_id[i - 1][0] : 0;
        take_base_action(w, a, 0, i, t, actions + 4 * (i - 1));
    }
}

/* env_base.py:349-398 _policy_actions -> lowlevel_state(opp_mode, i): observation of a frozen-policy
 * opponent (also refreshes its target).  Evaluated after the agents acted (id order, env_hetero.py:160-172). */
static void ll_opp_obs(const o_world *w, o_arena *a, int opp_mode, float *obs /* [n_opps, 30] */) {
    double st[OBS_MAX];
    for (int i = w->cfg.n_agents + 1; i <= w->A; i++) {
        int n = 0;
        if (a->ac[i - 1].alive) lowlevel_state_one(w, a, i, opp_mode, st, &n); /* dead units are skipped (env_hetero.py:161) */
        if (obs) for (int k = 0; k < 30; k++) obs[(i - w->cfg.n_agents - 1) * 30 + k] = k < n ? (float)st[k] : 0.0f;
    }
}

static void ll_begin(o_arena *a) {
    a->ev_mask = 0;
    for (int i = 0; i < MAXA; i++) { a->reward[i] = 0.0; a->reward_valid[i] = 0; a->opp_stat0[i] = 0.0; }
    a->steps += 1;
}

/* do_tick + rewards + done + state (env_hetero.py:184-225, env_base.py:89-90) */
static void ll_tick_and_rewards(const o_world *w, o_arena *a) {
    int nA = w->cfg.n_agents;
    o_event ev[4 * MAXA];
    int nev = do_tick(w, a, ev);
    double rews[MAXA];
    int destroyed[MAXA];
    combat_rewards(w, a, 0, ev, nev, a->opp_stat0, rews, destroyed);
    /* env_hetero.py:198-214 per-step escape shaping */
    if (w->cfg.agent_mode == HH_MODE_ESCAPE && w->cfg.esc_dist_rew) {
        for (int i = 1; i <= nA; i++) {
            if (!a->ac[i - 1].alive) continue;
            int ids[MAXA]; double dn[MAXA], dr[MAXA];
            int n = nearby(w, a, i, 0, ids, dn, dr);
            for (int j = 1; j <= n; j++) {
                if (dr[j - 1] < 0.06) {
                    rews[i - 1] += -0.02 / j;
                    if (a->ac[i - 1].spd < 200.0) rews[i - 1] += -0.02 / j;
                } else if (dr[j - 1] > 0.13) {
                    rews[i - 1] += 0.02 / j;
                    if (a->ac[i - 1].spd > 500.0) rews[i - 1] += 0.02 / j;
                }
            }
        }
    }
    /* env_hetero.py:217-223 */
    for (int i = 1; i <= nA; i++) {
        if (a->ac[i - 1].alive || destroyed[i - 1]) {
            if (w->cfg.glob_frac > 0.0 && w->cfg.agent_mode == HH_MODE_FIGHT)
                a->reward[i - 1] += rews[i - 1] + w->cfg.glob_frac * rews[i % 2];
            else
                a->reward[i - 1] += rews[i - 1];
        }
    }
    int ag, op;
    count_alive(w, a, &ag, &op);
    a->done = ag <= 0 || op <= 0 || a->steps >= w->cfg.horizon;
    for (int i = 0; i < nA; i++) if (a->reward_valid[i]) a->ep_ret += a->reward[i];
    if (a->done) finish_episode(a, ag, op, w->cfg.horizon);
    ll_state(w, a);
}

/* env_base.py:79-109 step -> env_hetero.py:105-186 _take_action -> 188-225 _get_rewards */
static void ll_step(const o_world *w, o_arena *a, const int8_t *actions /* [n_ctrl,4] */) {
    int nA = w->cfg.n_agents;
    ll_begin(a);
    ll_act_range(w, a, 1, nA, actions);
    if (w->cfg.ext_opp_actions) {
        ll_opp_obs(w, a, HH_MODE_FIGHT, 0);
        ll_act_range(w, a, nA + 1, w->A, actions);
    } else {
        for (int i = nA + 1; i <= w->A; i++) {
            if (!a->ac[i - 1].alive) continue;
            if (w->cfg.level == 1) opp_level1(w, a, i);
            else if (w->cfg.level == 2) opp_level2(w, a, i);
            else opp_level3(w, a, i);
        }
    }
    ll_tick_and_rewards(w, a);
}

/* ------------------------------------------------------------------ HighLevelEnv (env_hier.py) */
/* env_hier.py:49-98 state(): commander observation + sorted target lists for every unit */
static void hl_state(const o_world *w, o_arena *a) {
    int nA = w->cfg.n_agents;
    for (int id = 1; id <= w->A; id++) {
        int ids[MAXA]; double dn[MAXA], dr[MAXA];
        a->tgt_n[id - 1] = 0;
        for (int k = 0; k < O_TGT_K; k++) { a->tgt_id[id - 1][k] = 0; a->tgt_d[id - 1][k] = 0.0; }
        if (id <= nA) {
            double st[OBS_MAX];
            int n = 0;
            if (a->ac[id - 1].alive) {
                int no = nearby(w, a, id, 0, ids, dn, dr);
                if (no > 0) {
                    const o_ac *u = &a->ac[id - 1];
                    double x, y;
                    rel_pos(w, u, &x, &y);
                    st[n++] = x;
                    st[n++] = y;
                    st[n++] = hh_clip(hh_div_known(u->spd, HH_AC_MAX_SPEED(u->ac_type), HH_AC_INV_MAX_SPEED(u->ac_type)), 0.0, 1.0);
                    st[n++] = hh_clip(HH_DIVC(hh_pymod(u->hdg, 359.0), 359.0), 0.0, 1.0);
                    double os[20]; int m = 0;
                    for (int k = 0; k < 20; k++) os[k] = 0.0;
                    for (int k = 0; k < no; k++) {
                        m += opp_ac_values(w, a, 2, ids[k], id, dn[k], os + m);
                        a->tgt_id[id - 1][a->tgt_n[id - 1]] = ids[k];
                        a->tgt_d[id - 1][a->tgt_n[id - 1]] = dn[k];
                        a->tgt_n[id - 1]++;
                        if (m == HH_N_OPP_HL * 10) break;
                    }
                    for (int k = 0; k < 20; k++) st[n++] = os[k];
                    double fs[10]; m = 0;
                    for (int k = 0; k < 10; k++) fs[k] = 0.0;
                    int fids[MAXA]; double fdn[MAXA], fdr[MAXA];
                    int nf = nearby(w, a, id, 1, fids, fdn, fdr);
                    for (int k = 0; k < nf; k++) {
                        m += friendly_ac_values(w, a, id, fids[k], fs + m);
                        if (m == 10) break;
                    }
                    for (int k = 0; k < 10; k++) st[n++] = fs[k];
                }
            }
            obs_store(a, id - 1, st, n);
        } else if (a->ac[id - 1].alive) {
            int no = nearby(w, a, id, 0, ids, dn, dr);
            for (int k = 0; k < no && k < O_TGT_K; k++) { a->tgt_id[id - 1][k] = ids[k]; a->tgt_d[id - 1][k] = dn[k]; }
            a->tgt_n[id - 1] = no < O_TGT_K ? no : O_TGT_K;
        }
    }
}


/* ------------------------------------------------------------------ batch API */
#define API __attribute__((visibility("default")))

API int hho_create(const hh_config *cfg, void **out) {
    if (!cfg || !out || cfg->n_arenas <= 0) return HH_E_ARG;
    int A = cfg->n_agents + cfg->n_opps;
    if (A > MAXA || cfg->n_agents < 1 || cfg->n_opps < 1) return HH_E_ARG;
    o_world *w = (o_world *)calloc(1, sizeof(o_world));
    w->cfg = *cfg;
    w->A = A;
    if (cfg->env_kind == HH_ENV_HIGHLEVEL && (cfg->n_agents > HH_SIDE_MAX || cfg->n_opps > HH_SIDE_MAX)) { free(w); return HH_E_ARG; }
    w->tgt_k = cfg->env_kind == HH_ENV_HIGHLEVEL ? HH_TGT_K_OF(cfg->n_agents, cfg->n_opps) : HH_TGT_K;
    w->slots = cfg->env_kind == HH_ENV_HIGHLEVEL ? HH_HL_SLOTS(cfg->n_agents, cfg->n_opps) : A;
    w->n_ctrl = cfg->ext_opp_actions ? A : cfg->n_agents;
    if (cfg->env_kind == HH_ENV_HIGHLEVEL) w->D = HH_OBS_HL;
    else w->D = cfg->agent_mode == HH_MODE_FIGHT ? HH_OBS_FIGHT_AC1 : HH_OBS_ESC_AC1;
    double m = cfg->map_size;
    w->lat_hi = HH_MAP_LAT0 + m;
    w->lon_hi = HH_MAP_LON0 + m;
    w->ext_lat = w->lat_hi - HH_MAP_LAT0; /* map_limits.py:19-23 */
    w->ext_lon = w->lon_hi - HH_MAP_LON0;
    w->inv_ext_lat = 1.0 / w->ext_lat;
    w->inv_ext_lon = 1.0 / w->ext_lon;
    w->inv_diag = (1.0 - 0.0) / (hh_sqrt(2.0 * (m * m)) - 0.0); /* env_base.py:439,458-462 */
    w->ar = (o_arena *)calloc((size_t)cfg->n_arenas, sizeof(o_arena));
    for (int n = 0; n < cfg->n_arenas; n++) {
        w->ar[n].akey = hh_rng_arena_key(cfg->seed, cfg->arena_offset + (uint64_t)n);
        w->ar[n].last_outcome = 2;
        w->ar[n].done = 1; /* must be reset before stepping */
    }
    *out = w;
    return HH_OK;
}

API int hho_destroy(void *h) {
    o_world *w = (o_world *)h;
    if (!w) return HH_E_ARG;
    free(w->ar);
    free(w);
    return HH_OK;
}

API int hho_obs_dim(void *h) { return ((o_world *)h)->D; }
API int hho_n_ctrl(void *h) { return ((o_world *)h)->n_ctrl; }

static void copy_obs(const o_world *w, const o_arena *a, float *obs) {
    for (int i = 0; i < w->cfg.n_agents; i++)
        for (int k = 0; k < w->D; k++) obs[i * w->D + k] = a->obs[i][k];
}

API int hho_reset(void *h, const uint8_t *mask, float *obs) {
    o_world *w = (o_world *)h;
    int N = w->cfg.n_arenas;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        if (mask && !mask[n]) continue;
        arena_reset(w, &w->ar[n]);
        if (obs) copy_obs(w, &w->ar[n], obs + (size_t)n * w->cfg.n_agents * w->D);
    }
    return HH_OK;
}

/* one env.step() of arena n (+ the caller-side reset when auto_reset); output rows are the arena's own */
static void step_arena(const o_world *w, int n, const int8_t *act /* [n_ctrl,4] */, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done) {
    const int nA = w->cfg.n_agents;
    o_arena *a = &w->ar[n];
    if (a->done) {
        for (int i = 0; i < nA; i++) { a->reward[i] = 0.0; a->reward_valid[i] = 0; }
        a->ev_mask = 0;
    } else {
        ll_step(w, a, act);
    }
    if (reward) for (int i = 0; i < nA; i++) reward[i] = (float)a->reward[i];
    if (reward_valid) for (int i = 0; i < nA; i++) reward_valid[i] = (uint8_t)a->reward_valid[i];
    if (done) *done = (uint8_t)a->done;
    if (a->done && w->cfg.auto_reset) {
        uint32_t em = a->ev_mask; /* masks describe the step that just ended */
        arena_reset(w, a);
        a->ev_mask = em;
    }
    if (obs) copy_obs(w, a, obs);
}

API int hho_step(void *h, const int8_t *actions, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done) {
    o_world *w = (o_world *)h;
    int N = w->cfg.n_arenas, nA = w->cfg.n_agents;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) return HH_E_ARG;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++)
        step_arena(w, n, actions + (size_t)n * w->n_ctrl * 4, obs ? obs + (size_t)n * nA * w->D : 0, reward ? reward + (size_t)n * nA : 0,
                   reward_valid ? reward_valid + (size_t)n * nA : 0, done ? done + n : 0);
    return HH_OK;
}

/* levels 4-5: the step split around the frozen opponent policy (env_hetero.py:160-172):
 *   hho_step_begin(agent actions) -> observations of the opponents (after the agents acted)
 *   hho_step_finish(opponent actions) -> like hho_step */
API int hho_step_begin(void *h, const int8_t *agent_actions /* [N, n_agents, 4] */, int opp_mode, float *opp_obs /* [N, n_opps, 30] */) {
    o_world *w = (o_world *)h;
    int N = w->cfg.n_arenas, nA = w->cfg.n_agents, nO = w->cfg.n_opps;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL || !w->cfg.ext_opp_actions) return HH_E_ARG;
    for (int n = 0; n < N; n++) {
        o_arena *a = &w->ar[n];
        float *oo = opp_obs ? opp_obs + (size_t)n * nO * 30 : 0;
        if (a->done) {
            for (int i = 0; i < nA; i++) { a->reward[i] = 0.0; a->reward_valid[i] = 0; }
            a->ev_mask = 0;
            if (oo) for (int k = 0; k < nO * 30; k++) oo[k] = 0.0f;
            continue;
        }
        int8_t act[MAXA * 4] = {0};
        for (int i = 0; i < nA * 4; i++) act[i] = agent_actions[(size_t)n * nA * 4 + i];
        ll_begin(a);
        ll_act_range(w, a, 1, nA, act);
        /* env_hetero.py:55-59: level 5 draws the opponents' policy set (and with it their observation mode) per episode */
        int mode = opp_mode >= 0 ? opp_mode : (hh_l5_policy_pick(a->akey, (uint32_t)a->episode) == 5 ? HH_MODE_ESCAPE : HH_MODE_FIGHT);
        ll_opp_obs(w, a, mode, oo);
    }
    return HH_OK;
}

/* env_hetero.py:55-59: k = randint(3,5) of every arena's current episode (level 5, fight mode; 0 otherwise) */
API int hho_opp_policy(void *h, int8_t *k_out) {
    o_world *w = (o_world *)h;
    int l5 = w->cfg.env_kind == HH_ENV_LOWLEVEL && w->cfg.level == 5 && w->cfg.agent_mode == HH_MODE_FIGHT;
    for (int n = 0; n < w->cfg.n_arenas; n++) k_out[n] = l5 ? (int8_t)hh_l5_policy_pick(w->ar[n].akey, (uint32_t)w->ar[n].episode) : 0;
    return HH_OK;
}

API int hho_step_finish(void *h, const int8_t *opp_actions /* [N, n_opps, 4] */, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done) {
    o_world *w = (o_world *)h;
    int N = w->cfg.n_arenas, nA = w->cfg.n_agents, nO = w->cfg.n_opps;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL || !w->cfg.ext_opp_actions) return HH_E_ARG;
    for (int n = 0; n < N; n++) {
        o_arena *a = &w->ar[n];
        if (!a->done) {
            int8_t act[MAXA * 4] = {0};
            for (int i = 0; i < nO * 4; i++) act[nA * 4 + i] = opp_actions[(size_t)n * nO * 4 + i];
            ll_act_range(w, a, nA + 1, w->A, act);
            ll_tick_and_rewards(w, a);
        }
        if (reward) for (int i = 0; i < nA; i++) reward[(size_t)n * nA + i] = (float)a->reward[i];
        if (reward_valid) for (int i = 0; i < nA; i++) reward_valid[(size_t)n * nA + i] = (uint8_t)a->reward_valid[i];
        if (done) done[n] = (uint8_t)a->done;
        if (a->done && w->cfg.auto_reset) {
            uint32_t em = a->ev_mask;
            arena_reset(w, a);
            a->ev_mask = em;
        }
        if (obs) copy_obs(w, a, obs + (size_t)n * nA * w->D);
    }
    return HH_OK;
}

/* T steps of every arena.  Arenas are independent, so the loop nest is arena-outer / tick-inner: one OpenMP team for the
 * whole rollout, each thread walks its arenas through all T ticks with the arena's state hot in its cache (this is also
 * the shape timed as bench.py's cpu_baseline).  Same results as T calls of hho_step. */
API int hho_rollout(void *h, int n_steps, const int8_t *actions, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done) {
    o_world *w = (o_world *)h;
    const size_t N = (size_t)w->cfg.n_arenas, nA = (size_t)w->cfg.n_agents, D = (size_t)w->D, nc = (size_t)w->n_ctrl;
    if (w->cfg.env_kind != HH_ENV_LOWLEVEL) return HH_E_ARG;
    /* schedule(static): thread k owns one contiguous block of arenas for the whole call and on every call, so the rows it writes
     * (contiguous in n inside each tick's slab) are first touched by it and stay on its NUMA node, and neighbouring threads share at
     * most one cache line per slab (dynamic chunks of 4 arenas shared two lines per 832 B) */
#pragma omp parallel for schedule(static)
    for (int n = 0; n < (int)N; n++)
        for (int t = 0; t < n_steps; t++) {
            const size_t r = (size_t)t * N + (size_t)n;
            step_arena(w, n, actions + r * nc * 4, obs ? obs + r * nA * D : 0, reward ? reward + r * nA : 0,
                       reward_valid ? reward_valid + r * nA : 0, done ? done + r : 0);
        }
    return HH_OK;
}

/* threads an OpenMP team of this library gets (what bench.py's cpu_baseline reports as `threads`) / set it (one-thread leg) */
API int hho_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
API void hho_omp_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
/* threads that actually ran the body of a parallel region of the size hho_rollout opens (counted, not assumed) */
API int hho_omp_team_size(void) {
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp single
        n = omp_get_num_threads();
    }
#endif
    return n;
}

API int hho_episode_stats(void *h, float *ret, int32_t *len, int8_t *outcome) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        if (ret) ret[n] = w->ar[n].last_ret;
        if (len) len[n] = w->ar[n].last_len;
        if (outcome) outcome[n] = (int8_t)w->ar[n].last_outcome;
    }
    return HH_OK;
}

/* hh_abi.h: hh_action_faults */
API int hho_action_faults(void *h, uint8_t *out, int clear) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        if (out) out[n] = (uint8_t)(w->ar[n].act_fault != 0);
        if (clear) w->ar[n].act_fault = 0;
    }
    return HH_OK;
}

API int hho_get_event_masks(void *h, uint32_t *masks) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) masks[n] = w->ar[n].ev_mask;
    return HH_OK;
}

API int hho_get_state(void *h, hh_state_view *v) {
    o_world *w = (o_world *)h;
    int A = w->A;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        const o_arena *a = &w->ar[n];
        int ag, op;
        count_alive(w, a, &ag, &op);
        for (int s = 0; s < A; s++) {
            const o_ac *u = &a->ac[s];
            const o_rk *r = &a->rk[s];
            size_t b = (size_t)n * A + s;
            double *f = v->ac_f + b * HH_ACF_K;
            f[0] = u->lat; f[1] = u->lon; f[2] = u->hdg; f[3] = u->spd; f[4] = u->cmd_hdg; f[5] = u->cmd_spd;
            int32_t *q = v->ac_i + b * HH_ACI_K;
            q[0] = u->alive; q[1] = u->ac_type; q[2] = u->cannon_remain; q[3] = u->cannon_burst; q[4] = u->cannon_max;
            q[5] = u->missile_remain; q[6] = u->rocket_max; q[7] = u->missile_wait; q[8] = u->has_missile;
            q[9] = a->tgt_n[s] ? a->tgt_id[s][0] : 0;
            double *g = v->rk_f + b * HH_RKF_K;
            int32_t *p = v->rk_i + b * HH_RKI_K;
            if (r->alive) { /* dead rocket slots read as zeros */
                g[0] = r->lat; g[1] = r->lon; g[2] = r->hdg; g[3] = r->cmd_hdg;
                p[0] = 1; p[1] = r->target; p[2] = r->life; p[3] = r->seq;
            } else {
                g[0] = g[1] = g[2] = g[3] = 0.0;
                p[0] = p[1] = p[2] = p[3] = 0;
            }
            for (int k = 0; k < w->tgt_k; k++) {
                v->tgt_id[b * w->tgt_k + k] = k < a->tgt_n[s] ? a->tgt_id[s][k] : 0;
                v->tgt_d[b * w->tgt_k + k] = k < a->tgt_n[s] ? a->tgt_d[s][k] : 0.0;
            }
        }
        int32_t *ai = v->ar_i + (size_t)n * HH_ARI_K;
        ai[0] = a->steps; ai[1] = ag; ai[2] = op; ai[3] = a->escaping; ai[4] = a->escaping_time; ai[5] = a->episode;
    }
    return HH_OK;
}

API int hho_set_state(void *h, const hh_state_view *v) {
    o_world *w = (o_world *)h;
    int A = w->A;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        a->next_seq = 0;
        for (int s = 0; s < A; s++) {
            o_ac *u = &a->ac[s];
            o_rk *r = &a->rk[s];
            size_t b = (size_t)n * A + s;
            const double *f = v->ac_f + b * HH_ACF_K;
            u->lat = f[0]; u->lon = f[1]; u->hdg = f[2]; u->spd = f[3]; u->cmd_hdg = f[4]; u->cmd_spd = f[5];
            const int32_t *q = v->ac_i + b * HH_ACI_K;
            u->alive = q[0]; u->ac_type = q[1]; u->cannon_remain = q[2]; u->cannon_burst = q[3]; u->cannon_max = q[4];
            u->missile_remain = q[5]; u->rocket_max = q[6]; u->missile_wait = q[7]; u->has_missile = q[8];
            const double *g = v->rk_f + b * HH_RKF_K;
            r->lat = g[0]; r->lon = g[1]; r->hdg = g[2]; r->cmd_hdg = g[3];
            const int32_t *p = v->rk_i + b * HH_RKI_K;
            r->alive = p[0]; r->target = p[1]; r->life = p[2]; r->seq = p[3];
            if (r->seq > a->next_seq) a->next_seq = r->seq;
            a->tgt_n[s] = 0;
            for (int k = 0; k < w->tgt_k; k++) {
                a->tgt_id[s][k] = v->tgt_id[b * w->tgt_k + k];
                a->tgt_d[s][k] = v->tgt_d[b * w->tgt_k + k];
                if (a->tgt_id[s][k]) a->tgt_n[s] = k + 1;
            }
        }
        const int32_t *ai = v->ar_i + (size_t)n * HH_ARI_K;
        a->steps = ai[0]; a->escaping = ai[3]; a->escaping_time = ai[4]; a->episode = ai[5];
        int ag, op;
        count_alive(w, a, &ag, &op);
        a->done = ag <= 0 || op <= 0 || a->steps >= w->cfg.horizon;
        if (w->cfg.env_kind == HH_ENV_HIGHLEVEL) hl_state(w, a); else ll_state(w, a);
    }
    return HH_OK;
}

/* current observation of every arena (after reset / set_state) */
API int hho_get_obs(void *h, float *obs) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) copy_obs(w, &w->ar[n], obs + (size_t)n * w->cfg.n_agents * w->D);
    return HH_OK;
}

/* ------------------------------------------------------------------ HighLevelEnv macro step */
static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* env_hier.py:142-190 _action_assess */
static void hl_action_assess(const o_world *w, o_arena *a, const int8_t *cmd /* [n_agents] */) {
    int nA = w->cfg.n_agents;
    for (int i = 1; i <= w->A; i++) {
        const o_ac *u = &a->ac[i - 1];
        if (u->alive) {
            if (i <= nA) {
                a->reward[i - 1] = 0.0;
                int c = cmd[i - 1];
                if (c > 0) {
                    int opp_id = 0;
                    if (c - 1 < a->tgt_n[i - 1]) opp_id = a->tgt_id[i - 1][c - 1];
                    else c = 1; /* try/except IndexError -> opp_id None, action forced to 1 */
                    if (!opp_id) a->reward[i - 1] = -0.1;
                    if (w->cfg.hier_action_assess && opp_id) {
                        const o_ac *o = &a->ac[opp_id - 1];
                        if (dist_raw(u, o) < 0.1 && focus_deg(u, o) < 15.0 && focus_deg(o, u) > 40.0) a->reward[i - 1] = 0.1;
                        else a->reward[i - 1] = 0.0;
                    }
                } else if (w->cfg.hier_action_assess) {
                    const o_ac *o = &a->ac[a->tgt_id[i - 1][0] - 1];
                    if (dist_raw(o, u) < 0.1 && focus_deg(o, u) < 15.0 && focus_deg(u, o) > 40.0) a->reward[i - 1] = 0.1;
                }
                a->cmd_act[i - 1] = c;
            } else {
                /* Fraction(ratio, 100).limit_denominator().as_integer_ratio() -> weights [den-num, num] */
                int g = gcd_i(w->cfg.hier_opp_fight_ratio, 100);
                int num = w->cfg.hier_opp_fight_ratio / (g ? g : 1), den = 100 / (g ? g : 1);
                double total = (double)den + 0.0;
                int fight = rng_u(a, i, HH_SITE_HL_FIGHT, 0) * total >= (double)(den - num); /* bisect(cum, u*total, 0, 1) */
                int ag_id;
                if (fight) {
                    int possible = a->tgt_n[i - 1];
                    if (possible > 1 && (rng_u(a, i, HH_SITE_HL_OTHER, 0) * 4.0 >= 1.0))
                        ag_id = hh_rng_randint(rng_u(a, i, HH_SITE_HL_PICK, 0), 2, possible);
                    else
                        ag_id = 1;
                } else {
                    ag_id = 0;
                }
                a->cmd_act[i - 1] = ag_id;
            }
        } else {
            if (i <= nA) a->reward[i - 1] = 0.0;
            a->cmd_act[i - 1] = 0;
        }
    }
    for (int i = 0; i < nA; i++) a->reward_valid[i] = 1; /* every agent id gets a key (env_hier.py:154,188) */
}

/* index into the stored target list the way Python does: commander_actions[i]-1, -1 = last */
static int hl_target_index(const o_arena *a, int id) {
    int c = a->cmd_act[id - 1];
    return c > 0 ? c - 1 : a->tgt_n[id - 1] - 1;
}

/* env_hier.py:100-112 lowlevel_state for one unit; returns policy mode 1 fight / 2 escape */
static int hl_pilot_obs_one(const o_world *w, const o_arena *a, int id, float *out /* [30] */) {
    double st[OBS_MAX];
    int ids[MAXA]; double dn[MAXA], dr[MAXA];
    int nf = nearby(w, a, id, 1, ids, dn, dr);
    int fri = nf ? ids[0] : 0;
    int n, mode;
    if (a->cmd_act[id - 1] != 0) {
        int k = a->cmd_act[id - 1] - 1;
        n = fight_state_values(w, a, id, a->tgt_id[id - 1][k], a->tgt_d[id - 1][k], fri, st);
        mode = 1;
    } else {
        n = esc_state_values(w, a, id, a->tgt_n[id - 1], a->tgt_id[id - 1], a->tgt_d[id - 1], fri, st);
        mode = 2;
    }
    for (int k = 0; k < 30; k++) out[k] = k < n ? (float)st[k] : 0.0f;
    return mode;
}

/* env_hier.py:192-208 _surrounding_event */
static int hl_surrounding_event(const o_world *w, const o_arena *a) {
    for (int i = 1; i <= w->cfg.n_agents; i++)
        for (int j = w->cfg.n_agents + 1; j <= w->A; j++)
            if (a->ac[i - 1].alive && a->ac[j - 1].alive) {
                const o_ac *x = &a->ac[i - 1], *y = &a->ac[j - 1];
                if (dist_raw(x, y) < 0.1 && (focus_deg(x, y) < 15.0 || focus_deg(y, x) < 15.0)) return 1;
            }
    return 0;
}

/* one sub-step of env_hier.py:125-138 for one arena */
static void hl_substep(const o_world *w, o_arena *a, const int8_t *actions /* [A,4] */) {
    int nA = w->cfg.n_agents;
    o_event ev[4 * MAXA];
    for (int i = nA + 1; i <= w->A; i++) { /* agents already acted in hl_side_act (same id order as env_hier.py:126-130) */
        if (!a->ac[i - 1].alive) continue;
        int k = hl_target_index(a, i);
        int tgt = k >= 0 ? a->tgt_id[i - 1][k] : 0;
        take_base_action(w, a, 1, i, tgt, actions + 4 * (i - 1));
    }
    int nev = do_tick(w, a, ev);
    double rews[MAXA];
    int destroyed[MAXA];
    double dummy[MAXA] = {0};
    int kill_event = combat_rewards(w, a, 1, ev, nev, dummy, rews, destroyed);
    /* env_hier.py:210-224 _get_rewards */
    for (int i = 1; i <= nA; i++) {
        if (a->ac[i - 1].alive || destroyed[i - 1]) {
            if (w->cfg.glob_frac > 0.0) {
                double other = 0.0;
                for (int j = 1; j <= nA; j++) if (j != i) other += rews[j - 1];
                a->reward[i - 1] += rews[i - 1] + w->cfg.glob_frac * other;
            } else {
                a->reward[i - 1] += rews[i - 1];
            }
        }
    }
    a->hl_kill = kill_event;
    if (a->hl_s > 10) a->hl_situ = hl_surrounding_event(w, a); /* s > self.min_sub_steps (10) */
    a->hl_s += 1;
    a->steps += 1;
    a->hl_running = a->hl_s <= 15 && !a->hl_kill && !a->hl_situ; /* while s <= n_sub_steps(15) and not ... */
}

API int hho_hl_begin(void *h, const int8_t *cmd /* [N, n_agents] */) {
    o_world *w = (o_world *)h;
    if (w->cfg.env_kind != HH_ENV_HIGHLEVEL) return HH_E_ARG;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        for (int i = 0; i < MAXA; i++) { a->reward[i] = 0.0; a->reward_valid[i] = 0; }
        a->hl_s = 0; a->hl_kill = 0; a->hl_situ = 0;
        a->hl_running = !a->done;
        if (a->hl_running) hl_action_assess(w, a, cmd + (size_t)n * w->cfg.n_agents);
    }
    return HH_OK;
}

/* The reference walks the units in id order and lets each one observe and act before the next one
 * observes (env_hier.py:126-130), so the opponents' observations already contain the agents'
 * same-sub-step cannon / missile flags.  Batched restatement with identical results:
 *   side 0: observations of the agents -> (pilot inference) -> hho_hl_agents_act
 *   side 1: observations of the opponents -> (pilot inference) -> hho_hl_tick
 * (within a side nobody observes a same-side unit's weapon flags, so the order inside a side is free). */
API int hho_hl_agents_act(void *h, const int8_t *actions /* [N, A, 4], agent rows used */) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        if (!a->hl_running) continue;
        a->ev_mask = 0;
        for (int i = 1; i <= w->cfg.n_agents; i++) {
            if (!a->ac[i - 1].alive) continue;
            int k = hl_target_index(a, i);
            int tgt = k >= 0 ? a->tgt_id[i - 1][k] : 0;
            take_base_action(w, a, 1, i, tgt, actions + ((size_t)n * w->A + (i - 1)) * 4);
        }
    }
    return HH_OK;
}

API int hho_hl_pilot_obs(void *h, int side, float *obs /* [N, A, 30] */, uint8_t *mode /* [N, A] */) {
    o_world *w = (o_world *)h;
    int A = w->A;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        for (int i = 1; i <= A; i++) {
            float *o = obs + ((size_t)n * A + (i - 1)) * 30;
            int md = 0;
            int mine = side == 0 ? i <= w->cfg.n_agents : i > w->cfg.n_agents;
            if (a->hl_running && a->ac[i - 1].alive && mine) {
                md = hl_pilot_obs_one(w, a, i, o);
                if (w->cfg.opp_side_selector && i > w->cfg.n_agents && md == 1) md |= HH_SEL_OPP_SIDE; /* "fight_*_opp", env_base.py:387-390 */
                md |= a->ac[i - 1].ac_type << 2; /* policy type | aircraft type */
            }
            else for (int k = 0; k < 30; k++) o[k] = 0.0f;
            mode[(size_t)n * A + (i - 1)] = (uint8_t)md;
        }
    }
    return HH_OK;
}

API int hho_hl_tick(void *h, const int8_t *actions /* [N, A, 4] */) {
    o_world *w = (o_world *)h;
    int running = 0;
#pragma omp parallel for schedule(static) reduction(+ : running)
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        if (a->hl_running) hl_substep(w, a, actions + (size_t)n * w->A * 4);
        running += a->hl_running;
    }
    return running;
}

API int hho_hl_end(void *h, float *obs /* [N, n_agents, 34] */, float *reward, uint8_t *reward_valid, uint8_t *done) {
    o_world *w = (o_world *)h;
    int nA = w->cfg.n_agents;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        o_arena *a = &w->ar[n];
        for (int k = 0; k < HH_EVAL_K; k++) a->eval_last[k] = 0;
        if (!a->done) {
            int ag, op;
            count_alive(w, a, &ag, &op);
            a->done = ag <= 0 || op <= 0 || a->steps >= w->cfg.horizon;
            {   /* env_base.py:91-107 eval_info: `for k, v in action.items(): if self.sim.unit_exists(k)` after the step */
                int *e = a->eval_last, h = w->cfg.horizon;
                e[0] = op <= 0 && a->steps < h; e[1] = ag <= 0 && a->steps < h; e[2] = a->steps >= h && ag > 0 && op > 0;
                for (int k = 1; k <= w->A; k++) {
                    if (!a->ac[k - 1].alive) continue;
                    int v = a->cmd_act[k - 1];
                    if (v) { if (k <= nA) { e[3]++; e[7]++; e[8 + v]++; } else { e[5]++; e[8]++; } }
                    else { if (k <= nA) { e[4]++; e[7]++; } else { e[6]++; e[8]++; } }
                }
                for (int k = 0; k < HH_EVAL_K; k++) a->eval_tot[k] += e[k];
            }
            for (int i = 0; i < nA; i++) if (a->reward_valid[i]) a->ep_ret += a->reward[i];
            if (a->done) finish_episode(a, ag, op, w->cfg.horizon);
            hl_state(w, a);
        }
        a->hl_running = 0;
        if (reward) for (int i = 0; i < nA; i++) reward[(size_t)n * nA + i] = (float)a->reward[i];
        if (reward_valid) for (int i = 0; i < nA; i++) reward_valid[(size_t)n * nA + i] = (uint8_t)a->reward_valid[i];
        if (done) done[n] = (uint8_t)a->done;
        if (a->done && w->cfg.auto_reset) {
            uint32_t em = a->ev_mask;
            arena_reset(w, a);
            a->ev_mask = em;
        }
        if (obs) copy_obs(w, a, obs + (size_t)n * nA * w->D);
    }
    return HH_OK;
}

API int hho_eval_info(void *h, int32_t *last /* [N, HH_EVAL_K] */, int32_t *total) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++)
        for (int k = 0; k < HH_EVAL_K; k++) {
            if (last) last[(size_t)n * HH_EVAL_K + k] = w->ar[n].eval_last[k];
            if (total) total[(size_t)n * HH_EVAL_K + k] = w->ar[n].eval_tot[k];
        }
    return HH_OK;
}

API int hho_hl_get_cmd(void *h, int32_t *cmd /* [N, A] */, int32_t *sub /* [N] */) {
    o_world *w = (o_world *)h;
    for (int n = 0; n < w->cfg.n_arenas; n++) {
        for (int i = 0; i < w->A; i++) cmd[(size_t)n * w->A + i] = w->ar[n].cmd_act[i];
        if (sub) sub[n] = w->ar[n].hl_s;
    }
    return HH_OK;
}


/* probes for tests/test_math.py and tests/test_geodesic.py */
API void hho_math_eval(int fn, int n, const double *a, const double *b, double *o0, double *o1) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
            case 0: hh_sincos(a[i], &o0[i], &o1[i]); break;
            case 1: o0[i] = hh_atan2(a[i], b[i]); break;
            case 2: o0[i] = hh_acos(a[i]); break;
            case 3: hh_sincosd(a[i], &o0[i], &o1[i]); break;
            case 4: o0[i] = hh_atan2d(a[i], b[i]); break;
            case 5: o0[i] = hh_pymod(a[i], b[i]); break;
            case 6: o0[i] = hh_remainder(a[i], b[i]); break;
            case 7: o0[i] = hh_fmod(a[i], b[i]); break;
            case 8: o0[i] = hh_round3(a[i]); break;
            case 9: o0[i] = hh_div_known(a[i], b[i], 1.0 / b[i]); break;
            case 10: o0[i] = hh_pymod_turn(a[i], b[i]); break;
            case 11: o0[i] = hh_sqrt(a[i]); break;
            case 12: o0[i] = hh_clip(a[i], 0.0, 1.0); break;
            case 13: o0[i] = hh_clip(a[i], -b[i], b[i]); break;
            default: break;
        }
    }
}
API void hho_geo_direct(int n, const double *lat, const double *lon, const double *azi, const double *s, double *lat2, double *lon2) {
    for (int i = 0; i < n; i++) hh_geo_direct(lat[i], lon[i], azi[i], s[i], &lat2[i], &lon2[i]);
}
API void hho_geo_move(int n, const double *lat, const double *lon, const double *azi, const double *s, double *lat2, double *lon2) {
    for (int i = 0; i < n; i++) hh_geo_move(lat[i], lon[i], azi[i], s[i], &lat2[i], &lon2[i]);
}
API void hho_geo_inverse_estimate(int n, const double *lat1, const double *lon1, const double *lat2, const double *lon2, double *s12, double *azi1) {
    for (int i = 0; i < n; i++) hh_geo_inverse_estimate(lat1[i], lon1[i], lat2[i], lon2[i], &s12[i], &azi1[i]);
}
API void hho_geo_inverse(int n, const double *lat1, const double *lon1, const double *lat2, const double *lon2, double *s12, double *azi1) {
    for (int i = 0; i < n; i++) hh_geo_inverse(lat1[i], lon1[i], lat2[i], lon2[i], &s12[i], &azi1[i]);
}
/* probe for tests: the planar stage of the missile-launch predicate computed the way the kernels do (heading
 * unit vector, planar focus angle, cross product; env_base.py:424-439) next to the exact predicate (ac1.py:135-146).
 * pre: 1 / 0 / -1 (undecided); exact: 0 / 1; beta_p, beta: planar and geodesic relative bearing [deg] */
API void hho_missile_cone_planar(int n, const double *lat1, const double *lon1, const double *hdg, const double *lat2,
                                 const double *lon2, int32_t *pre, int32_t *exact, double *beta_p, double *beta) {
    for (int i = 0; i < n; i++) {
        double sn, cs;
        hh_sincos(hh_pymod(90.0 - hdg[i], 360.0) * (HH_PI / 180.0), &sn, &cs);
        double n1 = hh_sqrt(cs * cs + sn * sn);
        double dx = lon2[i] - lon1[i], dy = lat2[i] - lat1[i];
        double n2 = hh_sqrt(dx * dx + dy * dy);
        double dot = cs * dx + sn * dy;
        double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
        double foc = hh_acos(x) * (180.0 / HH_PI);
        double cross = cs * dy - sn * dx;
        pre[i] = hh_missile_cone_planar(lat1[i], lon1[i], lat2[i], lon2[i], foc, cross, n2);
        beta_p[i] = cross < 0.0 ? foc : -foc;
        double s12, azi;
        hh_geo_inverse(lat1[i], lon1[i], lat2[i], lon2[i], &s12, &azi);
        double brg = normalize_angle(azi);
        beta[i] = signed_heading_diff(hdg[i], brg);
        int in = 0;
        if (s12 / 1000.0 <= HH_MISSILE_RANGE_KM) {
            double delta = hh_fabs(signed_heading_diff(normalize_angle(hdg[i] + HH_MISSILE_HALF_DEG), brg));
            in = (int)delta <= (int)HH_MISSILE_HALF_DEG;
        }
        exact[i] = in;
    }
}
/* probe for tests: planar "certainly outside the cannon cone" stage next to the exact predicate (ac1.py:106-115,135-141) */
API void hho_cannon_cone_planar(int n, int ac_type, const double *lat1, const double *lon1, const double *hdg, const double *lat2,
                                const double *lon2, int32_t *outside, int32_t *exact) {
    for (int i = 0; i < n; i++) {
        double sn, cs;
        hh_sincos(hh_pymod(90.0 - hdg[i], 360.0) * (HH_PI / 180.0), &sn, &cs);
        outside[i] = hh_cannon_cone_planar_outside(lat1[i], lon1[i], lat2[i], lon2[i], cs, sn, ac_type);
        double km, brg;
        dist_bearing(lat1[i], lon1[i], lat2[i], lon2[i], &km, &brg);
        exact[i] = km < HH_AC_CANNON_KM(ac_type) && hh_fabs(signed_heading_diff(hdg[i], brg)) <= HH_AC_CANNON_HALF(ac_type);
    }
}
/* hh_abi.h: hh_action_tape_uniform (the benchmark's keyed synthetic actions), out [T, N, n_units, 4] */
API int hho_action_tape_uniform(uint64_t seed, uint64_t arena_offset, int step0, int T, int N, int n_units, int8_t *out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)T * N * n_units; i++) {
        const long per_t = (long)N * n_units, t = i / per_t, r = i - t * per_t, n = r / n_units, s = r - n * n_units;
        const uint32_t w = hh_rng_action_word(seed, arena_offset + (uint64_t)n, (uint32_t)(step0 + (int)t), (uint32_t)(s + 1));
        out[4 * i] = (int8_t)(w & 0xff); out[4 * i + 1] = (int8_t)((w >> 8) & 0xff); out[4 * i + 2] = (int8_t)((w >> 16) & 0xff); out[4 * i + 3] = (int8_t)((w >> 24) & 0xff);
    }
    return HH_OK;
}

API double hho_rng_u01(uint64_t seed, uint64_t arena, uint32_t episode, uint32_t tick, uint32_t unit, uint32_t site, uint32_t sub) {
    return hh_rng_u01(hh_rng_tick_key(hh_rng_arena_key(seed, arena), episode, tick), unit, site, sub);
}
