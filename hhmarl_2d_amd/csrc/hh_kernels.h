/*
 * hh_kernels.h — gfx950 kernels of the batched air-combat world.
 *
 * Replaces, for thousands of arenas at once, the reference's per-process Python path
 *   envs/env_base.py:79-109 step  ->  envs/env_hetero.py:105-186 _take_action
 *   -> warsim/simulator/cmano_simulator.py:138-157 do_tick (ac1.py:81-133, ac2.py:68-107,
 *      rocket_unit.py:37-73)  ->  env_hetero.py:188-225 rewards  ->  env_hetero.py:65-103 state
 * and envs/env_base.py:62-77,551-585 reset.
 *
 * This file is the LDS-exchange implementation (any arena size; RESET / OBSERVE / split steps / 3-vs-3, and the A/B
 * reference of the 2-vs-2 rollout whose production form is the register-exchange kernel of hh_kernels_quad.h).
 * One persistent kernel, hh_k_world<A,B> (one lane per aircraft slot, see hh_device.h), keeps
 * the state of its arenas in registers for T ticks; per tick it runs
 *   K1 "step"    fused action decode + scripted opponents + turn/thrust kinematics + WGS84 move,
 *                then the weapon envelopes: every geodesic range/bearing test that survives an
 *                exactness-preserving prefilter is pushed on a workgroup-wide LDS queue and the
 *                queue is drained densely by all lanes with ONE inlined Karney Inverse
 *                (work compaction: ~1 solve per arena-tick instead of 6 divergent ones per lane);
 *                id-ordered kill resolution on integer masks; rewards; done;
 *   pair table   every lane computes distance / focus angle / heading difference from its aircraft
 *                to the other aircraft of its arena ONCE into LDS (the "per-agent-pair" staging);
 *                the observation, target selection, scripted opponents, reward shaping and the
 *                next tick's pre-step statistics are all lookups into it;
 *   K3 "reset"   finished arenas are re-sampled in place when auto_reset is set;
 *   K2 "observe" per-agent observation rows packed into an LDS staging tile and written with
 *                unit-stride (coalesced) stores.
 * T = 1 is hh_step; run = RESET / OBSERVE execute only K3 / pair table / K2.
 * No MFMA: there is no dense contraction on this path; it is FP64 VALU + HBM streaming.
 *
 * Arithmetic is the bit-reproducible include/hh_math.h / hh_geodesic.h set, compiled with
 * -ffp-contract=off, so results are bit-identical to the CPU oracle (tests/test_gpu_parity.py).
 */
#ifndef HH_KERNELS_H
#define HH_KERNELS_H

#include "hh_device.h"

/* phase timers for tuning builds only (-DHH_PROFILE_PHASES): s_memtime deltas per phase, summed per wave
 * into hh_prof_cycles[]; compiled out of the product */
#ifdef HH_PROFILE_PHASES
__device__ unsigned long long hh_prof_cycles[24]; /* 0..11 phases, 12..15 queue statistics, 16..19 output wave (wait X, table, wait Y, rows), 20 / 21 barrier X / Y waits of the simulation wave */
#define HH_PROF_DECL unsigned long long prof_t0_ = __builtin_readcyclecounter(), prof_acc_[14] = {0}
#define HH_PROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); prof_acc_[k] += t_ - prof_t0_; prof_t0_ = t_; } while (0)
#define HH_PROF_ARGS , unsigned long long &prof_t0_, unsigned long long *prof_acc_
#define HH_PROF_PASS , prof_t0_, prof_acc_
#define HH_PROF_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 12; k_++) atomicAdd(&hh_prof_cycles[k_], prof_acc_[k_]); \
                                                          atomicAdd(&hh_prof_cycles[20], prof_acc_[12]); atomicAdd(&hh_prof_cycles[21], prof_acc_[13]); } } while (0)
#else
#define HH_PROF_DECL
#define HH_PROF(k)
#define HH_PROF_ARGS
#define HH_PROF_PASS
#define HH_PROF_FLUSH
#endif

/* ===================================================================== LDS exchange area */
template <int A, int B>
struct Shared {
    static constexpr int GPB = B / A;
    /* published unit state: pre-tick while a tick runs, refreshed after it */
    double lat0[B], lon0[B], hdg[B];
    double uc[B], us[B], un[B]; /* heading unit vector (env_base.py:428) and its norm */
    /* pair tables, [slot j][lane]: from the lane's aircraft towards slot j of its arena */
    double p_dist[A][B]; /* planar distance in degrees (env_base.py:434-439, un-normalised) */
    double p_foc[A][B];  /* focus angle [deg] at the lane's aircraft towards j (env_base.py:424-432) */
    float p_hd[A][B];    /* normalised angle between heading vectors (env_base.py:448-456), symmetric; read only into the
                            float32 observation, so it is kept in that precision (the same value a later cast would give) */
    float nlat[B], nlon[B], nspd[B], nhdg[B]; /* observation entries every observer of this aircraft shares:
                                                 relative position, speed and heading normalised (env_base.py:117-121) */
    double rew[B];
    unsigned long long g_tkey[GPB]; /* keyed-RNG tick key per arena (cannon draws made by worker lanes) */
    double rk_speed[12];            /* rocket_unit.py:25-35 speed profile by age (register-exchange kernel) */
    int flags[B];                   /* bit0 alive, bits1-2 ac_type, bit3 shot flag */
    int aux[B];                     /* per-phase scratch */
    int res[B];                     /* envelope results per requesting lane: bit0 launch ok, bits1-8 cannon hit
                                       on slot j, bit9 rocket fuse on target, bit10 fuse on "friendly" (ten-slot arenas:
                                       cannon bits 1-10, fuse bits 12 / 13: HH_RES_FUSE_BIT) */
    int g_alive[GPB], g_nev[GPB], g_ev[GPB][A > 8 ? A : 8], g_rkdead[GPB];
    union alignas(16) {
        struct {
            double lat1[B], lon1[B], hdg1[B]; /* position / heading after this tick's aircraft update */
            double rk_lat[B], rk_lon[B];      /* rocket position before its move (speculative for a pending launch) */
            int q_code[B * (A > 6 ? 12 : 8)]; /* Inverse work queue: src lane | kind<<8 | slot<<10; per lane at most 1 launch + (A - 1) cannon + 2 fuse tests */
            int q_count;
        } t;
        /* observation staging tile (after the tick): agents' rows; 3-vs-3: every unit's 30-float pilot row.  Kept as small
         * as the arena size allows — at two waves per SIMD eight workgroups share the CU's 160 KB */
        float obs[GPB * (A / 2) * HH_OBS_HL]; /* 3-vs-3: also the acting side's 30-float pilot rows (GPB * 3 * 30) */
    } u;
};

#define FL_ALIVE 1
#define FL_SHOT 8
/* first of the two rocket-fuse bits of an envelope result word, by lanes per arena group */
#define HH_RES_FUSE_BIT(GS) ((GS) > 8 ? 12 : 9)

template <int A, int B>
__device__ __forceinline__ int sh_alive(const Shared<A, B> &sh, int idx) { return sh.flags[idx] & FL_ALIVE; }
template <int A, int B>
__device__ __forceinline__ int sh_type(const Shared<A, B> &sh, int idx) { return (sh.flags[idx] >> 1) & 3; }

/* heading unit vector (east, north) of env_base.py:428 */
__device__ __forceinline__ void heading_vec(double hdg, double &c, double &s) {
    hh_sincos(hh_pymod360(90.0 - hdg) * (HH_PI / 180.0), &s, &c);
}
/* publish the state other lanes read */
template <int A, int B>
__device__ __forceinline__ void publish_hv(Shared<A, B> &sh, int tid, const Unit &m, double c, double s) {
    sh.lat0[tid] = m.lat;
    sh.lon0[tid] = m.lon;
    sh.hdg[tid] = m.hdg;
    sh.uc[tid] = c;
    sh.us[tid] = s;
    sh.un[tid] = hh_sqrt(c * c + s * s);
    int shot = m.burst > 0 || (m.ac_type == 1 && m.has_missile);
    sh.flags[tid] = (m.alive ? FL_ALIVE : 0) | ((m.ac_type & 3) << 1) | (shot ? FL_SHOT : 0);
}

template <int A, int B>
__device__ __forceinline__ void publish(Shared<A, B> &sh, int tid, const Unit &m) {
    double c, s;
    heading_vec(m.hdg, c, s);
    publish_hv(sh, tid, m, c, s);
}

/* publish + the normalised entries of the observation (needs the map extents) */
template <int A, int B>
__device__ __forceinline__ void publish_obs_hv(const DevCfg &c, Shared<A, B> &sh, int tid, const Unit &m, double hc, double hs) {
    publish_hv(sh, tid, m, hc, hs);
    sh.nlat[tid] = (float)hh_clip(hh_div_known(m.lat - HH_MAP_LAT0, c.ext_lat, c.inv_ext_lat), 0.0, 1.0);
    sh.nlon[tid] = (float)hh_clip(hh_div_known(m.lon - HH_MAP_LON0, c.ext_lon, c.inv_ext_lon), 0.0, 1.0);
    sh.nspd[tid] = (float)hh_clip(hh_div_known(m.spd, HH_AC_MAX_SPEED(m.ac_type), HH_AC_INV_MAX_SPEED(m.ac_type)), 0.0, 1.0);
    sh.nhdg[tid] = (float)hh_clip(HH_DIVC(hh_pymod359(m.hdg), 359.0), 0.0, 1.0);
}
template <int A, int B>
__device__ __forceinline__ void publish_obs(const DevCfg &c, Shared<A, B> &sh, int tid, const Unit &m) {
    double hc, hs;
    heading_vec(m.hdg, hc, hs);
    publish_obs_hv(c, sh, tid, m, hc, hs);
}

/* the per-arena pair table (call between two barriers, after publish).  Branch-free and fully unrolled on
 * purpose: the kernel runs at one or two waves per SIMD, so the 2(A-1)+A/2 independent acos chains must
 * overlap inside the lane (ILP); entries involving a dead aircraft are computed but never read. */
template <int A, int B>
__device__ __forceinline__ void pair_tables(Shared<A, B> &sh, int tid, int base, int s, bool active) {
#ifdef HH_ABL_NO_PAIRS
    return;
#endif
    const double c1 = sh.uc[tid], s1 = sh.us[tid], n1 = sh.un[tid];
    const double la = sh.lat0[tid], lo = sh.lon0[tid];
    double dist[A - 1], foc[A - 1], hd[A / 2];
#pragma unroll
    for (int k = 1; k < A; k++) {
        int j = s + k;
        if (j >= A) j -= A;
        double dx = sh.lon0[base + j] - lo, dy = sh.lat0[base + j] - la;
        double n2 = hh_sqrt(dx * dx + dy * dy);
        double dot = c1 * dx + s1 * dy;
        double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
        dist[k - 1] = n2;
        foc[k - 1] = hh_acos(x) * (180.0 / HH_PI);
    }
#pragma unroll
    for (int k = 1; k <= A / 2; k++) {
        int j = s + k;
        if (j >= A) j -= A;
        double c2 = sh.uc[base + j], s2 = sh.us[base + j], n2 = sh.un[base + j];
        double dot = c1 * c2 + s1 * s2;
        double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
        hd[k - 1] = hh_clip(HH_DIVC(hh_acos(x) * (180.0 / HH_PI), 180.0), 0.0, 1.0);
    }
    if (!active) return;
#pragma unroll
    for (int k = 1; k < A; k++) {
        int j = s + k;
        if (j >= A) j -= A;
        sh.p_dist[j][tid] = dist[k - 1];
        sh.p_foc[j][tid] = foc[k - 1];
    }
#pragma unroll
    for (int k = 1; k <= A / 2; k++) {
        int j = s + k;
        if (j >= A) j -= A;
        sh.p_hd[j][tid] = (float)hd[k - 1];
        sh.p_hd[s][base + j] = (float)hd[k - 1];
    }
}

/* env_base.py:400-422 _nearby_object from the pair table: up to 3 live units of the other side
 * (or own side), stable-sorted by normalised distance.  ids are slot indices. */
struct Near3 {
    int n, i0, i1, i2;
    double d0, d1, d2, r0, r1, r2;
};
template <int A, int B>
__device__ __forceinline__ void nearby(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, bool friendly, Near3 &o) {
    o.n = 0; o.i0 = o.i1 = o.i2 = 0; o.d0 = o.d1 = o.d2 = 0.0; o.r0 = o.r1 = o.r2 = 0.0;
    bool me_agent = s < c.nA;
    int lo = (me_agent != friendly) ? c.nA : 0;
    int hi = (me_agent != friendly) ? A : c.nA;
    /* all LDS reads first (independent, one wait), then the tiny insertion sort on registers */
    double dr_[A];
    int al_[A];
#pragma unroll
    for (int j = 0; j < A; j++) { dr_[j] = sh.p_dist[j][tid]; al_[j] = sh.flags[base + j] & FL_ALIVE; }
#pragma unroll
    for (int j = 0; j < A; j++) {
        if (j < lo || j >= hi || j == s || !al_[j]) continue;
        double dr = dr_[j];
        double dn = c.inv_diag * dr;
        int p = (o.n >= 1 && o.d0 <= dn) + (o.n >= 2 && o.d1 <= dn) + (o.n >= 3 && o.d2 <= dn);
        if (p <= 1) { o.i2 = o.i1; o.d2 = o.d1; o.r2 = o.r1; }
        if (p == 0) { o.i1 = o.i0; o.d1 = o.d0; o.r1 = o.r0; o.i0 = j; o.d0 = dn; o.r0 = dr; }
        else if (p == 1) { o.i1 = j; o.d1 = dn; o.r1 = dr; }
        else if (p == 2) { o.i2 = j; o.d2 = dn; o.r2 = dr; }
        if (o.n < 3) o.n++;
    }
}

__device__ __forceinline__ double norm180(double deg) { return hh_clip(HH_DIVC(deg, 180.0), 0.0, 1.0); }          /* focus, norm=True */
__device__ __forceinline__ double aspect(double deg) { return hh_clip(HH_DIVC(180.0 - deg, 180.0), 0.0, 1.0); } /* env_base.py:441-446 */

/* ===================================================================== K3: reset */
/* env_base.py:489-549 / env_hier.py:226-250 _sample_state + env_base.py:551-585 _reset_scenario */
template <int A>
__device__ __forceinline__ void reset_unit(const DevCfg &c, int s, Unit &m, Arena &ar) {
    bool agent = s < c.nA;
    int i = agent ? s : s - c.nA;
    int id = s + 1;
    if (s >= c.nA + c.nO) { /* unused slot of an n-vs-m arena (HighLevelEnv with fewer than six aircraft): never alive */
        m = Unit{};
        m.ac_type = 2; m.cannon_max = 1;
        return;
    }
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
    arena_rekey(ar);
}

/* ===================================================================== K2: observe */
/* env_base.py:185-212 opp_ac_values from the pair table; mode 0 fight / 1 esc / 2 HighLevel */
template <int A, int B>
__device__ __forceinline__ int opp_block(const DevCfg &c, const Shared<A, B> &sh, int mode, int tid, int base, int s, int oj, double dist, float *out) {
    const int o = base + oj;
    const int t = sh_type(sh, o);
    const double f_so = sh.p_foc[oj][tid], f_os = sh.p_foc[s][o];
    int n = 0;
    (void)t;
    out[n++] = sh.nlat[o];
    out[n++] = sh.nlon[o];
    out[n++] = sh.nspd[o];
    out[n++] = sh.nhdg[o];
    out[n++] = sh.p_hd[oj][tid];
    if (mode == 0) {
        out[n++] = (float)norm180(f_os);
        out[n++] = (float)aspect(f_so);
    } else {
        out[n++] = (float)norm180(f_so);
        out[n++] = (float)norm180(f_os);
    }
    if (mode == 2) {
        out[n++] = (float)aspect(f_so);
        out[n++] = (float)aspect(f_os);
    }
    out[n++] = (float)dist;
    if (mode != 2) out[n++] = (sh.flags[o] & FL_SHOT) ? 1.0f : 0.0f;
    return n;
}

/* env_base.py:166-183 friendly_ac_values */
template <int A, int B>
__device__ __forceinline__ void friend_block(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, int fj, float *out) {
    const int f = base + fj;
    if (sh_alive(sh, f)) {
        out[0] = sh.nlat[f];
        out[1] = sh.nlon[f];
        out[2] = (float)norm180(sh.p_foc[fj][tid]);
        out[3] = (float)norm180(sh.p_foc[s][f]);
        out[4] = (float)(c.inv_diag * sh.p_dist[fj][tid]);
    } else {
        out[0] = out[1] = out[2] = out[3] = out[4] = 0.0f;
    }
}

/* env_hetero.py:65-103 lowlevel_state for the lane's own unit (fight / escape), also refreshes
 * opp_to_attack (m.tgt0).  Writes D floats (zero padded) to `out` (LDS staging row). */
template <int A, int B>
__device__ __forceinline__ void lowlevel_obs(const DevCfg &c, const Shared<A, B> &sh, int tid, int base, int s, int mode, Unit &m, float *out, int D) {
    for (int k = 0; k < D; k++) out[k] = 0.0f;
    m.n_tgt = 0; m.tgt0 = 0; m.tgt_d0 = 0.0;
    if (!m.alive) return;
    Near3 nb;
    nearby(c, sh, tid, base, s, false, nb);
    if (nb.n == 0) return;
    m.n_tgt = 1; m.tgt0 = nb.i0 + 1; m.tgt_d0 = nb.d0;
    /* env_hetero.py:71-75 fri_ac_id */
    int fri = s < c.nA ? (s == 1 ? 0 : 1) : (s == 3 ? 2 : 3);
    int n = 0;
    out[n++] = sh.nlat[tid];
    out[n++] = sh.nlon[tid];
    out[n++] = sh.nspd[tid];
    out[n++] = sh.nhdg[tid];
    if (mode == HH_MODE_FIGHT) {
        const int oj = nb.i0;
        out[n++] = (float)norm180(sh.p_foc[oj][tid]);
        out[n++] = (float)aspect(sh.p_foc[s][base + oj]);
        out[n++] = sh.p_hd[oj][tid];
        out[n++] = (float)nb.d0;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) {
            out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
            out[n++] = m.missile_wait == 0 ? 1.0f : 0.0f;
            out[n++] = (m.has_missile || m.burst > 0) ? 1.0f : 0.0f;
        } else {
            out[n++] = m.burst > 0 ? 1.0f : 0.0f;
        }
        n += opp_block(c, sh, 0, tid, base, s, oj, nb.d0, out + n);
    } else {
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
        out[n++] = (sh.flags[tid] & FL_SHOT) ? 1.0f : 0.0f;
        opp_block(c, sh, 1, tid, base, s, nb.i0, nb.d0, out + n);
        if (nb.n >= 2) opp_block(c, sh, 1, tid, base, s, nb.i1, nb.d1, out + n + 9);
        n += 18;
    }
    friend_block(c, sh, tid, base, s, fri, out + n);
}

/* ===================================================================== K1: step */
struct StepOut {
    double reward;
    int valid;
    int kill_event; /* any aircraft removed this tick (env_base.py:248,256,308) */
    double opp_stat0; /* tmode 1 input: env_hetero.py:169-170 statistic taken when the agent acted */
};

/* phase I of the tick: dense pass over the workgroup's envelope queue.  Every entry is decided by the
 * filtered exact predicate (mid-latitude estimate with proven error bounds; the Karney solution for
 * the undecided sliver) and its verdict is OR-ed into the requesting lane's result word. */
/* GS = lanes per arena group (A: aircraft packed; 8: the two-quads-of-three layout of hh_kernels_oct.h, where lane position 0..2
 * = agents, 4..6 = opponents and `opp_id0` = unit id of the first opponent: the cannon draw is keyed by unit ids) */
template <int GS, int B, bool INLINE_EXACT, class SH>
__device__ __forceinline__ void drain_envelope_queue_t(SH &sh, int tid, int count, int opp_id0) {
#ifdef HH_ABL_NO_ENVELOPE
    count = 0;
#endif
#pragma unroll 1
    for (int q = tid; q < count; q += B) {
        int code = sh.u.t.q_code[q];
        int src = code & 0xff, kind = (code >> 8) & 3, j = (code >> 10) & (GS > 8 ? 15 : 7);
        int ss = src % GS, sb = src - ss;
        double la1, lo1, la2, lo2;
        if (kind <= 1) { la1 = sh.lat0[src]; lo1 = sh.lon0[src]; }
        else { la1 = sh.u.t.rk_lat[src]; lo1 = sh.u.t.rk_lon[src]; }
        bool moved = kind >= 2 || (kind == 1 && j < ss);
        la2 = moved ? sh.u.t.lat1[sb + j] : sh.lat0[sb + j];
        lo2 = moved ? sh.u.t.lon1[sb + j] : sh.lon0[sb + j];
        /* filtered exact predicate: decide from the mid-latitude estimate when it is farther from every
         * threshold than its proven error bound; otherwise redo the test with the Karney solution */
        const int t = (sh.flags[src] >> 1) & 3;
        const double hdg_src = kind == 0 ? sh.hdg[src] : sh.u.t.hdg1[src];
        const double sep = hh_max(hh_fabs(la2 - la1), hh_fabs(lo2 - lo1));
        const bool dom = hh_fabs(la1) <= HH_GEO_EST_MAX_LAT && hh_fabs(la2) <= HH_GEO_EST_MAX_LAT &&
                         hh_fabs(lo1) < 170.0 && hh_fabs(lo2) < 170.0;
        int verdict = -1; /* -1 undecided, 0 outside the envelope, 1 inside */
        if (dom && sep <= (kind == 0 ? HH_GEO_EST_LONG_DEG : HH_GEO_EST_SHORT_DEG)) {
            double s_m, az;
            hh_geo_inverse_estimate(la1, lo1, la2, lo2, &s_m, &az);
            if (kind == 0) {
                double delta = hh_fabs(d_signed_heading_diff(d_normalize_angle(hdg_src + HH_MISSILE_HALF_DEG), az));
                if (s_m >= HH_MISSILE_RANGE_KM * 1000.0 + HH_GEO_EST_LONG_ABS_M) verdict = 0;
                else if (s_m <= HH_MISSILE_RANGE_KM * 1000.0 - HH_GEO_EST_LONG_ABS_M && s_m > HH_GEO_EST_MIN_M &&
                         hh_fabs(delta - (HH_MISSILE_HALF_DEG + 1.0)) > HH_GEO_EST_LONG_AZI)
                    verdict = delta < HH_MISSILE_HALF_DEG + 1.0; /* int(delta) <= 60 */
            } else {
                const double r_m = (kind == 1 ? HH_AC_CANNON_KM(t) : HH_ROCKET_FUSE_KM) * 1000.0;
                const double eps = HH_GEO_EST_SHORT_REL * s_m + HH_GEO_EST_SHORT_ABS_M;
                if (s_m >= r_m + eps) verdict = 0;
                else if (s_m < r_m - eps && s_m > HH_GEO_EST_MIN_M) {
                    if (kind >= 2) verdict = 1;
                    else {
                        double d = hh_fabs(d_signed_heading_diff(hdg_src, az));
                        if (d <= HH_AC_CANNON_HALF(t) - HH_GEO_EST_SHORT_AZI) verdict = 1;
                        else if (d > HH_AC_CANNON_HALF(t) + HH_GEO_EST_SHORT_AZI) verdict = 0;
                    }
                }
            }
        }
#ifndef HH_ABL_NO_EXACT
        if (HH_RARE(verdict < 0)) verdict = INLINE_EXACT ? d_envelope_exact_body(kind, t, la1, lo1, la2, lo2, hdg_src)
                                                : d_envelope_exact(kind, t, la1, lo1, la2, lo2, hdg_src);
#endif
        int bit = 0;
        if (verdict) {
            if (kind == 0) bit = 1;
            else if (kind == 1) { /* ac1.py:112-113 Bernoulli hit, drawn only when in the cone */
                const int id_src = GS == 8 ? (ss < 4 ? ss + 1 : opp_id0 + (ss - 4)) : ss + 1;
                const int id_tgt = GS == 8 ? (j < 4 ? j + 1 : opp_id0 + (j - 4)) : j + 1;
                double u = hh_rng_u01(sh.g_tkey[src / GS], (uint32_t)id_src, HH_SITE_CANNON, (uint32_t)id_tgt);
                if (u < HH_AC_HIT_PROB(t)) bit = 2 << j;
            } else bit = kind == 2 ? (1 << HH_RES_FUSE_BIT(GS)) : (1 << (HH_RES_FUSE_BIT(GS) + 1));
        }
        if (bit) atomicOr(&sh.res[src], bit);
    }
}
template <int A, int B, bool INLINE_EXACT = false>
__device__ __forceinline__ void drain_envelope_queue(Shared<A, B> &sh, int tid, int count) {
    drain_envelope_queue_t<A, B, INLINE_EXACT>(sh, tid, count, 0);
}

__device__ __forceinline__ void arm_cannon(Unit &m) { /* ac1.py:69-70 / ac2.py:65-66 fire_cannon */
    int b = HH_AC_BURST(m.ac_type);
    m.burst = m.cannon_remain < b ? m.cannon_remain : b;
}

/* planar stage of the launch predicate from the LDS pair table (hh_envelope.h); lane's aircraft -> slot j, both where
 * they stood when the table was built: 1 inside / 0 outside / -1 ask the queue */
template <int A, int B>
__device__ __forceinline__ int launch_planar(const Shared<A, B> &sh, int tid, int base, int j) {
    const double la = sh.lat0[tid], lo = sh.lon0[tid], tl = sh.lat0[base + j], to = sh.lon0[base + j];
    const double cross = sh.uc[tid] * (tl - la) - sh.us[tid] * (to - lo);
    return hh_missile_cone_planar(la, lo, tl, to, sh.p_foc[j][tid], cross, sh.p_dist[j][tid]);
}

/* the same stage when no pair table has been built for the published state: the one entry it needs, computed on demand
 * (HighLevelEnv's HL_TICK launch: only the few opponents that try a launch pay for it instead of every lane for the table) */
template <int A, int B>
__device__ __forceinline__ int launch_planar_direct(const Shared<A, B> &sh, int tid, int base, int j) {
    const double la = sh.lat0[tid], lo = sh.lon0[tid], tl = sh.lat0[base + j], to = sh.lon0[base + j];
    const double c1 = sh.uc[tid], s1 = sh.us[tid], n1 = sh.un[tid];
    const double dx = to - lo, dy = tl - la;
    const double n2 = hh_sqrt(dx * dx + dy * dy);
    const double x = hh_clip((c1 * dx + s1 * dy) / (n1 * n2 + 1e-10), -1.0, 1.0);
    return hh_missile_cone_planar(la, lo, tl, to, hh_acos(x) * (180.0 / HH_PI), c1 * dy - s1 * dx, n2);
}

template <int A, int B, bool IX = false>
__device__ __forceinline__ void tick(const DevCfg &c, Shared<A, B> &sh, int tid, int g, int s, int base, bool active,
                                     Unit &m, Arena &ar, const int8_t *act, StepOut &out, uint32_t &ev_mask_out,
                                     const int tmode, const bool run_arena HH_PROF_ARGS) {
    /* tmode 0: fused LowLevelEnv step (commands + tick).  tmode 1: tick only — commands and launches were
     * already applied per side by act_phase (HighLevelEnv sub-steps; steps is advanced by the caller) */
    const int id = s + 1;
    const bool running = tmode == 0 ? (active && !ar.done) : (active && run_arena);
    const bool agent = s < c.nA;
    const bool hl = c.env_kind == HH_ENV_HIGHLEVEL;
    if (tmode == 0) { out.reward = 0.0; out.valid = 0; } /* tmode 1: the caller passes what act_phase produced */
    uint32_t evm = 0;
    out.kill_event = 0;
    if (running && tmode == 0) { ar.steps += 1; arena_rekey(ar); }
    const bool snap = running && m.alive; /* in do_tick's start-of-tick snapshot */
    double opp_stat0 = tmode == 0 ? 0.0 : out.opp_stat0;
    int want_launch = 0, launch_tgt = 0; /* launch_tgt: slot index */
    int wait_after = -1;                 /* scripted opponents: missile_wait value set after the attempt */
    bool base_gate = false;              /* _take_base_action missile gate passed */

    /* ---------------- phase A: commands (env_hetero.py:160-182), pair table = pre-tick state ---------------- */
    if (snap && tmode == 0) {
        if (agent || c.ext_opp) {
            int t = m.n_tgt ? m.tgt0 : 0;
            if (!agent) {
                /* env_base.py:349-398 _policy_actions -> lowlevel_state(opp_mode, i): refresh target */
                Near3 nb;
                nearby(c, sh, tid, base, s, false, nb);
                m.n_tgt = nb.n ? 1 : 0; m.tgt0 = nb.n ? nb.i0 + 1 : 0; m.tgt_d0 = nb.n ? nb.d0 : 0.0;
                t = m.tgt0;
            } else {
                out.valid = 1;
                if (t && sh_alive(sh, base + t - 1)) opp_stat0 = norm180(sh.p_foc[s][base + t - 1]); /* env_hetero.py:169-170 */
            }
            /* env_base.py:214-238 _take_base_action */
            double nh = hh_pymod360(m.hdg + (double)(((int)act[0] - 6) * 15));
            if (nh >= 360.0 || nh < 0.0) nh = 0.0;
            m.cmd_hdg = nh;
            double mx = HH_AC_MAX_SPEED(m.ac_type);
            m.cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
            bool agent_ll = agent && !hl;
            if (act[2] && m.cannon_remain > 0) {
                arm_cannon(m);
                if (agent_ll && c.agent_mode == HH_MODE_ESCAPE && m.cannon_remain < 90) out.reward -= 0.1;
            }
            if (m.ac_type == 1 && act[3]) {
                if (t && m.missile_remain > 0 && !m.has_missile && m.missile_wait == 0) {
                    base_gate = true;
                    want_launch = 1;
                    launch_tgt = t - 1;
                }
            }
        } else if (c.level <= 2) {
            /* env_hetero.py:118-136 levels 1-2 */
            if (c.level == 2) {
                arm_cannon(m);
                bool man = ar.steps <= 5;
                if (!man) man = (ar.steps % hh_rng_randint(d_rng(ar, id, HH_SITE_L2_PERIOD, 0), 35, 45)) <= 5;
                if (man) {
                    int r = hh_rng_randint(d_rng(ar, id, HH_SITE_L2_TURN, 0), 0, 1);
                    m.cmd_hdg = hh_pymod360(m.hdg + (r ? -90.0 : 90.0));
                    m.cmd_spd = (double)(100 + hh_rng_randint(d_rng(ar, id, HH_SITE_L2_SPEED, 0), 0, 4) * 75);
                }
            }
            if (!m.has_missile && (ar.steps % 40) < 3 && hh_rng_randint(d_rng(ar, id, HH_SITE_L12_COIN, 0), 0, 1) &&
                m.missile_wait == 0 && m.ac_type == 1) {
                Near3 nb;
                nearby(c, sh, tid, base, s, false, nb);
                if (nb.n) { want_launch = 1; launch_tgt = nb.i0; wait_after = 5; }
            }
        }
    }
    HH_PROF(11);
    /* env_hetero.py:138-158 level 3: the arena-level escape flag is consumed once per live
     * opponent in id order; every lane replays the tiny integer sequence so that
     * each opponent lane sees the flag as it was at its turn and all lanes agree on the result */
    if (running && tmode == 0 && !c.ext_opp && c.level >= 3 && !hl) {
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
                double y = hh_clip(hh_div_known(m.lat - HH_MAP_LAT0, c.ext_lat, c.inv_ext_lat), 0.0, 1.0);
                double x = hh_clip(hh_div_known(m.lon - HH_MAP_LON0, c.ext_lon, c.inv_ext_lon), 0.0, 1.0);
                double uh = d_rng(ar, id, HH_SITE_ESC_HDG, 0);
                double lo_h = y < 0.5 ? (x < 0.5 ? 30.0 : 300.0) : (x < 0.5 ? 120.0 : 210.0);
                heading = (double)(int)hh_rng_uniform(uh, lo_h, lo_h + 30.0);
                speed = (double)(int)hh_rng_uniform(d_rng(ar, id, HH_SITE_ESC_SPEED, 0), 300.0, 600.0);
                fire = hh_rng_randint(d_rng(ar, id, HH_SITE_ESC_FIRE, 0), 0, 1);
            } else {
                /* env_hetero.py:247-271 _hardcoded_opp */
                Near3 nb;
                nearby(c, sh, tid, base, s, false, nb);
                heading = m.hdg;
                speed = (double)(int)hh_rng_uniform(d_rng(ar, id, HH_SITE_HC_SPEED1, 0), 100.0, 400.0);
                if (nb.n) {
                    int ag = base + nb.i0;
                    /* env_base.py:464-487 _correct_angle_sign */
                    double sn, cs;
                    hh_sincos(hh_pymod360(m.hdg) * (HH_PI / 180.0), &sn, &cs);
                    double x1 = m.lon + hh_round3(sn), y1 = m.lat + hh_round3(cs);
                    double val = (x1 - m.lon) * (sh.lat0[ag] - m.lat) - (sh.lon0[ag] - m.lon) * (y1 - m.lat);
                    double sign = val < 0.0 ? 1.0 : -1.0;
                    double r = hh_rng_uniform(d_rng(ar, id, HH_SITE_HC_R, 0), 0.7, 1.3);
                    double focus = sh.p_foc[nb.i0][tid];
                    if (nb.d0 > 0.008 && focus > 4.0) heading = hh_pymod360(heading + r * sign * focus);
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
            if (fire) arm_cannon(m);
            if (fire_m && opp >= 0 && !m.has_missile && m.missile_wait == 0 && m.ac_type == 1) {
                want_launch = 1; launch_tgt = opp; wait_after = 10;
            }
        }
    }

    HH_PROF(0);
    /* ---------------- phase B: aircraft kinematics + move (ac1.py:81-133) ---------------- */
    const double lat_old = m.lat, lon_old = m.lon, hdg_old = m.hdg;
    bool fired = false;
    const int rk_pre = m.rk_alive;      /* rocket in flight at tick start (before this step's launches) */
    const int has_missile_pre = m.has_missile; /* actual_missile as _take_base_action sees it (before the tick) */
    const bool try_launch = want_launch && !m.has_missile && m.missile_remain > 0; /* ac1.py:73 */
    if (snap) {
        int t = m.ac_type;
        if (m.hdg != m.cmd_hdg) {
            double delta = d_signed_heading_diff(m.hdg, m.cmd_hdg);
            double max_deg = HH_AC_TURN_RATE(t) * 1.0;
            if (hh_fabs(delta) <= max_deg) m.hdg = m.cmd_hdg;
            else { m.hdg += delta >= 0.0 ? max_deg : -max_deg; m.hdg = hh_pymod360(m.hdg); }
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
        if (m.has_missile) { /* ac1.py:117-128, rocket launched in an earlier step */
            if (!m.rk_alive) m.has_missile = 0;
            else m.rk_cmd = hh_clip(m.rk_hdg * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
        }
    }
    /* Moves.  The rocket of this slot — in flight, or the one a pending launch would create (position and
     * heading of the launcher before its update, steered by this tick's noise draw: ac1.py:76-79,127) — is
     * moved speculatively together with the aircraft so that the two RK4 chains overlap; the result is
     * committed after the kill resolution only if the rocket exists and survives (rocket_unit.py:61-73). */
    const bool rk_spec = running && (rk_pre ? m.rk_life <= HH_ROCKET_MAX_LIFE : try_launch);
    double rk_nlat = 0.0, rk_nlon = 0.0, rk_nhdg = 0.0, rk_ncmd = 0.0;
    {
        const bool mv_a = snap && m.spd > 0.0;
#ifdef HH_ABL_NO_MOVE
        const bool any_rk = false;
#else
        const bool any_rk = __ballot(rk_spec) != 0ULL;
#endif
        if (HH_USUAL(any_rk)) {
            const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
            double r_lat = rk_pre ? m.rk_lat : lat_old, r_lon = rk_pre ? m.rk_lon : lon_old;
            double r_hdg = rk_pre ? m.rk_hdg : hdg_old;
            rk_ncmd = rk_pre ? m.rk_cmd
                             : hh_clip(hdg_old * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
            if (r_hdg != rk_ncmd) {
                double delta = d_signed_heading_diff(r_hdg, rk_ncmd);
                if (hh_fabs(delta) <= HH_ROCKET_TURN_RATE) r_hdg = rk_ncmd;
                else r_hdg += delta >= 0.0 ? HH_ROCKET_TURN_RATE : -HH_ROCKET_TURN_RATE;
            }
            rk_nhdg = r_hdg;
            int life = rk_pre ? m.rk_life : 0;
            double r_spd = speed_table[rk_spec ? life : 0];
            double a_lat, a_lon;
            d_geo_move2(m.lat, m.lon, m.hdg, mv_a ? m.spd * HH_KNOTS_TO_MS * 1.0 : 0.0, a_lat, a_lon,
                        rk_spec ? r_lat : 5.0, rk_spec ? r_lon : 7.0, r_hdg, r_spd * HH_KNOTS_TO_MS * 1.0, rk_nlat, rk_nlon);
            if (mv_a) { m.lat = a_lat; m.lon = a_lon; }
        } else {
#ifndef HH_ABL_NO_MOVE
            if (mv_a) d_geo_move(m.lat, m.lon, m.hdg, m.spd * HH_KNOTS_TO_MS * 1.0, m.lat, m.lon);
#endif
        }
    }
    double hv_c, hv_s; /* heading vector after the turn: cannon prefilter now, published with the post-tick state */
    heading_vec(m.hdg, hv_c, hv_s);
    sh.u.t.lat1[tid] = m.lat;
    sh.u.t.lon1[tid] = m.lon;
    sh.u.t.hdg1[tid] = m.hdg;
    sh.u.t.rk_lat[tid] = rk_pre ? m.rk_lat : lat_old;
    sh.u.t.rk_lon[tid] = rk_pre ? m.rk_lon : lon_old;
    sh.aux[tid] = snap ? 1 : 0;
    sh.res[tid] = 0;
    if (tid == 0) sh.u.t.q_count = 0;
    if (s == 0 && active) sh.g_tkey[g] = ar.tkey;
    hh_wg_sync<B>();

    HH_PROF(1);
    /* ---------------- phase Q: enqueue every geodesic envelope test that survives the prefilter ---------------- */
    const int rk_tgt = rk_pre ? m.rk_target - 1 : launch_tgt; /* slot the (possibly pending) rocket is aimed at */
    const bool rk_maybe = running && (rk_pre || try_launch);
    const int launch_pre = try_launch ? launch_planar(sh, tid, base, launch_tgt) : -1; /* pair table = pre-tick geometry */
    {
        /* a lane queues at most 1 launch + (A - 1) cannon candidates + 2 rocket fuses: 8 with six unit slots, 12 with ten (the ten-slot instances with
         * friendly fire: nine cannon candidates) — one code register per entry, so that the count a lane reserves is the count it writes */
        constexpr int QCAP = A > 6 ? 12 : 8;
        static_assert(1 + (A - 1) + 2 <= QCAP, "envelope queue: code registers per lane");
        int nq = 0;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c8 = 0, c9 = 0, c10 = 0, c11 = 0;
#define HH_PUSH(code)                                                                                       \
    do {                                                                                                    \
        int cd_ = (code);                                                                                   \
        switch (nq) { case 0: c0 = cd_; break; case 1: c1 = cd_; break; case 2: c2 = cd_; break; case 3: c3 = cd_; break; \
                      case 4: c4 = cd_; break; case 5: c5 = cd_; break; case 6: c6 = cd_; break;           \
                      default:                                                                              \
                          if (QCAP == 8) c7 = cd_;                                                          \
                          else switch (nq) { case 7: c7 = cd_; break; case 8: c8 = cd_; break; case 9: c9 = cd_; break; case 10: c10 = cd_; break; default: c11 = cd_; break; } \
                          break; }                                                                          \
        nq++;                                                                                               \
    } while (0)
        if (try_launch && launch_pre < 0) HH_PUSH(tid | (0 << 8) | (launch_tgt << 10));
        if (fired) {
            int t = m.ac_type;
#pragma unroll
            for (int j = 0; j < A; j++) {
                if (j == s) continue;
                if (!sh.aux[base + j]) continue; /* not alive at tick start -> can never be "currently alive" */
                bool enemy = agent ? (j >= c.nA) : (j < c.nA);
                if (!(c.friendly_kill || enemy)) continue;
                /* target already moved iff its id is lower (cmano_simulator.py:142) */
                double tl = j < s ? sh.u.t.lat1[base + j] : sh.lat0[base + j];
                double to = j < s ? sh.u.t.lon1[base + j] : sh.lon0[base + j];
                if (d_maybe_within_km(lat_old, lon_old, tl, to, HH_AC_CANNON_KM(t)) &&
                    !hh_cannon_cone_planar_outside(lat_old, lon_old, tl, to, hv_c, hv_s, t))
                    HH_PUSH(tid | (1 << 8) | (j << 10));
            }
        }
        if (rk_maybe) {
            double rl = sh.u.t.rk_lat[tid], ro = sh.u.t.rk_lon[tid];
            if (d_maybe_within_km(rl, ro, sh.u.t.lat1[base + rk_tgt], sh.u.t.lon1[base + rk_tgt], HH_ROCKET_FUSE_KM))
                HH_PUSH(tid | (2 << 8) | (rk_tgt << 10));
            if (c.friendly_kill) {
                int fid = s == 1 ? 0 : 1; /* rocket_unit.py:46: 1 if source.id == 2 else 2 */
                if (d_maybe_within_km(rl, ro, sh.u.t.lat1[base + fid], sh.u.t.lon1[base + fid], HH_ROCKET_FUSE_KM))
                    HH_PUSH(tid | (3 << 8) | (fid << 10));
            }
        }
#undef HH_PUSH
        if (HH_RARE(nq != 0)) {
            int at = atomicAdd(&sh.u.t.q_count, nq);
            sh.u.t.q_code[at] = c0;
            if (nq > 1) sh.u.t.q_code[at + 1] = c1;
            if (nq > 2) sh.u.t.q_code[at + 2] = c2;
            if (nq > 3) sh.u.t.q_code[at + 3] = c3;
            if (nq > 4) sh.u.t.q_code[at + 4] = c4;
            if (nq > 5) sh.u.t.q_code[at + 5] = c5;
            if (nq > 6) sh.u.t.q_code[at + 6] = c6;
            if (nq > 7) sh.u.t.q_code[at + 7] = c7;
            if (QCAP > 8) {
                if (nq > 8) sh.u.t.q_code[at + 8] = c8;
                if (nq > 9) sh.u.t.q_code[at + 9] = c9;
                if (nq > 10) sh.u.t.q_code[at + 10] = c10;
                if (nq > 11) sh.u.t.q_code[at + 11] = c11;
            }
        }
    }
    hh_wg_sync<B>();

    HH_PROF(2);
    /* ---------------- phase I: dense pass over the queue (estimate filter + out-of-line exact Karney) ---------------- */
    drain_envelope_queue<A, B, IX>(sh, tid, sh.u.t.q_count);
    hh_wg_sync<B>();

    HH_PROF(3);
    /* ---------------- phase L: launch bookkeeping (env_base.py:227-236, ac1.py:76-79) ---------------- */
    const int myres = sh.res[tid] | (launch_pre == 1 ? 1 : 0);
    int launched = 0;
    if (try_launch && (myres & 1)) {
        launched = 1;
        m.rk_alive = 1; m.rk_lat = lat_old; m.rk_lon = lon_old; m.rk_hdg = hdg_old;
        m.rk_target = launch_tgt + 1; m.rk_life = 0;
        m.has_missile = 1;
        m.missile_remain = m.missile_remain - 1 > 0 ? m.missile_remain - 1 : 0;
        evm |= HH_EV_BIT(A, 3, s, agent);
        /* the launcher's own update in this tick already steers it (ac1.py:127) */
        m.rk_cmd = rk_ncmd;
    }
    if (base_gate) {
        double uu = d_rng(ar, id, HH_SITE_MISSILE_WAIT, 0);
        m.missile_wait = hl ? hh_rng_randint(uu, 8, 12) : hh_rng_randint(uu, 7, 17);
        if (agent && !hl && c.agent_mode == HH_MODE_ESCAPE && m.missile_remain < 3) out.reward -= 0.1;
    }
    if (snap && tmode == 0 && (agent || c.ext_opp)) { /* env_base.py:235-236, evaluated before do_tick */
        if (m.missile_wait > 0 && !(launched || has_missile_pre)) m.missile_wait -= 1;
    }
    if (want_launch && wait_after >= 0) m.missile_wait = wait_after;
    const int rk_at_start = m.rk_alive; /* rockets in do_tick's snapshot: in flight + launched this step */
    sh.aux[tid] = launched | ((fired ? (myres >> 1) & (A > 8 ? 0x3ff : 0xff) : 0) << 8);
    hh_wg_sync<B>();
    {   /* launch order = unit id order (cmano_simulator.py:104-108): seq = running id counter */
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            int l = active ? (sh.aux[base + j] & 1) : 0;
            total += l;
            if (j < s) before += l;
        }
        if (launched) m.rk_seq = ar.next_seq + before + 1;
        ar.next_seq += total;
    }
    int rkw = 0; /* bit0 present, bit1 fuse on target, bit2 fuse on "friendly", bit3 end of life, bits4-6 target, bits 8.. seq */
    if (running && rk_at_start) {
        int eol = m.rk_life > HH_ROCKET_MAX_LIFE;
        rkw = 1 | (((myres >> HH_RES_FUSE_BIT(A)) & 1) << 1) | (((myres >> (HH_RES_FUSE_BIT(A) + 1)) & 1) << 2) | (eol << 3) | ((m.rk_target - 1) << 4) | (m.rk_seq << 8);
    }
    sh.res[tid] = rkw; /* res was consumed into myres above; reuse it for the rocket word */

    /* ---------------- phases C + D: id-ordered resolution by one lane per arena ---------------- */
    hh_wg_sync<B>();
    if (s == 0 && active) {
        int alive = 0, nev = 0, dead = 0;
        int aux_[A], res_[A];
#pragma unroll
        for (int j = 0; j < A; j++) { aux_[j] = sh.aux[base + j]; res_[j] = sh.res[base + j]; alive |= (sh_alive(sh, base + j) ? 1 : 0) << j; }
        if (running) {
            /* aircraft phase: shooter i (alive at tick start, even if killed earlier in this tick) hits the
             * still-alive targets in id order (ac1.py:106-115) */
#pragma unroll
            for (int i = 0; i < A; i++) {
                int ci = aux_[i] >> 8;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    if (((ci >> j) & 1) && ((alive >> j) & 1)) {
                        alive &= ~(1 << j);
                        sh.g_ev[g][nev++] = i | (j << 4);
                    }
                }
            }
            /* rocket phase in launch order (rocket_unit.py:37-58) */
            int done_mask = 0;
            for (int k = 0; k < A; k++) {
                int best = -1, best_seq = 0x7fffffff;
                int w = 0;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    int wj = res_[j];
                    if ((wj & 1) && !((done_mask >> j) & 1) && (wj >> 8) < best_seq) { best = j; best_seq = wj >> 8; w = wj; }
                }
                if (best < 0) break;
                done_mask |= 1 << best;
                int tg = (w >> 4) & 15;
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
    hh_wg_sync<B>();
    if (running && rk_at_start) {
        if ((sh.g_rkdead[g] >> s) & 1) {
            m.rk_alive = 0; m.rk_target = 0; m.rk_life = 0; m.rk_seq = 0;
            m.rk_lat = m.rk_lon = m.rk_hdg = m.rk_cmd = 0.0;
        } else { /* commit the speculative turn + move (rocket_unit.py:61-73) */
            m.rk_hdg = rk_nhdg; m.rk_lat = rk_nlat; m.rk_lon = rk_nlon;
            m.rk_life += 1;
        }
    }

    HH_PROF(4);
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
    const int nev = active ? sh.g_nev[g] : 0;
    if (running && agent) {
        double sc = c.rew_scale;
        if (oob) { rews += (hl ? -2.0 : -5.0) * sc; destroyed = 1; }
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
    /* event masks for parity checks */
    for (int e = 0; e < nev; e++) {
        int w = sh.g_ev[g][e];
        evm |= HH_EV_BIT(A, (w >> 8) & 1, (w >> 4) & 15, 0);
    }
    if (oob) evm |= HH_EV_BIT(A, 2, s, 0);
    ev_mask_out = evm;
    sh.aux[tid] = oob;
    sh.rew[tid] = rews;
    /* post-tick state + pair table: escape shaping now, observation next, pre-step lookups of the next tick */
    HH_PROF(5);
    publish_obs_hv(c, sh, tid, m, hv_c, hv_s);
    hh_wg_sync<B>();
    HH_PROF(6);
    pair_tables(sh, tid, base, s, active);
    HH_PROF(7);
    if (running) {
        int ag = 0, op = 0, kill = nev > 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            int al = sh_alive(sh, base + j);
            if (j < c.nA) ag += al; else op += al;
            kill |= sh.aux[base + j]; /* out-of-bounds removals */
        }
        out.kill_event = kill;
        if (tmode == 0) ar.done = (ag <= 0 || op <= 0 || ar.steps >= c.horizon) ? 1 : 0;
    }
    hh_wg_sync<B>();
    if (running && agent) {
        if (!hl && c.agent_mode == HH_MODE_ESCAPE && c.esc_dist_rew && m.alive) {
            /* env_hetero.py:198-214 */
            Near3 nb;
            nearby(c, sh, tid, base, s, false, nb);
            double dr[3] = {nb.r0, nb.r1, nb.r2};
#pragma unroll
            for (int j = 1; j <= 3; j++) {
                if (j > nb.n) break;
                if (dr[j - 1] < 0.06) { rews += -0.02 / j; if (m.spd < 200.0) rews += -0.02 / j; }
                else if (dr[j - 1] > 0.13) { rews += 0.02 / j; if (m.spd > 500.0) rews += 0.02 / j; }
            }
        }
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
    hh_wg_sync<B>(); /* all reads of rew/aux/g_* done before the caller reuses them */
    HH_PROF(8);
}

/* index into the stored target list like Python: commander_actions[i]-1, with -1 = last */
template <bool X = false> /* X: ten-slot arenas, lists of up to five (an opponent's agents) */
__device__ __forceinline__ int hl_target_slot(const Unit &m, double &dist) {
    /* fields are read into values first: a select between the addresses of struct members would keep the whole Unit in
     * scratch memory instead of registers */
    const int n_tgt = m.n_tgt, t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2;
    const double d0 = m.tgt_d0, d1 = m.tgt_d1, d2 = m.tgt_d2;
    int k = m.cmd_act > 0 ? m.cmd_act - 1 : n_tgt - 1;
    int t = 0;
    double d = 0.0;
    if (k == 0 && n_tgt > 0) { t = t0; d = d0; }
    if (k == 1 && n_tgt > 1) { t = t1; d = d1; }
    if (k == 2 && n_tgt > 2) { t = t2; d = d2; }
    if constexpr (X) {
        const UnitW &x = static_cast<const UnitW &>(m);
        const int t3 = x.tgt3, t4 = x.tgt4;
        const double d3 = x.tgt_d3, d4 = x.tgt_d4;
        if (k == 3 && n_tgt > 3) { t = t3; d = d3; }
        if (k == 4 && n_tgt > 4) { t = t4; d = d4; }
    }
    dist = d;
    return t; /* 1-based unit id, 0 = none */
}

/* env_base.py:214-238 _take_base_action for the lanes selected by `acts` (one side), including the missile
 * envelope test (one pass over the workgroup queue) and launch bookkeeping.  Used where pilot / frozen-policy
 * inference runs between the two sides' actions: HighLevelEnv sub-steps (hl) and LowLevelEnv levels 4-5. */
template <int A, int B, bool IX = false, bool TAB = true>
__device__ __forceinline__ void act_phase(const DevCfg &c, Shared<A, B> &sh, int tid, int s, int base, bool active, bool running,
                                          Unit &m, Arena &ar, const int8_t *act, bool acts, bool hl, double &pre_reward,
                                          double &opp_stat0, int &valid, uint32_t &evm) {
    const int id = s + 1;
    const bool agent = s < c.nA;
    const bool snap = active && running && m.alive && acts;
    int want_launch = 0, launch_tgt = 0;
    bool base_gate = false;
    if (snap) {
        double dd;
        int t = hl ? hl_target_slot<(A > 8)>(m, dd) : (m.n_tgt ? m.tgt0 : 0);
        if (!hl && agent) { /* env_hetero.py:168-170 */
            valid = 1;
            if (t && sh_alive(sh, base + t - 1)) opp_stat0 = norm180(sh.p_foc[s][base + t - 1]);
        }
        double nh = hh_pymod360(m.hdg + (double)(((int)act[0] - 6) * 15));
        if (nh >= 360.0 || nh < 0.0) nh = 0.0;
        m.cmd_hdg = nh;
        double mx = HH_AC_MAX_SPEED(m.ac_type);
        m.cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
        const bool agent_ll = agent && !hl;
        if (act[2] && m.cannon_remain > 0) {
            arm_cannon(m);
            if (agent_ll && c.agent_mode == HH_MODE_ESCAPE && m.cannon_remain < 90) pre_reward -= 0.1;
        }
        if (m.ac_type == 1 && act[3]) {
            if (t && m.missile_remain > 0 && !m.has_missile && m.missile_wait == 0) {
                base_gate = true;
                want_launch = 1;
                launch_tgt = t - 1;
            }
        }
    }
    const bool try_launch = want_launch && !m.has_missile && m.missile_remain > 0;
    sh.res[tid] = 0;
    if (tid == 0) sh.u.t.q_count = 0;
    hh_wg_sync<B>();
    const int launch_pre = !try_launch ? -1 : (TAB ? launch_planar(sh, tid, base, launch_tgt) : launch_planar_direct(sh, tid, base, launch_tgt));
    if (try_launch && launch_pre < 0) {
        int at = atomicAdd(&sh.u.t.q_count, 1);
        sh.u.t.q_code[at] = tid | (0 << 8) | (launch_tgt << 10);
    }
    hh_wg_sync<B>();
    drain_envelope_queue<A, B, IX>(sh, tid, sh.u.t.q_count);
    hh_wg_sync<B>();
    int launched = 0;
    if (try_launch && ((sh.res[tid] & 1) || launch_pre == 1)) { /* ac1.py:76-79 */
        launched = 1;
        m.rk_alive = 1; m.rk_lat = m.lat; m.rk_lon = m.lon; m.rk_hdg = m.hdg; m.rk_cmd = m.hdg;
        m.rk_target = launch_tgt + 1; m.rk_life = 0;
        m.has_missile = 1;
        m.missile_remain = m.missile_remain - 1 > 0 ? m.missile_remain - 1 : 0;
        evm |= HH_EV_BIT(A, 3, s, agent);
    }
    if (base_gate) {
        double uu = d_rng(ar, id, HH_SITE_MISSILE_WAIT, 0);
        m.missile_wait = hl ? hh_rng_randint(uu, 8, 12) : hh_rng_randint(uu, 7, 17);
        if (agent && !hl && c.agent_mode == HH_MODE_ESCAPE && m.missile_remain < 3) pre_reward -= 0.1;
    }
    if (snap) {
        if (m.missile_wait > 0 && !m.has_missile) m.missile_wait -= 1;
    }
    sh.aux[tid] = launched;
    hh_wg_sync<B>();
    {   /* rocket ids in unit id order (cmano_simulator.py:104-108) */
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
    /* weapon flags other lanes observe (env_base.py:208-211) */
    int shot = m.burst > 0 || (m.ac_type == 1 && m.has_missile);
    sh.flags[tid] = (m.alive ? FL_ALIVE : 0) | ((m.ac_type & 3) << 1) | (shot ? FL_SHOT : 0);
    hh_wg_sync<B>();
}

/* hh_opp_policy: k of every arena's current episode (level 5, fight mode) */
__global__ __launch_bounds__(256) void hh_k_opp_policy(DevPtrs P, DevCfg c, int8_t *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= c.N) return;
    int k = 0;
    if (c.env_kind == HH_ENV_LOWLEVEL && c.level == 5 && c.agent_mode == HH_MODE_FIGHT)
        k = hh_l5_policy_pick(hh_rng_arena_key(c.seed, c.arena_offset + (uint64_t)n), (uint32_t)P.ar_pack[n].y);
    out[n] = (int8_t)k;
}

/* ===================================================================== the kernel */
enum { HH_RUN_ROLLOUT = 0, HH_RUN_RESET = 1, HH_RUN_OBSERVE = 2, HH_RUN_LL_BEGIN = 3, HH_RUN_LL_FINISH = 4 };

/* W = waves per SIMD the register allocation is held to: 1 = no spills, lowest per-tick latency (few arenas);
 * 2 = 256 registers per lane, spills to scratch but two resident waves per SIMD: +35 % throughput once there
 * are more waves than SIMDs (> 16384 arenas).  Same source, same results. */
/* SPLIT adds the two half-step run modes of LowLevelEnv levels 4-5 (frozen opponent policies, env_hetero.py:160-172):
 *   LL_BEGIN   steps += 1, agents' _take_base_action          -> observations of the opponents [N, n_opps, 30] in obs_out
 *   LL_FINISH  opponents' _take_base_action, then the tick, rewards, done, reset, agents' observation like ROLLOUT (T = 1)
 * `actions` holds the acting side's rows only; `mask` carries nothing; opp_mode (fight 0 / escape 1) arrives in T for LL_BEGIN. */
template <int A, int B, int W, bool SPLIT>
__global__ __launch_bounds__(B, W) void hh_k_world(DevPtrs P, DevCfg c, int run, int T, const int8_t *__restrict__ actions,
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
    HH_PROF_DECL;
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
    sh.aux[tid] = 0;
    if (run == HH_RUN_RESET && P.pol_lut && blockIdx.x == 0 && tid <= 8) P.pol_counts[tid * HH_BIN_STRIDE] = 0; /* rows nobody consumed */
    bool need_reset = run == HH_RUN_RESET && active && (mask == nullptr || mask[n]);
    uint32_t evm_last = 0;
    int act_fault = 0; /* some action word of this lane was out of range and ran sanitised (hh_act_unpack) */
    int tcur = (P.trace != nullptr && active && n < P.trace_K) ? P.trace_pos[n] : 0; /* trace cursor of the lane's arena */
    if (run == HH_RUN_ROLLOUT || run >= HH_RUN_LL_BEGIN) { /* pair table of the pre-tick state */
        publish_obs(c, sh, tid, m);
        hh_wg_sync<B>();
        pair_tables(sh, tid, base, s, active);
        hh_wg_sync<B>();
    }
    if constexpr (SPLIT) {
        if (run == HH_RUN_LL_BEGIN) {
            /* T < 0 (HH_OPP_MODE_EPISODE): the arena's own level-5 draw of this episode decides (env_hetero.py:55-59) */
            const int opp_mode = T >= 0 ? T : (hh_l5_policy_pick(ar.akey, (uint32_t)ar.episode) == 5 ? HH_MODE_ESCAPE : HH_MODE_FIGHT);
            const bool running = active && !ar.done;
            if (running) { ar.steps += 1; arena_rekey(ar); }
            int8_t act[4] = {0, 0, 0, 0};
            if (active && s < c.nA) {
                int w = *reinterpret_cast<const int *>(actions + ((size_t)n * c.nA + s) * 4);
                hh_act_unpack(w, act, act_fault, running && m.alive);
            }
            double pre = 0.0, os0 = 0.0;
            int valid = 0;
            act_phase<A, B, (W >= 2)>(c, sh, tid, s, base, active, running, m, ar, act, s < c.nA, false, pre, os0, valid, evm_last);
            if (active) {
                m.cmd_act = valid;          /* reward key present (agents alive at step start) */
                P.acc_rew[u] = pre;         /* escape-mode ammunition penalties (env_base.py:223-233) */
                m.tgt_d1 = os0;             /* tgt_d1 is unused by LowLevelEnv: carries opp_stats[i][0] to LL_FINISH */
            }
            /* a bound policy bank (hh_bind_policy): the opponents' rows [N, n_opps] are binned by network here — selector = policy type |
             * aircraft type << 2, + 16 (k - 3) for the policy set k of the arena's level-5 draw (pilots.OpponentNets' encoding) */
            int pslot = 0;
            HhBinTicket bt{0, 0};
            if (P.pol_lut && obs_out) {
                int sb = 0;
                if (running && m.alive && s >= c.nA) {
                    const int k = c.level == 5 ? (T >= 0 ? (opp_mode == HH_MODE_ESCAPE ? 5 : 3) : hh_l5_policy_pick(ar.akey, (uint32_t)ar.episode)) : 3;
                    sb = ((opp_mode == HH_MODE_ESCAPE ? 2 : 1) | (m.ac_type << 2)) + 16 * (k - 3);
                }
                pslot = sb ? (int)P.pol_lut[sb] : 0;
                bt = hh_bin_rows_issue(P.pol_counts, pslot);
            }
            /* observation of the frozen-policy opponents, after the agents acted (shot flags refreshed by act_phase) */
            hh_wg_sync<B>(); /* the queue area of act_phase is free: stage the rows there, store them coalesced */
            if (active && s >= c.nA) {
                float *row = &sh.u.obs[(g * c.nO + (s - c.nA)) * 30];
                if (running && m.alive) lowlevel_obs<A, B>(c, sh, tid, base, s, opp_mode, m, row, 30);
                else for (int q = 0; q < 30; q++) row[q] = 0.0f;
            }
            hh_wg_sync<B>();
            if (obs_out) {
                const int rows = min(GPB, c.N - (int)blockIdx.x * GPB);
                const int cnt = rows * c.nO * 30;
                float *dst = obs_out + (size_t)blockIdx.x * GPB * c.nO * 30;
                for (int q = tid; q < cnt; q += B) dst[q] = sh.u.obs[q];
                if (P.pol_lut) hh_bin_rows_finish(bt, P.pol_lists, P.pol_max_rows, n * c.nO + (s - c.nA), pslot);
            }
            hh_wg_sync<B>();
            T = 0; /* no tick in this launch */
        } else if (run == HH_RUN_LL_FINISH) {
            T = 1;
        }
    }
    /* the action word of tick t+1 is requested while tick t computes (one wave per SIMD cannot hide a ~1 us
     * HBM round trip at the top of every tick) */
    const bool has_act = active && s < c.n_ctrl && run == HH_RUN_ROLLOUT;
    int act_next = 0;
    if (has_act && T > 0) act_next = *reinterpret_cast<const int *>(actions + (((size_t)0 * c.N + n) * c.n_ctrl + s) * 4);
    for (int t = 0; t < T; t++) {
        if (run == HH_RUN_ROLLOUT || (SPLIT && run == HH_RUN_LL_FINISH)) {
            StepOut so;
            int8_t act[4] = {0, 0, 0, 0};
            if (has_act) {
                int w = act_next;
                if (t + 1 < T) act_next = *reinterpret_cast<const int *>(actions + (((size_t)(t + 1) * c.N + n) * c.n_ctrl + s) * 4);
                hh_act_unpack(w, act, act_fault, !ar.done && m.alive);
            }
            const bool was_running = active && !ar.done;
            if constexpr (SPLIT) {
                if (run == HH_RUN_LL_FINISH) {
                    if (active && s >= c.nA) {
                        int w = *reinterpret_cast<const int *>(actions + ((size_t)n * c.nO + (s - c.nA)) * 4);
                        hh_act_unpack(w, act, act_fault, was_running && m.alive);
                    }
                    double pre = 0.0, os0 = 0.0;
                    int vl = 0;
                    uint32_t evm_act = 0;
                    act_phase<A, B, (W >= 2)>(c, sh, tid, s, base, active, was_running, m, ar, act, s >= c.nA, false, pre, os0, vl, evm_act);
                    so.reward = (active && s < c.nA) ? P.acc_rew[u] : 0.0;
                    so.valid = (active && s < c.nA) ? m.cmd_act : 0;
                    so.opp_stat0 = m.tgt_d1;
                    uint32_t evm_tick = 0;
                    tick<A, B, (W >= 2)>(c, sh, tid, g, s, base, active, m, ar, act, so, evm_tick, 1, was_running HH_PROF_PASS);
                    evm_last |= evm_act | evm_tick;
                    if (was_running) {
                        int ag = 0, op = 0;
#pragma unroll
                        for (int j = 0; j < A; j++) { int al = sh_alive(sh, base + j); if (j < c.nA) ag += al; else op += al; }
                        ar.done = (ag <= 0 || op <= 0 || ar.steps >= c.horizon) ? 1 : 0;
                    }
                    if (active) { m.cmd_act = 0; m.tgt_d1 = 0.0; }
                } else {
                    tick<A, B, (W >= 2)>(c, sh, tid, g, s, base, active, m, ar, act, so, evm_last, 0, true HH_PROF_PASS);
                }
            } else {
                tick<A, B, (W >= 2)>(c, sh, tid, g, s, base, active, m, ar, act, so, evm_last, 0, true HH_PROF_PASS);
            }
            /* outputs of this tick */
            if (active && s < c.nA) {
                size_t o = ((size_t)t * c.N + n) * c.nA + s;
                if (reward_out) reward_out[o] = (float)so.reward;
                if (valid_out) valid_out[o] = (uint8_t)so.valid;
            }
            /* episode statistics: one lane per arena, agent order */
            sh.rew[tid] = so.valid ? so.reward : 0.0;
            hh_wg_sync<B>();
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
            if (active && s == 0 && done_out) done_out[(size_t)t * c.N + n] = (uint8_t)ar.done;
            if (was_running) trace_append(P, A, n, s, m, ar, tcur);
            need_reset = active && ar.done && c.auto_reset;
        }
        const int any_reset = hh_wg_sync_or<B>(need_reset ? 1 : 0);
        if (need_reset) { /* K3 */
            reset_arena_scalars(ar);
            reset_unit<A>(c, s, m, ar);
            ep_ret = 0.0;
            trace_append(P, A, n, s, m, ar, tcur); /* first row of the new episode */
        }
        if (run != HH_RUN_ROLLOUT || any_reset) { /* state changed (or never published) */
            publish_obs(c, sh, tid, m);
            hh_wg_sync<B>();
            pair_tables(sh, tid, base, s, active);
            hh_wg_sync<B>();
        }
        need_reset = false;
        HH_PROF(9);
        /* K2: observation rows staged in LDS, then written with unit-stride stores */
#ifndef HH_ABL_NO_OBS
        if (active && s < c.nA) lowlevel_obs<A, B>(c, sh, tid, base, s, c.agent_mode, m, &sh.u.obs[(g * c.nA + s) * D], D);
#endif
        hh_wg_sync<B>();
        if (obs_out) {
            const int rows = min(GPB, c.N - (int)blockIdx.x * GPB);
            const int cnt = rows * c.nA * D;
            float *dst = obs_out + ((size_t)t * c.N + (size_t)blockIdx.x * GPB) * c.nA * D;
            if (run == HH_RUN_RESET && mask != nullptr) {
                for (int k = tid; k < cnt; k += B) if (mask[blockIdx.x * GPB + k / (c.nA * D)]) dst[k] = sh.u.obs[k];
            } else {
                for (int k = tid; k < cnt; k += B) dst[k] = sh.u.obs[k];
            }
        }
        hh_wg_sync<B>();
        HH_PROF(10);
    }
    HH_PROF_FLUSH;
    if (active) {
        unit_store(P, U, u, m);
        if (s == 0) {
            if (P.trace != nullptr && n < P.trace_K) P.trace_pos[n] = tcur;
            arena_store(P, n, ar);
            P.ep_ret[n] = ep_ret;
            if (run == HH_RUN_ROLLOUT || run == HH_RUN_LL_BEGIN) P.ev_mask[n] = 0;
        }
    }
    if (run == HH_RUN_ROLLOUT || run >= HH_RUN_LL_BEGIN) {
        /* OR-reduce the per-lane event bits of the last tick into the arena word */
        __syncthreads();
        if (active && evm_last) atomicOr(&P.ev_mask[n], evm_last);
        hh_act_fault_commit(P, n, active, act_fault);
    }
}

#endif /* HH_KERNELS_H */
