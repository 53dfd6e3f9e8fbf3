/*
 * hh_kernels_quad.h — the 2-vs-2 (LowLevelEnv) rollout kernel with REGISTER exchange.
 *
 * Same path, same arithmetic and same results as hh_k_world<4, 64, W, false> run in ROLLOUT mode (hh_kernels.h):
 *   envs/env_base.py:79-109 step -> envs/env_hetero.py:105-186 _take_action -> cmano_simulator.py:138-157 do_tick
 *   -> env_hetero.py:188-225 rewards -> env_hetero.py:65-103 state, plus env_base.py:62-77 reset on done.
 *
 * What differs is how the four aircraft of an arena see each other.  An arena is one aligned QUAD of lanes, so
 * every cross-aircraft read is a DPP quad permute (v_mov_b32_dpp quad_perm, a VALU modifier: no memory, no
 * wait) instead of an LDS write + s_waitcnt + read.  At one wave per SIMD there is nothing to hide an LDS
 * round trip behind; the generic kernel spends ~37 % of its wave cycles in s_waitcnt (profiles/README.md).
 *   - the per-arena pair table (distance, focus both ways, heading difference, normalised observation
 *     entries, flags, positions) lives in registers, indexed by RELATIVE slot k = (j - s) & 3;
 *   - the id-ordered kill resolution runs redundantly on all four lanes from quad-broadcast words (SIMT: the
 *     same cost as one lane doing it) so its result needs no broadcast back;
 *   - the envelope tests keep the workgroup-wide LDS queue (work compaction across arenas is what LDS is for),
 *     but slots are assigned with wave ballots, so the count is a scalar and an empty queue costs nothing.
 * DPP reads are only made from wave-uniform control flow (a disabled source lane would read as 0).
 *
 * The generic kernel remains the implementation of RESET / OBSERVE / the split step and of 3-vs-3.
 */
#ifndef HH_KERNELS_QUAD_H
#define HH_KERNELS_QUAD_H

#include "hh_kernels.h"

#define HH_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int CTRL>
__device__ __forceinline__ int q_perm_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float q_perm_f(float v) { return __int_as_float(q_perm_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double q_perm_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = q_perm_i<CTRL>(lo);
    hi = q_perm_i<CTRL>(hi);
    return __hiloint2double(hi, lo);
}
/* value held by slot (s + K) & 3 of the lane's arena */
#define HH_QROT(K) HH_QP((K) & 3, ((K) + 1) & 3, ((K) + 2) & 3, ((K) + 3) & 3)
template <int K> __device__ __forceinline__ int q_rot_i(int v) { return q_perm_i<HH_QROT(K)>(v); }
template <int K> __device__ __forceinline__ float q_rot_f(float v) { return q_perm_f<HH_QROT(K)>(v); }
template <int K> __device__ __forceinline__ double q_rot_d(double v) { return q_perm_d<HH_QROT(K)>(v); }
/* value held by absolute slot J */
template <int J> __device__ __forceinline__ int q_bc_i(int v) { return q_perm_i<HH_QP(J, J, J, J)>(v); }
template <int J> __device__ __forceinline__ double q_bc_d(double v) { return q_perm_d<HH_QP(J, J, J, J)>(v); }
/* the other aircraft of the same side (slot s ^ 1) */
__device__ __forceinline__ double q_mate_d(double v) { return q_perm_d<HH_QP(1, 0, 3, 2)>(v); }

/* lanes 32..63 of a wave that carries 8 arenas in lanes 0..31 (the small-world form) are HELPERS of lane - 32: one
 * v_permlane32_swap_b32 hands a word down (main -> helper, main keeps its own) or up (helper -> main, helper keeps its own) */
__device__ __forceinline__ int q_down_i(int v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[0]; }
__device__ __forceinline__ int q_up_i(int v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1]; }
__device__ __forceinline__ double q_down_d(double v) { return __hiloint2double(q_down_i(__double2hiint(v)), q_down_i(__double2loint(v))); }
__device__ __forceinline__ double q_up_d(double v) { return __hiloint2double(q_up_i(__double2hiint(v)), q_up_i(__double2loint(v))); }

/* table lookup by relative slot k in 1..3 (k outside -> entry 3).  Operands by VALUE: a select between array
 * addresses would pin the table in scratch memory. */
template <class T>
__device__ __forceinline__ T q_sel3(T a1, T a2, T a3, int k) {
    T r = a3;
    if (k == 2) r = a2;
    if (k == 1) r = a1;
    return r;
}
#define q_sel(arr, k) q_sel3((arr)[0], (arr)[1], (arr)[2], (k))

/* what the other lanes read from this aircraft */
struct QPub {
    double uc, us, un;             /* heading unit vector and its norm (env_base.py:428) */
    float nlat, nlon, nspd, nhdg;  /* normalised observation entries (env_base.py:117-121) */
    int flags;                     /* FL_ALIVE | type << 1 | FL_SHOT */
};

/* the arena as seen from this lane, entry k-1 = slot (s + k) & 3 */
struct QTab {
    double lat[3], lon[3];                    /* position */
    double dist[3], foc[3], focr[3], hd[3];   /* planar distance [deg], focus me->k, focus k->me, heading difference */
    float nlat[3], nlon[3], nspd[3], nhdg[3];
    int fl[3];
    int amask;                                /* alive bits by ABSOLUTE slot, own bit included */
};

/* position / speed / heading part (final once the aircraft has moved) and the status flags (final at the end of the tick).
 * _vec: the heading unit vector — read by the tick itself (cannon prefilter, launch test) and by the pair table; _norm: the normalised
 * observation entries — read only by whoever formats observation rows */
__device__ __forceinline__ void quad_publish_vec(const Unit &m, QPub &p) {
    double sn, cs;
    hh_sincos(hh_pymod360(90.0 - m.hdg) * (HH_PI / 180.0), &sn, &cs);
    p.uc = cs;
    p.us = sn;
    p.un = hh_sqrt(cs * cs + sn * sn);
}
__device__ __forceinline__ void quad_publish_norm(const DevCfg &c, const Unit &m, QPub &p) {
    p.nlat = (float)hh_clip(hh_div_known(m.lat - HH_MAP_LAT0, c.ext_lat, c.inv_ext_lat), 0.0, 1.0);
    p.nlon = (float)hh_clip(hh_div_known(m.lon - HH_MAP_LON0, c.ext_lon, c.inv_ext_lon), 0.0, 1.0);
    p.nspd = (float)hh_clip(hh_div_known(m.spd, HH_AC_MAX_SPEED(m.ac_type), HH_AC_INV_MAX_SPEED(m.ac_type)), 0.0, 1.0);
    p.nhdg = (float)hh_clip(HH_DIVC(hh_pymod359(m.hdg), 359.0), 0.0, 1.0);
}
__device__ __forceinline__ void quad_publish_motion(const DevCfg &c, const Unit &m, QPub &p) {
    quad_publish_vec(m, p);
    quad_publish_norm(c, m, p);
}

/* ---- the pair table on the OUTPUT wave (two-wave forms compiled for a preset configuration, OWT below) ----
 * What a tick's post-move pair table needs — four positions and heading vectors — is final at the end of phase B, a third of the way
 * into the tick, and nothing the simulation wave does between there and the end of the tick (envelope tests, kill resolution, rewards)
 * reads it.  So the simulation wave hands the moved positions to the output wave at a workgroup barrier (X), goes on with the envelope
 * phases, and takes the finished distances and focus angles back at the second barrier of the tick (Y), while the output wave — on
 * another SIMD of the CU — runs the table's two square roots and four acos chains (16 % of a tick) and then formats the observation
 * rows from ITS copy of the table.  Only integers cross at Y in the other direction (status flags, ammunition, reward, done).
 * All three mailboxes are indexed by lane and single-buffered: every write is separated from every read of the other wave by one of
 * the two barriers. */
struct QPosMail { double lat[64], lon[64], spd[64], hdg[64]; int ac_type[64], steps[64], episode[64], escw[64]; };   /* sim -> out at X; escw: the arena's
                                                                      escape flag | timer << 8 after this tick's script | alive mask at tick start << 16 */
struct QTabMail { double dist[3][64], foc[3][64], focr[3][64]; double uc[64], us[64], sx[64], sy[64]; unsigned long long tk[64];
                  double sp_hdg[64], sp_spd[64]; int sp_w[64];      /* out -> sim at Y; sp_*: QPre */
                  double nr0[64], nfoc[64], nfocr[64]; int nbw[64]; }; /* QTgt / Near2 of the lane, computed by the output wave on the prediction that the tick
                                                                      * changes no alive mask (QPre.spec): nbw = n | j0 << 2 | k0 << 4 */
/* What the simulation wave of the OWT forms reads of the post-tick pair table during the next tick is ONE relative slot per lane — the lane's launch target
 * (env_base.py:227-236: opp_to_attack for an agent, the script's nearest agent for an opponent): its planar distance and focus angle in the launch
 * stage, its focus on the lane in one reward term (env_hetero.py:169-170 opp_stats).  So that wave carries these three doubles across the tick instead of the
 * table's nine, and on the four ticks in five that change no alive mask it does not even select them: the output wave, which runs _nearby_object on its
 * own copy of the table for the speculated script anyway, hands them over with (n, j0, k0) — 12 LDS reads instead of 17 and no quad_nearby behind barrier Y. */
struct QTgt { double foc, focr, dist; };
/* The heading unit vector after the turn (a sincos and a square root) is computed by the OUTPUT wave too: the table needs it exactly, the simulation wave
 * needs it inside the tick only for the cannon prefilter (hh_envelope.h: hh_cannon_cone_planar_outside, a one-sided test with a 0.3 deg margin of which the
 * planar-vs-geodesic bound uses 0.26), for which the exact vector of the tick before, rotated by the turn just made (<= 5 deg: cos / sin by their Taylor
 * polynomials, error < 1e-9 rad), is as good — and it takes the exact one back at Y for the next tick (launch test, next rotation). */
struct QSlimMail { int flags[64], w5[64], w6[64], w7[64]; double rew[64]; int full; };                       /* sim -> out at Y */
/* w7: aircraft type, alive, has_missile (8 bits each) | reward key << 24 | done << 25 | "the arena ran this tick" << 26 | alive bits of the arena
 * after the tick and BEFORE a reset (4 bits) << 27.  The reward crosses as the double the tick computed: the output wave also keeps the episode
 * statistics (return summed in agent order, length, outcome: what the logging all-gather moves), one more thing the next tick does not read. */
/* The output wave also runs AHEAD of the simulation wave: between X and Y of tick t it computes what tick t + 1 will need that depends only on
 * what tick t has already fixed — the keyed-RNG tick key of (episode, steps + 1), this lane's script draw from it in both variants (escaping /
 * pursuing: which one applies is decided in tick t + 1), and the rounded sine / cosine of the heading after the turn that _correct_angle_sign
 * (env_base.py:464-487) takes.  `ok` is wave-uniform: false on the first tick of a launch and after a tick that reset an arena of the wave
 * (episode / steps / heading changed behind the prediction); the tick then computes them itself, the same expressions. */
struct QPre {
    bool ok;
    bool spec;            /* wave-uniform: ok, and the tick before changed no alive mask in this wave — the prediction the output wave ran the level-3 script on */
    double sp_hdg, sp_spd; /* the script's commanded heading / speed for this lane's opponent */
    int sp_w;             /* fire | fire_m << 1 | (target slot + 1) << 2 | arena escape flag << 8 | escape timer << 16 (after the script's tick) */
    unsigned long long tkey;
    double sx, sy;
};
__device__ __forceinline__ void quad_publish_flags(const Unit &m, QPub &p) {
    int shot = m.burst > 0 || (m.ac_type == 1 && m.has_missile);
    p.flags = (m.alive ? FL_ALIVE : 0) | ((m.ac_type & 3) << 1) | (shot ? FL_SHOT : 0);
}
__device__ __forceinline__ void quad_publish(const DevCfg &c, const Unit &m, QPub &p) {
    quad_publish_motion(c, m, p);
    quad_publish_flags(m, p);
}

/* the pair table of pair_tables() in registers.  WAVE-UNIFORM control flow only.
 * DUAL (8 arenas in lanes 0..31, helpers in lanes 32..63): the helper of a lane receives its position and heading vector, the DPP
 * fetches then deliver the same neighbours to both halves, and the four acos chains of a lane are split two and two — the main lane
 * computes the focus towards slots +1 and +2, its helper the focus towards slot +3 and the heading difference — the same expressions
 * on the same operands, handed up afterwards: half the table's arithmetic per lane for 14 lane swaps. */
/* the entries of the others that only observation rows read (and the alive mask): WAVE-UNIFORM control flow only */
__device__ __forceinline__ void quad_tables_obs(const QPub &p, int s, QTab &t) {
#define HH_QFETCHO(K)                                                                 \
    t.nlat[K - 1] = q_rot_f<K>(p.nlat); t.nlon[K - 1] = q_rot_f<K>(p.nlon);               \
    t.nspd[K - 1] = q_rot_f<K>(p.nspd); t.nhdg[K - 1] = q_rot_f<K>(p.nhdg);               \
    t.fl[K - 1] = q_rot_i<K>(p.flags);
    HH_QFETCHO(1)
    HH_QFETCHO(2)
    HH_QFETCHO(3)
#undef HH_QFETCHO
    t.amask = ((p.flags & FL_ALIVE) << s) | ((t.fl[0] & FL_ALIVE) << ((s + 1) & 3)) | ((t.fl[1] & FL_ALIVE) << ((s + 2) & 3)) |
              ((t.fl[2] & FL_ALIVE) << ((s + 3) & 3));
}
/* OBS = false: the geometry only (positions, distances, focus angles, heading differences); the caller adds quad_tables_obs when it has the flags */
template <bool DUAL, bool OBS = true>
__device__ __forceinline__ void quad_tables(const Unit &m, const QPub &p, int s, bool helper, QTab &t) {
    double lat = m.lat, lon = m.lon, c1 = p.uc, s1 = p.us, n1 = p.un;
    if (DUAL) { lat = q_down_d(lat); lon = q_down_d(lon); c1 = q_down_d(c1); s1 = q_down_d(s1); n1 = q_down_d(n1); }
    double ouc[3], ous[3], oun[3];
#define HH_QFETCH(K)                                                                  \
    t.lat[K - 1] = q_rot_d<K>(lat); t.lon[K - 1] = q_rot_d<K>(lon);                       \
    ouc[K - 1] = q_rot_d<K>(c1); ous[K - 1] = q_rot_d<K>(s1); oun[K - 1] = q_rot_d<K>(n1);
    HH_QFETCH(1)
    HH_QFETCH(2)
    HH_QFETCH(3)
#undef HH_QFETCH
    /* heading difference (env_base.py:448-456): read only by agents about opponents (observation), so the four
     * agent-opponent pairs are split one per lane — (0,2) (1,3) (2,1) (3,0) — and handed over; the expression is
     * symmetric in its operands, so either end computes the same bits */
    const int kh = s < 2 ? 2 : (s == 2 ? 3 : 1);
    const double c2 = q_sel(ouc, kh), s2 = q_sel(ous, kh), n2 = q_sel(oun, kh);
    double hx;
    if (!DUAL) {
        /* planar distances: the pair (s, s+3) is the pair (s', s'+1) of lane s' = s+3, and dx*dx + dy*dy does not
         * change when both differences flip sign, so two square roots per lane cover the six pairs */
        {
            double dx0 = t.lon[0] - lon, dy0 = t.lat[0] - lat;
            double dx1 = t.lon[1] - lon, dy1 = t.lat[1] - lat;
            t.dist[0] = hh_sqrt(dx0 * dx0 + dy0 * dy0);
            t.dist[1] = hh_sqrt(dx1 * dx1 + dy1 * dy1);
            t.dist[2] = q_rot_d<3>(t.dist[0]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double dx = t.lon[k] - lon, dy = t.lat[k] - lat;
            double nn = t.dist[k];
            double dot = c1 * dx + s1 * dy;
            double x = hh_clip(dot / (n1 * nn + 1e-10), -1.0, 1.0);
            t.foc[k] = hh_acos(x) * (180.0 / HH_PI);
        }
        {
            double dot = c1 * c2 + s1 * s2;
            double x = hh_clip(dot / (n1 * n2 + 1e-10), -1.0, 1.0);
            hx = hh_clip(HH_DIVC(hh_acos(x) * (180.0 / HH_PI), 180.0), 0.0, 1.0);
        }
    } else {
        /* chain A: main = focus towards slot +1, helper = focus towards slot +3 (its distance is the same bits from either end) */
        const double dxA = (helper ? t.lon[2] : t.lon[0]) - lon, dyA = (helper ? t.lat[2] : t.lat[0]) - lat;
        const double nA = hh_sqrt(dxA * dxA + dyA * dyA);
        const double xA = hh_clip((c1 * dxA + s1 * dyA) / (n1 * nA + 1e-10), -1.0, 1.0);
        const double focA = hh_acos(xA) * (180.0 / HH_PI);
        /* chain B: main = focus towards slot +2, helper = the heading difference */
        const double dxB = t.lon[1] - lon, dyB = t.lat[1] - lat;
        const double nBm = hh_sqrt(dxB * dxB + dyB * dyB);
        const double dotBm = c1 * dxB + s1 * dyB, dotBh = c1 * c2 + s1 * s2;
        const double xB = hh_clip((helper ? dotBh : dotBm) / (n1 * (helper ? n2 : nBm) + 1e-10), -1.0, 1.0);
        const double angB = hh_acos(xB) * (180.0 / HH_PI);
        t.dist[0] = nA; t.dist[1] = nBm;
        t.dist[2] = q_rot_d<3>(t.dist[0]);
        t.foc[0] = focA; t.foc[1] = angB;
        t.foc[2] = q_up_d(focA);
        hx = q_up_d(hh_clip(HH_DIVC(angB, 180.0), 0.0, 1.0));
    }
    t.hd[0] = q_rot_d<1>(hx); /* used by slot 1 about slot 2 */
    t.hd[1] = hx;             /* agents: the opponent two slots up */
    t.hd[2] = q_rot_d<3>(hx); /* used by slot 0 about slot 3 */
    /* focus of slot s+k at me = that lane's entry for ITS relative slot 4-k */
    t.focr[0] = q_rot_d<1>(t.foc[2]);
    t.focr[1] = q_rot_d<2>(t.foc[1]);
    t.focr[2] = q_rot_d<3>(t.foc[0]);
    if constexpr (OBS) quad_tables_obs(p, s, t);
}

/* env_base.py:400-422 _nearby_object for the OTHER side of a 2-vs-2 arena: live opponents, stable-sorted by
 * normalised distance (ties keep id order) */
struct Near2 {
    int n, j0, k0, j1, k1;
    double d0, d1, r0, r1;
};
__device__ __forceinline__ void quad_nearby(const DevCfg &c, const QTab &t, int s, Near2 &o) {
    const int ja = s < 2 ? 2 : 0, jb = ja + 1;
    const int ka = (ja - s) & 3, kb = (jb - s) & 3;
    const int ala = (t.amask >> ja) & 1, alb = (t.amask >> jb) & 1;
    const double ra = q_sel(t.dist, ka), rb = q_sel(t.dist, kb);
    const double da = c.inv_diag * ra, db = c.inv_diag * rb;
    const bool a_first = ala && (!alb || da <= db);
    o.n = ala + alb;
    o.j0 = a_first ? ja : jb; o.k0 = a_first ? ka : kb; o.d0 = a_first ? da : db; o.r0 = a_first ? ra : rb;
    o.j1 = a_first ? jb : ja; o.k1 = a_first ? kb : ka; o.d1 = a_first ? db : da; o.r1 = a_first ? rb : ra;
    if (o.n == 0) { o.j0 = o.k0 = 0; o.d0 = o.r0 = 0.0; }
    if (o.n < 2) { o.j1 = o.k1 = 0; o.d1 = o.r1 = 0.0; }
}

/* env_base.py:185-212 opp_ac_values (mode 0 fight / 1 escape) from the register table */
__device__ __forceinline__ int quad_opp_block(const QTab &t, int mode, int k, double dist, float *out) {
    const double f_so = q_sel(t.foc, k), f_os = q_sel(t.focr, k);
    int n = 0;
    out[n++] = q_sel(t.nlat, k);
    out[n++] = q_sel(t.nlon, k);
    out[n++] = q_sel(t.nspd, k);
    out[n++] = q_sel(t.nhdg, k);
    out[n++] = (float)q_sel(t.hd, k);
    if (mode == 0) {
        out[n++] = (float)norm180(f_os);
        out[n++] = (float)aspect(f_so);
    } else {
        out[n++] = (float)norm180(f_so);
        out[n++] = (float)norm180(f_os);
    }
    out[n++] = (float)dist;
    out[n++] = (q_sel(t.fl, k) & FL_SHOT) ? 1.0f : 0.0f;
    return n;
}

/* env_hetero.py:65-103 lowlevel_state of the lane's own unit into its LDS staging row (D floats, zero padded);
 * refreshes opp_to_attack (m.tgt0) */
__device__ __forceinline__ void quad_lowlevel_obs(const DevCfg &c, const QTab &t, const QPub &p, int s, int mode, Unit &m, float *out, int D) {
    m.n_tgt = 0; m.tgt0 = 0; m.tgt_d0 = 0.0;
    Near2 nb;
    quad_nearby(c, t, s, nb);
    if (!m.alive || nb.n == 0) {
        for (int k = 0; k < D; k++) out[k] = 0.0f;
        return;
    }
    m.n_tgt = 1; m.tgt0 = nb.j0 + 1; m.tgt_d0 = nb.d0;
    /* zero padding: D is the AC1 row of the mode (26 fight / 30 escape) and an AC2 row is two / one entries shorter, so the padding is the row's last two
     * entries at most — zeroed HERE, unconditionally, and overwritten by the longer rows (a lane's LDS stores land in program order), instead of a
     * `for (; n < D; n++)` loop behind the row: a run-time trip count is an exec-mask loop with a taken branch per entry */
    out[D - 2] = 0.0f;
    out[D - 1] = 0.0f;
    int n = 0;
    out[n++] = p.nlat;
    out[n++] = p.nlon;
    out[n++] = p.nspd;
    out[n++] = p.nhdg;
    if (mode == HH_MODE_FIGHT) {
        out[n++] = (float)norm180(q_sel(t.foc, nb.k0));
        out[n++] = (float)aspect(q_sel(t.focr, nb.k0));
        out[n++] = (float)q_sel(t.hd, nb.k0);
        out[n++] = (float)nb.d0;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) {
            out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
            out[n++] = m.missile_wait == 0 ? 1.0f : 0.0f;
            out[n++] = (m.has_missile || m.burst > 0) ? 1.0f : 0.0f;
        } else {
            out[n++] = m.burst > 0 ? 1.0f : 0.0f;
        }
        n += quad_opp_block(t, 0, nb.k0, nb.d0, out + n);
    } else {
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
        out[n++] = (p.flags & FL_SHOT) ? 1.0f : 0.0f;
        quad_opp_block(t, 1, nb.k0, nb.d0, out + n);
        if (nb.n >= 2) quad_opp_block(t, 1, nb.k1, nb.d1, out + n + 9);
        else for (int k = 0; k < 9; k++) out[n + 9 + k] = 0.0f;
        n += 18;
    }
    /* env_base.py:166-183 friendly_ac_values; env_hetero.py:71-75: the friend is the other aircraft of the side */
    const int kf = ((s ^ 1) - s) & 3;
    if ((t.amask >> (s ^ 1)) & 1) {
        out[n++] = q_sel(t.nlat, kf);
        out[n++] = q_sel(t.nlon, kf);
        out[n++] = (float)norm180(q_sel(t.foc, kf));
        out[n++] = (float)norm180(q_sel(t.focr, kf));
        out[n++] = (float)(c.inv_diag * q_sel(t.dist, kf));
    } else {
        for (int k = 0; k < 5; k++) out[n++] = 0.0f;
    }
}

/* "does any lane of the wave want this?" as a scalar the compiler cannot fold back into the lanes' own test: a rare body then sits
 * behind a uniform branch (~30 cycles when nobody wants it) instead of an exec-mask region (~55) */
__device__ __forceinline__ bool q_any(bool x) {
    unsigned long long b = __ballot(x);
    asm volatile("" : "+s"(b));
    return b != 0ULL;
}

/* ordering point between LDS accesses of ONE wave (the LDS unit executes a wave's instructions in order, so this
 * only has to stop the compiler from moving them and to drain the counter); the two-wave kernel below cannot use
 * __syncthreads() inside the simulation wave — that would be a workgroup barrier the output wave does not take */
__device__ __forceinline__ void q_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

/* ---- the level-3 opponent script (env_hetero.py:138-158 + 227-271) as functions: the simulation wave runs them inside the tick; the output wave
 * of the two-wave forms runs the SAME functions a tick ahead on a prediction (QPre.spec) ---- */
/* the arena-level escape flag for the tick with counter `steps` and key `tkey`: consumed once per live opponent in id order.  esc / esc_t:
 * the arena's flag and timer, updated in place.  Call from code every lane of the arena takes (a ballot inside). */
__device__ __forceinline__ void quad_l3_flags(int steps, unsigned long long tkey, int amask0, int s, int &esc, int &esc_t, bool &my_escaping, bool (&escj)[2]) {
    const bool draw_tick = steps % 60 == 0; /* one tick in sixty per arena: the draws sit behind a wave-uniform test */
    const bool any_draw = q_any(draw_tick);
#pragma unroll
    for (int j = 2; j < 4; j++) {
        const bool aj = ((amask0 >> j) & 1) != 0;
        if (HH_RARE(any_draw)) { /* wave-uniform */
            if (aj & draw_tick & (esc == 0)) {
                esc = hh_rng_randint(hh_rng_u01(tkey, (uint32_t)(j + 1), HH_SITE_L3_ESC_COIN, 0u), 0, 1);
                if (esc) esc_t = (int)hh_rng_uniform(hh_rng_u01(tkey, (uint32_t)(j + 1), HH_SITE_L3_ESC_TIME, 0u), 20.0, 30.0);
            }
        }
        const bool mine = aj & (j == s), upd = aj & (esc != 0);
        my_escaping = mine ? (esc != 0) : my_escaping;
        escj[j - 2] = aj & (esc != 0);
        const int esc_t1 = esc_t - 1;
        esc_t = upd ? esc_t1 : esc_t;
        esc = (upd & (esc_t <= 0)) ? 0 : esc;
    }
}
/* what the script decides for one opponent; opp = target slot or -1.  rs / rc: round(sin(hdg), 3), round(cos(hdg), 3) (_correct_angle_sign) */
struct QScriptOut { double heading, speed; int fire, fire_m, opp; };
__device__ __forceinline__ void quad_l3_script(const DevCfg &c, double lat, double lon, double hdg, int ac_type, bool my_escaping, double u0, double u1, double u2,
                                               const Near2 &nb, const QTab &tb, double focus_k0, double rs, double rc, QScriptOut &o) {
    int opp = -1, fire = 0, fire_m = 0;
    double heading, speed;
    if (my_escaping) { /* env_hetero.py:227-245 _escaping_opp */
        double y = hh_clip(hh_div_known(lat - HH_MAP_LAT0, c.ext_lat, c.inv_ext_lat), 0.0, 1.0);
        double x = hh_clip(hh_div_known(lon - HH_MAP_LON0, c.ext_lon, c.inv_ext_lon), 0.0, 1.0);
        double uh = u0;
        double lo_h = y < 0.5 ? (x < 0.5 ? 30.0 : 300.0) : (x < 0.5 ? 120.0 : 210.0);
        heading = (double)(int)hh_rng_uniform(uh, lo_h, lo_h + 30.0);
        speed = (double)(int)hh_rng_uniform(u1, 300.0, 600.0);
        fire = hh_rng_randint(u2, 0, 1);
    } else { /* env_hetero.py:247-271 _hardcoded_opp */
        heading = hdg;
        speed = (double)(int)hh_rng_uniform(u0, 100.0, 400.0);
        if (nb.n) {
            const double ag_lat = q_sel(tb.lat, nb.k0), ag_lon = q_sel(tb.lon, nb.k0);
            /* env_base.py:464-487 _correct_angle_sign */
            double x1 = lon + rs, y1 = lat + rc;
            double val = (x1 - lon) * (ag_lat - lat) - (ag_lon - lon) * (y1 - lat);
            double sign = val < 0.0 ? 1.0 : -1.0;
            double r = hh_rng_uniform(u1, 0.7, 1.3);
            double focus = focus_k0; /* = q_sel(tb.foc, nb.k0): by value, the OWT simulation wave does not hold the table's focus entries */
            const double turned = hh_pymod360(heading + r * sign * focus);
            heading = ((nb.d0 > 0.008) & (focus > 4.0)) ? turned : heading;
            const double us = u2;
            const double sp_near = (double)(int)hh_rng_uniform(us, 500.0, 800.0), sp_far = (double)(int)hh_rng_uniform(us, 100.0, 500.0);
            const double sp2 = focus < 30.0 ? sp_near : sp_far;
            speed = nb.d0 > 0.05 ? sp2 : speed;
            fire = nb.d0 < 0.03 && focus < 10.0;
            fire_m = nb.d0 < 0.09 && focus < 5.0;
            opp = nb.j0;
        }
        if (ac_type == 2) speed = hh_clip(speed, 0.0, 600.0);
    }
    if (heading >= 360.0 || heading < 0.0) heading = 0.0;
    o.heading = heading; o.speed = speed; o.fire = fire; o.fire_m = fire_m; o.opp = opp;
}

/* one fused LowLevelEnv step of the lane's arena; `tb`/`pub` hold the pre-tick table on entry and the post-tick
 * table on return.  Line-by-line counterpart of tick<4, 64>(tmode 0) in hh_kernels.h. */
template <bool IX, bool DUAL, bool OWT = false>
__device__ __forceinline__ void tick_quad(const DevCfg &c, Shared<4, 64> &sh, int tid, int g, int s, int base, bool active, bool helper, Unit &m,
                                          Arena &ar, const int8_t *act, QTab &tb, QPub &pub, Near2 &nbc, const QTgt &tg, StepOut &out,
                                          uint32_t &ev_mask_out, QPosMail *pos, const QPre &pre HH_PROF_ARGS) {
    /* OWT: tb.dist / foc / focr are NOT valid in here — the entries of the lane's target travel in `tg` (QTgt above) */
    /* OWT: the post-tick pair table is built by the output wave (QPosMail above).  This function then posts the moved positions and meets the
     * output wave at barrier X after phase B, and returns with tb.lat / lon / amask refreshed but tb.dist / foc / focr and nbc still the PRE-tick
     * ones: the caller takes the new ones from the output wave at barrier Y. */
    /* nbc: _nearby_object of the lane against `tb` — the pre-tick table on entry (what the scripts and the target refresh of this tick
     * read), the post-tick table on return: computed once per tick, straight-line on every lane, instead of once per reader */
    constexpr int A = 4;
    const int id = s + 1;
    const bool running = active && !ar.done;
    const bool agent = s < 2;
    out.reward = 0.0;
    out.valid = 0;
    out.kill_event = 0;
    uint32_t evm = 0;
    if (OWT && HH_USUAL(pre.ok)) { /* wave-uniform: the key of this tick was computed by the output wave during the last one */
        if (running) { ar.steps += 1; ar.tkey = pre.tkey; }
    } else if (running) { ar.steps += 1; arena_rekey(ar); }
    const bool snap = running && m.alive;
    const int amask0 = tb.amask; /* alive at tick start, by absolute slot */
    /* rocket_unit.py:25-35 speed profile of this slot's rocket (a launch in this tick starts at age 0).  Looked up
     * here, a whole phase before its use, so the table read is off the critical path. */
    const int rk_age0 = (m.rk_alive && m.rk_life <= HH_ROCKET_MAX_LIFE) ? m.rk_life : 0;
    const double rk_speed0 = sh.rk_speed[rk_age0];
    const int tgt_at_act = m.n_tgt ? m.tgt0 : 0; /* an agent's opp_to_attack when it acts (its lane never changes it inside the tick) */
    int want_launch = 0, launch_tgt = 0;
    int wait_after = -1;
    bool base_gate = false;

    /* ---------------- phase A: commands (env_hetero.py:160-182) ---------------- */
    /* Control flow of the tick: at one wave per SIMD a region under the exec mask costs ~45 cycles entered and ~55 skipped
     * (tools/ubench/issue.hip) — more than a dozen instructions.  Short bodies are therefore written as selects on values computed
     * unconditionally (named locals first, so that the front end emits a select and not a branch), nested tests are merged into one
     * region, and rare bodies sit behind ONE wave-uniform ballot test.  Same expressions, same bits. */
    {
        if (HH_USUAL(snap & (agent | (c.ext_opp != 0)))) {
            int t = m.n_tgt ? m.tgt0 : 0;
            if (!agent) { /* env_base.py:349-398 _policy_actions -> lowlevel_state(opp_mode, i): refresh target */
                const Near2 nb = nbc;
                m.n_tgt = nb.n ? 1 : 0; m.tgt0 = nb.n ? nb.j0 + 1 : 0; m.tgt_d0 = nb.n ? nb.d0 : 0.0;
                t = m.tgt0;
            }
            out.valid = agent ? 1 : out.valid;
            /* env_hetero.py:169-170 opp_stats[i][0] (the target's focus on the agent when it acted) is read by ONE reward term, the cannon kill:
             * evaluated there (phase E), from the same pre-tick table and target */
            /* env_base.py:214-238 _take_base_action */
#ifdef HHQ_ABL_DECODE /* tuning builds only (tools/build_variant.sh): WRONG RESULTS on purpose, timing of what a piece costs */
            m.cmd_hdg = m.hdg; m.cmd_spd = 300.0;
#else
            double nh = hh_pymod360(m.hdg + (double)(((int)act[0] - 6) * 15));
            if (nh >= 360.0 || nh < 0.0) nh = 0.0;
            m.cmd_hdg = nh;
            double mx = HH_AC_MAX_SPEED(m.ac_type);
            m.cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
#endif
            if (HH_USUAL(act[2] && m.cannon_remain > 0)) {
                arm_cannon(m);
                if (agent && c.agent_mode == HH_MODE_ESCAPE && m.cannon_remain < 90) out.reward -= 0.1;
            }
            {
                const bool gate = (m.ac_type == 1) & (act[3] != 0) & (t != 0) & (m.missile_remain > 0) & (m.has_missile == 0) & (m.missile_wait == 0);
                base_gate = gate;
                want_launch = gate ? 1 : want_launch;
                launch_tgt = gate ? t - 1 : launch_tgt;
            }
        } else if (snap && c.level <= 2) { /* env_hetero.py:118-136 levels 1-2 */
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
                const Near2 nb = nbc;
                if (nb.n) { want_launch = 1; launch_tgt = nb.j0; wait_after = 5; }
            }
        }
    }
    HH_PROF(11);
    /* OWT: the whole level-3 script of this tick may have been run a tick ahead by the output wave, on the prediction that the last tick changed no
     * alive mask in this wave (QPre.spec, wave-uniform): then its decisions are applied below and the flag / draws / script blocks are skipped */
    const bool spec = OWT && pre.ok && pre.spec && !c.ext_opp && c.level >= 3;
    /* env_hetero.py:138-158 level 3: the arena-level escape flag, consumed once per live opponent in id order */
    bool my_escaping = false;
    bool escj[2] = {false, false}; /* the flag as opponent slot 2 / 3 consumes it (every lane of the arena computes both) */
    if (HH_RARE(!spec)) {
    if (HH_USUAL(running && !c.ext_opp && c.level >= 3)) {
        int esc = ar.escaping, esc_t = ar.escaping_time;
        quad_l3_flags(ar.steps, ar.tkey, amask0, s, esc, esc_t, my_escaping, escj);
        ar.escaping = esc;
        ar.escaping_time = esc_t;
    }
    double u0 = 0.0, u1 = 0.0, u2 = 0.0;
    if (!c.ext_opp && c.level >= 3) { /* wave-uniform (configuration): every lane, the helpers included, takes part in the exchange */
        /* The three uniform draws of an opponent's script (escaping: heading, speed, fire; otherwise: speed, turn factor, second speed)
         * are the same ~45 instructions with different site keys, and while the opponents' lanes run their script the agents' lanes
         * and the helper lanes idle: so ONE pass over the mixer computes all three side by side — draw 0 on the opponent's own lane,
         * draw 1 on the agent lane two slots below, draw 2 on the opponent's helper lane (small-world form; otherwise a second
         * pass) — and three lane moves hand them over.  Keyed draws: the values are those of the sequential form. */
        {
            const int du = (s | 2) + 1;                           /* unit id of the opponent this lane draws for */
            const bool desc = (s & 1) ? escj[1] : escj[0];
            int role = agent ? 1 : 0, de = desc ? 1 : 0;
            if (DUAL) {
                const int deh = q_down_i(de);
                de = helper ? deh : de;
                role = helper ? 2 : role;
            }
            uint64_t key = ar.tkey;
            if (DUAL) {
                const int klo = q_down_i((int)(uint32_t)key), khi = q_down_i((int)(uint32_t)(key >> 32));
                key = helper ? (((uint64_t)(uint32_t)khi << 32) | (uint64_t)(uint32_t)klo) : key;
            }
            const int site_e = role == 0 ? HH_SITE_ESC_HDG : (role == 1 ? HH_SITE_ESC_SPEED : HH_SITE_ESC_FIRE);
            const int site_h = role == 0 ? HH_SITE_HC_SPEED1 : (role == 1 ? HH_SITE_HC_R : HH_SITE_HC_SPEED2);
            const double u = hh_rng_u01(key, (uint32_t)du, (uint32_t)(de ? site_e : site_h), 0u);
            u0 = u;
            u1 = q_perm_d<HH_QP(0, 1, 0, 1)>(u); /* slot 2 <- slot 0, slot 3 <- slot 1 */
            if (DUAL) u2 = q_up_d(u);
            else u2 = hh_rng_u01(ar.tkey, (uint32_t)id, (uint32_t)(my_escaping ? HH_SITE_ESC_FIRE : HH_SITE_HC_SPEED2), 0u);
        }
    }
#ifdef HHQ_ABL_SCRIPT
    if (false) {
#else
    if (HH_USUAL(running && !c.ext_opp && c.level >= 3)) {
#endif
        if (HH_USUAL(snap && !agent)) {
            double rs, rc;
            if (OWT && pre.ok) { rs = pre.sx; rc = pre.sy; } /* wave-uniform */
            else {
                double sn, cs;
                hh_sincos(hh_pymod360(m.hdg) * (HH_PI / 180.0), &sn, &cs);
                rs = hh_round3(sn); rc = hh_round3(cs);
            }
            QScriptOut so_;
            quad_l3_script(c, m.lat, m.lon, m.hdg, m.ac_type, my_escaping, u0, u1, u2, nbc, tb, OWT ? tg.foc : q_sel(tb.foc, nbc.k0), rs, rc, so_);
            m.cmd_hdg = so_.heading;
            m.cmd_spd = so_.speed;
            if (so_.fire) arm_cannon(m);
            if (so_.fire_m && so_.opp >= 0 && !m.has_missile && m.missile_wait == 0 && m.ac_type == 1) {
                want_launch = 1; launch_tgt = so_.opp; wait_after = 10;
            }
        }
    }
    } else { /* spec: apply what the output wave decided (same functions, same operands) */
        if (HH_USUAL(running)) {
            ar.escaping = (pre.sp_w >> 8) & 0xff;
            ar.escaping_time = (int)(int8_t)((pre.sp_w >> 16) & 0xff);
            if (HH_USUAL(snap && !agent)) {
                const int opp = ((pre.sp_w >> 2) & 7) - 1;
                m.cmd_hdg = pre.sp_hdg;
                m.cmd_spd = pre.sp_spd;
                if (pre.sp_w & 1) arm_cannon(m);
                if ((pre.sp_w & 2) && opp >= 0 && !m.has_missile && m.missile_wait == 0 && m.ac_type == 1) {
                    want_launch = 1; launch_tgt = opp; wait_after = 10;
                }
            }
        }
    }

    HH_PROF(0);
    /* ---------------- phase B: aircraft kinematics + move (ac1.py:81-133) ---------------- */
    const double lat_old = m.lat, lon_old = m.lon, hdg_old = m.hdg;
    bool fired = false;
    const int rk_pre = m.rk_alive;
    const int has_missile_pre = m.has_missile;
    const bool try_launch = want_launch && !m.has_missile && m.missile_remain > 0; /* ac1.py:73 */
    double turn_deg = 0.0;
    if (HH_USUAL(snap)) {
        int t = m.ac_type;
        {
            const double delta = d_signed_heading_diff(m.hdg, m.cmd_hdg);
            const double max_deg = HH_AC_TURN_RATE(t) * 1.0;
            const double turn = delta >= 0.0 ? max_deg : -max_deg;
            const double stepped = hh_pymod360(m.hdg + turn);
            const bool reach = hh_fabs(delta) <= max_deg, moved = m.hdg != m.cmd_hdg;
            const double nh = reach ? m.cmd_hdg : stepped;
            turn_deg = moved ? (reach ? delta : turn) : 0.0; /* the turn actually made, signed [deg] (OWT: rotates the heading vector below) */
            m.hdg = moved ? nh : m.hdg;
        }
        {
            const double delta = m.cmd_spd - m.spd;
            const double max_delta = HH_AC_ACCEL(t) * 1.0;
            const double stepped = m.spd + (delta >= 0.0 ? max_delta : -max_delta);
            const double ns = hh_fabs(delta) <= max_delta ? m.cmd_spd : stepped;
            m.spd = m.spd != m.cmd_spd ? ns : m.spd;
        }
        if (HH_USUAL(m.burst > 0)) {
            fired = true;
            m.burst = m.burst - 1 > 0 ? m.burst - 1 : 0;
            m.cannon_remain = m.cannon_remain - 1 > 0 ? m.cannon_remain - 1 : 0;
        }
        { /* ac1.py:117-128, rocket launched in an earlier step */
            const bool steer = (m.has_missile != 0) & (m.rk_alive != 0);
            if (HH_USUAL(steer)) m.rk_cmd = hh_clip(m.rk_hdg * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
            m.has_missile = ((m.has_missile != 0) & (m.rk_alive == 0)) ? 0 : m.has_missile;
        }
    }
    /* aircraft move + speculative move of this slot's rocket (in flight, or the one a pending launch creates) */
    const bool rk_spec = running && (rk_pre ? m.rk_life <= HH_ROCKET_MAX_LIFE : try_launch);
    double rk_nlat = 0.0, rk_nlon = 0.0, rk_nhdg = 0.0, rk_ncmd = 0.0;
    {
        const bool mv_a = snap && m.spd > 0.0;
        const bool any_rk = __ballot(rk_spec) != 0ULL;
        double r_lat = 5.0, r_lon = 7.0, r_hdg = 0.0;
        if (HH_USUAL(any_rk)) {
            r_lat = rk_pre ? m.rk_lat : lat_old; r_lon = rk_pre ? m.rk_lon : lon_old;
            r_hdg = rk_pre ? m.rk_hdg : hdg_old;
            rk_ncmd = m.rk_cmd;
            if (HH_RARE(q_any(!rk_pre & rk_spec))) /* a launch in this tick: rare */
                if (!rk_pre) rk_ncmd = hh_clip(hdg_old * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
            {
                const double delta = d_signed_heading_diff(r_hdg, rk_ncmd);
                const double stepped = r_hdg + (delta >= 0.0 ? HH_ROCKET_TURN_RATE : -HH_ROCKET_TURN_RATE);
                const double nh = hh_fabs(delta) <= HH_ROCKET_TURN_RATE ? rk_ncmd : stepped;
                r_hdg = r_hdg != rk_ncmd ? nh : r_hdg;
            }
            rk_nhdg = r_hdg;
        }
        if (DUAL) {
            /* ONE chain per lane: the main lane moves the aircraft, its helper the rocket (d_geo_move2's two interleaved chains give
             * the same bits as d_geo_move of each argument set) */
            double x_lat = m.lat, x_lon = m.lon, x_hdg = m.hdg, x_s = mv_a ? m.spd * HH_KNOTS_TO_MS * 1.0 : 0.0;
            if (HH_USUAL(any_rk)) {
                const double h_lat = q_down_d(rk_spec ? r_lat : 5.0), h_lon = q_down_d(rk_spec ? r_lon : 7.0);
                const double h_hdg = q_down_d(r_hdg), h_s = q_down_d(rk_speed0 * HH_KNOTS_TO_MS * 1.0);
                x_lat = helper ? h_lat : x_lat; x_lon = helper ? h_lon : x_lon; x_hdg = helper ? h_hdg : x_hdg; x_s = helper ? h_s : x_s;
            }
            double o_lat, o_lon;
#ifdef HHQ_ABL_MOVE
            o_lat = x_lat + 1e-5 * x_s; o_lon = x_lon + 1e-5 * x_hdg;
#else
            d_geo_move(x_lat, x_lon, x_hdg, x_s, o_lat, o_lon);
#endif
            if (mv_a) { m.lat = o_lat; m.lon = o_lon; }
            if (HH_USUAL(any_rk)) { rk_nlat = q_up_d(o_lat); rk_nlon = q_up_d(o_lon); }
        } else if (HH_USUAL(any_rk)) {
            const double r_spd = rk_speed0;
            double a_lat, a_lon;
            d_geo_move2(m.lat, m.lon, m.hdg, mv_a ? m.spd * HH_KNOTS_TO_MS * 1.0 : 0.0, a_lat, a_lon,
                        rk_spec ? r_lat : 5.0, rk_spec ? r_lon : 7.0, r_hdg, r_spd * HH_KNOTS_TO_MS * 1.0, rk_nlat, rk_nlon);
            if (mv_a) { m.lat = a_lat; m.lon = a_lon; }
        } else {
            if (mv_a) d_geo_move(m.lat, m.lon, m.hdg, m.spd * HH_KNOTS_TO_MS * 1.0, m.lat, m.lon);
        }
    }
    const double rk0_lat = rk_pre ? m.rk_lat : lat_old, rk0_lon = rk_pre ? m.rk_lon : lon_old;
    /* position, speed and heading are final for this tick: their published form is computed here, where it also
     * serves the cannon prefilter (heading vector after the turn) and overlaps the envelope phases */
    QPub pn;
    if constexpr (OWT) {
        { /* the exact heading vector of the tick before (pub), rotated by the turn just made: QTabMail's comment */
            const double th = turn_deg * (HH_PI / 180.0), t2 = th * th;
            const double cd = 1.0 + t2 * (-0.5 + t2 * (1.0 / 24.0));
            const double sd = th * (1.0 + t2 * ((-1.0 / 6.0) + t2 * (1.0 / 120.0)));
            pn.uc = pub.uc * cd + pub.us * sd; /* east  = sin(h + d) */
            pn.us = pub.us * cd - pub.uc * sd; /* north = cos(h + d) */
            pn.un = 1.0;
        }
        pn.nlat = pn.nlon = pn.nspd = pn.nhdg = 0.0f; /* formatted by the output wave from the raw values */
        /* every lane posts (the helpers' slots are never read): no exec-mask region */
        pos->lat[tid] = m.lat;
        pos->lon[tid] = m.lon;
        pos->spd[tid] = m.spd; pos->hdg[tid] = m.hdg; pos->ac_type[tid] = m.ac_type;
        pos->steps[tid] = ar.steps; pos->episode[tid] = ar.episode;
        pos->escw[tid] = (ar.escaping & 0xff) | ((ar.escaping_time & 0xff) << 8) | ((amask0 & 0xf) << 16);
        HH_PROF(1);
        __syncthreads(); /* barrier X: the output wave starts on the post-tick table */
        HH_PROF(12);
    } else {
        quad_publish_motion(c, m, pn);
    }
    /* positions of the other aircraft after their move (registers, by relative slot) */
    double lat1[3], lon1[3];
    lat1[0] = q_rot_d<1>(m.lat); lon1[0] = q_rot_d<1>(m.lon);
    lat1[1] = q_rot_d<2>(m.lat); lon1[1] = q_rot_d<2>(m.lon);
    lat1[2] = q_rot_d<3>(m.lat); lon1[2] = q_rot_d<3>(m.lon);

    HH_PROF(1);
    /* ---------------- phase Q: envelope tests that survive the prefilter -> workgroup queue (ballot-assigned slots) ---------------- */
    const int rk_tgt = rk_pre ? m.rk_target - 1 : launch_tgt;
    const bool rk_maybe = running && (rk_pre || try_launch);
    /* missile launch (ac1.py:72-79,135-146): launcher and target are tested where they stood before the tick, which
     * is the geometry of the pair table — its planar focus angle decides all but the ~1 % of launches within half a
     * degree of the cone's edges (hh_envelope.h); only those go through the queue */
    int launch_pre = -1;
    if (q_any(try_launch)) { /* wave-uniform; the stage is straight-line (hh_envelope.h), evaluated on every lane and kept by the launching ones */
        const int kl = (launch_tgt - s) & 3;
        const double t_lat = q_sel(tb.lat, kl), t_lon = q_sel(tb.lon, kl);
        const double cross = pub.uc * (t_lat - lat_old) - pub.us * (t_lon - lon_old);
        const int lp = hh_missile_cone_planar(lat_old, lon_old, t_lat, t_lon, OWT ? tg.foc : q_sel(tb.foc, kl), cross, OWT ? tg.dist : q_sel(tb.dist, kl));
        launch_pre = try_launch ? lp : -1;
    }
    int q_total = 0;
    {
        int push[6];
        int code[6];
        push[0] = try_launch && launch_pre < 0;
        code[0] = tid | (0 << 8) | (launch_tgt << 10);
        const int t = m.ac_type;
#pragma unroll
        for (int k = 1; k < A; k++) {
            const int j = (s + k) & 3;
            const bool snap_j = running && ((amask0 >> j) & 1); /* not alive at tick start -> can never be "currently alive" */
            const bool enemy = (j >= 2) != (s >= 2);
            /* target already moved iff its id is lower (cmano_simulator.py:142) */
            const double tl = j < s ? lat1[k - 1] : tb.lat[k - 1];
            const double to = j < s ? lon1[k - 1] : tb.lon[k - 1];
            /* every clause is a handful of compares and products and almost every tick SOME lane of the wave has fired: evaluated
             * unconditionally (&, not &&) — at one wave per SIMD an exec-mask region costs ~45 cycles whether or not it is entered */
#ifdef HHQ_ABL_PREFILTER
            push[k] = 0;
#else
            push[k] = (int)fired & (int)snap_j & ((int)(c.friendly_kill != 0) | (int)enemy) & (int)d_maybe_within_km(lat_old, lon_old, tl, to, HH_AC_CANNON_KM(t)) &
                      (int)!hh_cannon_cone_planar_outside(lat_old, lon_old, tl, to, pn.uc, pn.us, t);
#endif
            code[k] = tid | (1 << 8) | (j << 10);
        }
        {
            const int kt = (rk_tgt - s) & 3;
            const double tl = kt ? q_sel(lat1, kt) : m.lat, to = kt ? q_sel(lon1, kt) : m.lon;
            push[4] = (int)rk_maybe & (int)d_maybe_within_km(rk0_lat, rk0_lon, tl, to, HH_ROCKET_FUSE_KM);
            code[4] = tid | (2 << 8) | (rk_tgt << 10);
            const int fid = s == 1 ? 0 : 1; /* rocket_unit.py:46: 1 if source.id == 2 else 2 */
            const int kf = (fid - s) & 3;
            const double fl_ = kf ? q_sel(lat1, kf) : m.lat, fo_ = kf ? q_sel(lon1, kf) : m.lon;
            push[5] = (int)rk_maybe & (int)(c.friendly_kill != 0) & (int)d_maybe_within_km(rk0_lat, rk0_lon, fl_, fo_, HH_ROCKET_FUSE_KM);
            code[5] = tid | (3 << 8) | (fid << 10);
        }
        /* one wave-uniform test for the whole group (87 % of the wave-ticks queue nothing), then slots by ballot prefix */
        if (HH_RARE(__ballot(push[0] | push[1] | push[2] | push[3] | push[4] | push[5]) != 0ULL)) {
#pragma unroll
            for (int e = 0; e < 6; e++) {
                const unsigned long long bm = __ballot(push[e]);
                if (push[e]) {
                    int pos = q_total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
                    sh.u.t.q_code[pos] = code[e];
                }
                q_total += __popcll(bm);
#ifdef HH_PROFILE_PHASES
                if (tid == 0 && bm) atomicAdd(&hh_prof_cycles[e == 0 ? 13 : (e < 4 ? 14 : 15)], (unsigned long long)__popcll(bm));
#endif
            }
        }
#ifdef HH_PROFILE_PHASES
        if (tid == 0 && q_total) atomicAdd(&hh_prof_cycles[12], 1ULL);
#endif
    }
    HH_PROF(2);
    /* ---------------- phase I: dense pass over the queue ---------------- */
    int myres = 0;
    if (HH_RARE(q_total != 0)) { /* wave-uniform */
        /* what the dense pass reads about the requesting and the target aircraft */
        sh.lat0[tid] = lat_old; sh.lon0[tid] = lon_old; sh.hdg[tid] = hdg_old;
        sh.flags[tid] = pub.flags;
        sh.u.t.lat1[tid] = m.lat; sh.u.t.lon1[tid] = m.lon; sh.u.t.hdg1[tid] = m.hdg;
        sh.u.t.rk_lat[tid] = rk0_lat; sh.u.t.rk_lon[tid] = rk0_lon;
        sh.res[tid] = 0;
        if (s == 0 && !helper) sh.g_tkey[g] = ar.tkey;
        q_wave_sync();
        drain_envelope_queue<4, 64, IX>(sh, tid, q_total);
        q_wave_sync();
        myres = sh.res[tid];
    }
    if (launch_pre == 1) myres |= 1;

    HH_PROF(3);
    /* ---------------- phase L: launch bookkeeping (env_base.py:227-236, ac1.py:76-79) ---------------- */
    int launched = 0;
    const bool launch_now = try_launch & ((myres & 1) != 0);
    if (q_any(launch_now)) if (launch_now) {
        launched = 1;
        m.rk_alive = 1; m.rk_lat = lat_old; m.rk_lon = lon_old; m.rk_hdg = hdg_old;
        m.rk_target = launch_tgt + 1; m.rk_life = 0;
        m.has_missile = 1;
        m.missile_remain = m.missile_remain - 1 > 0 ? m.missile_remain - 1 : 0;
        evm |= 1u << (24 + s);
        m.rk_cmd = rk_ncmd; /* the launcher's own update in this tick already steers it (ac1.py:127) */
    }
    if (q_any(base_gate)) if (base_gate) {
        double uu = d_rng(ar, id, HH_SITE_MISSILE_WAIT, 0);
        m.missile_wait = hh_rng_randint(uu, 7, 17);
        if (agent && c.agent_mode == HH_MODE_ESCAPE && m.missile_remain < 3) out.reward -= 0.1;
    }
    { /* env_base.py:235-236, evaluated before do_tick */
        const bool dec = snap & (agent | (c.ext_opp != 0)) & (m.missile_wait > 0) & !(launched | has_missile_pre);
        const int w1 = m.missile_wait - 1;
        m.missile_wait = dec ? w1 : m.missile_wait;
    }
    m.missile_wait = (want_launch & (wait_after >= 0)) ? wait_after : m.missile_wait;
    const int rk_at_start = m.rk_alive;
    /* launch order = unit id order (cmano_simulator.py:104-108) */
    const int aux = launched | ((fired ? (myres >> 1) & 0xff : 0) << 8);
    int aux_[A];
    aux_[0] = q_bc_i<0>(aux); aux_[1] = q_bc_i<1>(aux); aux_[2] = q_bc_i<2>(aux); aux_[3] = q_bc_i<3>(aux);
    {
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            int l = aux_[j] & 1;
            total += l;
            if (j < s) before += l;
        }
        if (launched) m.rk_seq = ar.next_seq + before + 1;
        ar.next_seq += total;
    }
    int rkw = 0; /* bit0 present, bit1 fuse on target, bit2 fuse on "friendly", bit3 end of life, bits4-6 target, bits 8.. seq */
    {
        const int eol = m.rk_life > HH_ROCKET_MAX_LIFE;
        const int w = 1 | (((myres >> 9) & 1) << 1) | (((myres >> 10) & 1) << 2) | (eol << 3) | ((m.rk_target - 1) << 4) | (m.rk_seq << 8);
        rkw = (running & (rk_at_start != 0)) ? w : 0;
    }
    int res_[A];
    res_[0] = q_bc_i<0>(rkw); res_[1] = q_bc_i<1>(rkw); res_[2] = q_bc_i<2>(rkw); res_[3] = q_bc_i<3>(rkw);

    /* ---------------- phases C + D: id-ordered resolution, computed identically by the four lanes ---------------- */
    int alive = amask0, nev = 0, dead = 0;
    int evpack = 0; /* 5 bits per event: killer slot | victim slot << 2 | by rocket << 4 */
    { /* (an arena that is not running has no shot and no rocket word: nothing below changes anything for it) */
        /* aircraft phase: shooter i (alive at tick start, even if killed earlier in this tick) hits the still-alive
         * targets in id order (ac1.py:106-115).  Most ticks nobody in the WAVE has a hit to apply. */
        const bool any_hit = ((aux_[0] | aux_[1] | aux_[2] | aux_[3]) >> 8) != 0;
        if (HH_RARE(q_any(any_hit))) if (any_hit) {
#pragma unroll
            for (int i = 0; i < A; i++) {
                const int ci = aux_[i] >> 8;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    if (((ci >> j) & 1) && ((alive >> j) & 1)) {
                        alive &= ~(1 << j);
                        evpack |= (i | (j << 2)) << (5 * nev);
                        nev++;
                    }
                }
            }
        }
        /* rocket phase in launch order (rocket_unit.py:37-58).  A rocket whose word has no fuse / end-of-life bit
         * does nothing, so only the others are visited; with a single one the order is moot. */
        int nact = 0, w1 = 0, b1 = 0;
#pragma unroll
        for (int j = 0; j < A; j++) {
            if ((res_[j] & 1) && (res_[j] & 0xe)) { nact++; w1 = res_[j]; b1 = j; }
        }
        if (HH_RARE(q_any(nact > 0))) {
        if (nact == 1) {
            const int tg = (w1 >> 4) & 7;
            const int fid = b1 == 1 ? 0 : 1;
            if (((w1 >> 1) & 1) && ((alive >> tg) & 1)) {
                alive &= ~(1 << tg); dead |= 1 << b1;
                evpack |= (b1 | (tg << 2) | (1 << 4)) << (5 * nev);
                nev++;
            } else if (c.friendly_kill && ((alive >> fid) & 1) && ((w1 >> 2) & 1)) {
                alive &= ~(1 << fid); dead |= 1 << b1;
                evpack |= (b1 | (fid << 2) | (1 << 4)) << (5 * nev);
                nev++;
            } else if ((w1 >> 3) & 1) {
                dead |= 1 << b1;
            }
        } else if (nact > 1) {
            int done_mask = 0;
#pragma unroll
            for (int k = 0; k < A; k++) {
                int best = -1, best_seq = 0x7fffffff, w = 0;
#pragma unroll
                for (int j = 0; j < A; j++) {
                    const int wj = res_[j];
                    if ((wj & 1) && !((done_mask >> j) & 1) && (wj >> 8) < best_seq) { best = j; best_seq = wj >> 8; w = wj; }
                }
                if (best >= 0) {
                    done_mask |= 1 << best;
                    const int tg = (w >> 4) & 7;
                    const int fid = best == 1 ? 0 : 1;
                    if (((w >> 1) & 1) && ((alive >> tg) & 1)) {
                        alive &= ~(1 << tg); dead |= 1 << best;
                        evpack |= (best | (tg << 2) | (1 << 4)) << (5 * nev);
                        nev++;
                    } else if (c.friendly_kill && ((alive >> fid) & 1) && ((w >> 2) & 1)) {
                        alive &= ~(1 << fid); dead |= 1 << best;
                        evpack |= (best | (fid << 2) | (1 << 4)) << (5 * nev);
                        nev++;
                    } else if ((w >> 3) & 1) {
                        dead |= 1 << best;
                    }
                }
            }
        }
        }
    }
    if (HH_USUAL(running && rk_at_start)) { /* (usually some rocket of the wave is in flight: no any-lane test in front) */
        if ((dead >> s) & 1) {
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
    {
        const int al = (alive >> s) & 1;
        const bool inb = (HH_MAP_LON0 <= m.lon) & (m.lon <= c.lon_hi) & (HH_MAP_LAT0 <= m.lat) & (m.lat <= c.lat_hi);
        oob = (active & running & (al != 0) & !inb) ? 1 : 0;
        m.alive = active ? (oob ? 0 : al) : m.alive;
    }
    const int oobm = (int)(__ballot(oob) >> base) & 0xf; /* out-of-bounds removals of the arena */
    double rews = 0.0;
    int destroyed = 0;
    /* kills and removals are rare: rewards and event masks behind one wave-uniform test (nothing below does anything without one) */
    if (HH_RARE(q_any((nev > 0) | (oob != 0)))) {
    if (running && agent) {
        const double sc = c.rew_scale;
        if (oob) { rews += -5.0 * sc; destroyed = 1; }
        for (int e = 0; e < nev; e++) {
            const int w = evpack >> (5 * e);
            const int k = w & 3, d = (w >> 2) & 3, rocket = (w >> 4) & 1;
            if (k < 2) {
                if (d >= 2) {
                    if (k == s && c.agent_mode == HH_MODE_FIGHT) {
                        if (rocket) {
                            rews += (1.0 + ((1.5 - 1.0) / (1.0 - 0.0)) * ((double)m.missile_remain / (double)m.rocket_max - 0.0)) * sc;
                        } else {
                            double r1 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * ((double)m.cannon_remain / (double)m.cannon_max - 0.0);
                            /* the killer acted in this tick (it was alive at tick start), `tb` still holds the pre-tick table here */
                            const int t1 = tgt_at_act ? tgt_at_act - 1 : 0;
                            const bool os_ok = (tgt_at_act != 0) & (((amask0 >> t1) & 1) != 0);
                            const double opp_stat0 = os_ok ? norm180(OWT ? tg.focr : q_sel(tb.focr, (t1 - s) & 3)) : 0.0;
                            double r2 = 0.5 + ((1.0 - 0.5) / (1.0 - 0.0)) * (opp_stat0 - 0.0);
                            rews += (r1 + r2) * sc;
                        }
                    }
                } else {
                    if (k == s) rews += -2.0 * sc;
                    if (c.friendly_punish && d == s) { rews += -2.0 * sc; destroyed = 1; }
                }
            } else if (d < 2) {
                if (d == s) { rews += -2.0 * sc; destroyed = 1; }
            }
        }
    }
    for (int e = 0; e < nev; e++) { /* event masks for parity checks */
        const int w = evpack >> (5 * e);
        evm |= ((w >> 4) & 1) ? (1u << (8 + ((w >> 2) & 3))) : (1u << ((w >> 2) & 3));
    }
    if (oob) evm |= 1u << (16 + s);
    }
    ev_mask_out = evm;
    const double mate_rews = q_mate_d(rews);
    HH_PROF(5);
    /* post-tick table: escape shaping now, observation next, pre-step lookups of the next tick */
    quad_publish_flags(m, pn);
    pub = pn;
    HH_PROF(6);
    if constexpr (OWT) {
        /* what of the post-tick table this wave can tell by itself: where the others stand now and who is alive */
#pragma unroll
        for (int k = 0; k < 3; k++) { tb.lat[k] = lat1[k]; tb.lon[k] = lon1[k]; }
        tb.amask = (int)(__ballot(m.alive != 0) >> base) & 0xf;
    } else {
        quad_tables<DUAL>(m, pub, s, helper, tb);
        quad_nearby(c, tb, s, nbc);
    }
    HH_PROF(7);
    {
        const int ag = __popc(tb.amask & 3), op = __popc(tb.amask & 12);
        const int dn = ((ag <= 0) | (op <= 0) | (ar.steps >= c.horizon)) ? 1 : 0;
        const int ke = ((nev > 0) | (oobm != 0)) ? 1 : 0;
        out.kill_event = running ? ke : out.kill_event;
        ar.done = running ? dn : ar.done;
    }
    if (!OWT && c.agent_mode == HH_MODE_ESCAPE && c.esc_dist_rew) { /* wave-uniform: configuration (OWT instances are presets: no escape shaping) */
        if (running && agent && m.alive) { /* env_hetero.py:198-214 */
            const Near2 nb = nbc;
            const double dr[2] = {nb.r0, nb.r1};
#pragma unroll
            for (int j = 1; j <= 2; j++) {
                if (j <= nb.n) {
                    if (dr[j - 1] < 0.06) { rews += -0.02 / j; if (m.spd < 200.0) rews += -0.02 / j; }
                    else if (dr[j - 1] > 0.13) { rews += 0.02 / j; if (m.spd > 500.0) rews += 0.02 / j; }
                }
            }
        }
    }
    {
        const double shared = rews + c.glob_frac * mate_rews;
        const double add = (c.glob_frac > 0.0 && c.agent_mode == HH_MODE_FIGHT) ? shared : rews;
        const double nr = out.reward + add;
        const bool give = running & agent & ((m.alive != 0) | (destroyed != 0));
        out.reward = give ? nr : out.reward;
    }
    HH_PROF(8);
}

/* ROLLOUT of 2-vs-2 worlds: T fused steps per launch, one wave (16 arenas) per workgroup */
/* ---- two-wave form for small worlds: a simulation wave and an output wave per 16 arenas ----
 * Below ~16k arenas there are fewer waves than SIMDs and a tick costs the latency of one wave's dependent chains.
 * Formatting and storing the observation / reward / done rows (16 % of that) does not feed the next tick, so a second
 * wave of the workgroup — resident on another SIMD of the CU that would otherwise idle — takes it over: the simulation
 * wave posts the agents' table entries into a double-buffered LDS mailbox, meets the output wave at one workgroup
 * barrier per tick, and goes on with the next tick while the rows are built and written. */
struct ObsMail {
    double d[12][32];  /* dist[3] foc[3] focr[3] hd[3] of the agent's table, [field][agent row] */
    float f[16][32];   /* others' nlat nlon nspd nhdg [3] each, own nlat nlon nspd nhdg */
    int i[8][32];      /* fl[3], amask, own flags, 3 packed words of unit state / reward key / done */
    float rew[32];
};

__device__ __forceinline__ void mail_post(ObsMail &mb, int r, const QTab &t, const QPub &p, const Unit &m, const StepOut &so, int done) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mb.d[k][r] = t.dist[k]; mb.d[3 + k][r] = t.foc[k]; mb.d[6 + k][r] = t.focr[k]; mb.d[9 + k][r] = t.hd[k];
        mb.f[k][r] = t.nlat[k]; mb.f[3 + k][r] = t.nlon[k]; mb.f[6 + k][r] = t.nspd[k]; mb.f[9 + k][r] = t.nhdg[k];
        mb.i[k][r] = t.fl[k];
    }
    mb.f[12][r] = p.nlat; mb.f[13][r] = p.nlon; mb.f[14][r] = p.nspd; mb.f[15][r] = p.nhdg;
    mb.i[3][r] = t.amask;
    mb.i[4][r] = p.flags;
    mb.i[5][r] = (m.cannon_remain & 0xffff) | ((m.cannon_max & 0xffff) << 16);
    mb.i[6][r] = (m.missile_remain & 0xff) | ((m.rocket_max & 0xff) << 8) | ((m.missile_wait & 0xff) << 16) | ((m.burst & 0xff) << 24);
    mb.i[7][r] = (m.ac_type & 0xff) | ((m.alive & 0xff) << 8) | ((m.has_missile & 0xff) << 16) | ((so.valid & 1) << 24) | ((done & 1) << 25);
    mb.rew[r] = (float)so.reward;
}

__device__ __forceinline__ void mail_take(const ObsMail &mb, int r, QTab &t, QPub &p, Unit &m, int &valid, int &done, float &rew) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        t.dist[k] = mb.d[k][r]; t.foc[k] = mb.d[3 + k][r]; t.focr[k] = mb.d[6 + k][r]; t.hd[k] = mb.d[9 + k][r];
        t.nlat[k] = mb.f[k][r]; t.nlon[k] = mb.f[3 + k][r]; t.nspd[k] = mb.f[6 + k][r]; t.nhdg[k] = mb.f[9 + k][r];
        t.fl[k] = mb.i[k][r];
        t.lat[k] = t.lon[k] = 0.0;
    }
    p.nlat = mb.f[12][r]; p.nlon = mb.f[13][r]; p.nspd = mb.f[14][r]; p.nhdg = mb.f[15][r];
    p.uc = p.us = p.un = 0.0;
    t.amask = mb.i[3][r];
    p.flags = mb.i[4][r];
    const int w5 = mb.i[5][r], w6 = mb.i[6][r], w7 = mb.i[7][r];
    m = Unit{};
    m.cannon_remain = w5 & 0xffff; m.cannon_max = (w5 >> 16) & 0xffff;
    m.missile_remain = w6 & 0xff; m.rocket_max = (w6 >> 8) & 0xff; m.missile_wait = (w6 >> 16) & 0xff; m.burst = (w6 >> 24) & 0xff;
    m.ac_type = w7 & 0xff; m.alive = (w7 >> 8) & 0xff; m.has_missile = (w7 >> 16) & 0xff;
    valid = (w7 >> 24) & 1;
    done = (w7 >> 25) & 1;
    rew = mb.rew[r];
}

/* a workgroup's observation rows of one tick, staged in LDS (`tile`: GPB arenas x 2 agents x D floats, contiguous) -> [T, N, 2, D] with unit-stride 16-byte
 * stores.  The usual workgroup is full and its rows 16-byte aligned: the copy then has a compile-time trip count wherever D is one (the preset
 * instances) — three or four predicated stores instead of a run-time loop with a taken branch per round. */
template <int GPB>
__device__ __forceinline__ void quad_store_rows(float *__restrict__ obs_out, const float *tile, int t, int N, int D, int tid) {
    const int rows = min(GPB, N - (int)blockIdx.x * GPB);
    float *dst = obs_out + ((size_t)t * N + (size_t)blockIdx.x * GPB) * 2 * D;
    const int full = GPB * 2 * D;
    if (HH_USUAL(rows == GPB && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (full & 3) == 0)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        const int cnt4 = full >> 2;
#pragma unroll
        for (int k0 = 0; k0 < cnt4; k0 += 64) {
            const int k = k0 + tid;
            if (k < cnt4) dst4[k] = src4[k];
        }
    } else {
        const int cnt = rows * 2 * D;
        for (int k = tid; k < cnt; k += 64) dst[k] = tile[k];
    }
}

template <bool TWO> struct QuadMailbox { /* LDS of the two-wave form only */
    ObsMail mail[2];
    alignas(16) float tile[16 * 2 * HH_OBS_ESC_AC1];
    QPosMail pos;   /* the three mailboxes of the preset instances (pair table on the output wave) */
    QTabMail tab;
    QSlimMail slim;
};
template <> struct QuadMailbox<false> {};

/* env_hetero.py:99-101: the observation also refreshes opp_to_attack; the simulation wave needs only that part */
__device__ __forceinline__ void quad_target_refresh(const Near2 &nb, Unit &m) {
    const bool has = (m.alive != 0) & (nb.n != 0);
    const int j1 = nb.j0 + 1;
    m.n_tgt = has ? 1 : 0; m.tgt0 = has ? j1 : 0; m.tgt_d0 = has ? nb.d0 : 0.0;
}

/* PRE = preset: the reference's default training configuration (config.py:17-54: scripted opponents, friendly fire on, no friendly
 * punishment, no escape shaping, glob_frac 0, rew_scale 1) of one curriculum stage compiled with those values as constants — 1: level 3
 * fight (the stage the benchmark is quoted on), 2: level 1 fight, 3: level 2 fight, 4: level 3 escape.  The other configurations'
 * code (the other levels' opponent scripts, the other mode's observation and rewards) and its scalar registers drop out.  Every
 * other configuration runs the PRE = 0 instance of the same source; all give the same results. */
__host__ __device__ inline void hh_cfg_set_preset(DevCfg &c, int pre) {
    c.level = pre == 2 ? 1 : (pre == 3 ? 2 : 3);
    c.agent_mode = pre == 4 ? HH_MODE_ESCAPE : HH_MODE_FIGHT;
    c.ext_opp = 0; c.friendly_kill = 1; c.friendly_punish = 0; c.esc_dist_rew = 0; c.glob_frac = 0.0; c.rew_scale = 1.0;
    c.D = pre == 4 ? HH_OBS_ESC_AC1 : HH_OBS_FIGHT_AC1; c.n_ctrl = 2; c.nA = 2; c.nO = 2;
}
inline int hh_cfg_preset(const DevCfg &c) { /* which preset equals this configuration in every field it fixes (0: none) */
    for (int pre = 1; pre <= 4; pre++) {
        DevCfg d = c;
        hh_cfg_set_preset(d, pre);
        if (d.level == c.level && d.agent_mode == c.agent_mode && d.ext_opp == c.ext_opp && d.friendly_kill == c.friendly_kill &&
            d.friendly_punish == c.friendly_punish && d.esc_dist_rew == c.esc_dist_rew && d.glob_frac == c.glob_frac && d.rew_scale == c.rew_scale &&
            d.D == c.D && d.n_ctrl == c.n_ctrl && d.nA == c.nA && d.nO == c.nO)
            return pre;
    }
    return 0;
}
/* APW = arenas per simulation wave: 16 fills the 64 lanes; 8 (lanes 32..63 idle) is for worlds so small that half the SIMDs would
 * otherwise sit empty: a wave's tick costs the instructions of every branch ANY of its arenas takes (rocket in flight, cannon
 * burst, events, reset ...), so half the arenas per wave means fewer instructions per wave-tick at the same number of ticks. */
/* SHAPE = false: a general (PRE = 0) two-wave instance for configurations WITHOUT the escape distance shaping (env_hetero.py:198-214), the one
 * reward term that reads the post-tick distances inside the tick: such worlds can hand the pair table to the output wave like the presets. */
template <int W, int PRE, bool TWO, int APW = 16, bool DUAL = false, bool SHAPE = true>
__global__ __launch_bounds__(TWO ? 128 : 64, W) __attribute__((amdgpu_waves_per_eu(W, W))) void hh_k_world_quad(DevPtrs P, DevCfg c_in, int T, const int8_t *__restrict__ actions,
                                                                  float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                                  uint8_t *__restrict__ valid_out, uint8_t *__restrict__ done_out) {
    constexpr int A = 4, B = 64, GPB = APW;
    static_assert(APW == 16 || APW == 8, "arenas per wave");
    static_assert(!DUAL || APW == 8, "helper lanes exist only where half the wave is idle");
    DevCfg c_pre = c_in;
    hh_cfg_set_preset(c_pre, PRE);
    const DevCfg &c = PRE ? c_pre : c_in;
    __shared__ Shared<A, B> sh;
    __shared__ QuadMailbox<TWO> mbx;
    const int tid = threadIdx.x & 63;
    const int D = c.D;
    /* OWT: the post-tick pair table is computed by the output wave (QPosMail above).  Not with the escape distance shaping (env_hetero.py:198-214),
     * whose reward needs the post-tick distances before the tick's rows are posted: the presets never have it, the general instance only when
     * the launcher picked its SHAPE = false form (hh_world.hip). */
    static_assert(SHAPE || (PRE == 0 && TWO), "SHAPE = false names the general two-wave instance without escape shaping");
    constexpr bool OWT = TWO && (PRE != 0 || !SHAPE);
    if constexpr (OWT) {
        if (threadIdx.x >= 64) { /* ---------------- the output wave: pair table, then the observation rows ---------------- */
            const bool helper = DUAL && tid >= 32;
            const int mt = DUAL ? (tid & 31) : tid;
            const int g = mt >> 2, s = mt & 3; /* the simulation wave's lane layout: one lane per aircraft, helpers above */
            const int n = blockIdx.x * GPB + g;
            const bool row = !helper && s < 2 && g < GPB && n < c.N;
#ifdef HH_PROFILE_PHASES
            unsigned long long ot0_ = __builtin_readcyclecounter(), oacc_[4] = {0, 0, 0, 0};
#define HH_OPROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); oacc_[k] += t_ - ot0_; ot0_ = t_; } while (0)
#else
#define HH_OPROF(k)
#endif
            const unsigned long long akey = hh_rng_arena_key(c.seed, c.arena_offset + (uint64_t)n);
            const bool stat_lane = row && s == 0; /* one lane per arena keeps the episode statistics */
            double ep_ret = stat_lane ? P.ep_ret[n] : 0.0;
            for (int t = 0; t < T; t++) {
                __syncthreads(); /* barrier X: the moved positions are posted */
                HH_OPROF(0);
                Unit m = Unit{};
                QPub pub;
                m.lat = mbx.pos.lat[tid]; m.lon = mbx.pos.lon[tid]; m.spd = mbx.pos.spd[tid]; m.hdg = mbx.pos.hdg[tid];
                m.ac_type = mbx.pos.ac_type[tid];
                pub.flags = 0;
                /* the heading vector now; the normalised entries after Y (only rows read them): the expressions of the simulation wave's reset ticks.  The
                 * 8-arena form evaluates the tick's TWO sincos in one pass: the main lane its heading vector (of 90 - hdg), its helper lane the sine /
                 * cosine of hdg that the opponents' script rounds (_correct_angle_sign), handed up afterwards */
                double rs_ahead = 0.0, rc_ahead = 0.0;
                if constexpr (DUAL) {
                    const double hmain = q_down_d(m.hdg);
                    double sn, cs;
                    hh_sincos((helper ? hh_pymod360(hmain) : hh_pymod360(90.0 - hmain)) * (HH_PI / 180.0), &sn, &cs);
                    pub.uc = cs; pub.us = sn; pub.un = hh_sqrt(cs * cs + sn * sn);
                    rs_ahead = q_up_d(hh_round3(sn)); rc_ahead = q_up_d(hh_round3(cs));
                } else {
                    quad_publish_vec(m, pub);
                    double sn, cs;
                    hh_sincos(hh_pymod360(m.hdg) * (HH_PI / 180.0), &sn, &cs);
                    rs_ahead = hh_round3(sn); rc_ahead = hh_round3(cs);
                }
                /* ahead of the simulation wave (QPre), first what does not need the table — so that it shares the table's long dependent chains' shadow:
                 * the next tick's key, the arena's escape flag for that tick, this lane's script draw */
                const int steps_t = mbx.pos.steps[mt]; /* read HERE: after Y the simulation wave posts the next tick's */
                const unsigned long long tk1 = hh_rng_tick_key(akey, (uint32_t)mbx.pos.episode[mt], (uint32_t)(steps_t + 1));
                const bool l3 = !c.ext_opp && c.level >= 3; /* configuration: the level-3 script of tick t + 1, on the prediction that tick t removes nobody (QPre.spec) */
                const int ew = mbx.pos.escw[mt];
                int esc = ew & 0xff, esc_t = (int)(int8_t)((ew >> 8) & 0xff);
                const int am = (ew >> 16) & 0xf;
                bool my_escaping = false, escj[2] = {false, false};
                double u0 = 0.0, u1 = 0.0, u2 = 0.0;
                if (l3) {
                    quad_l3_flags(steps_t + 1, tk1, am, s, esc, esc_t, my_escaping, escj);
                    /* the three draws, as tick_quad spreads them over the lanes */
                    const int du = (s | 2) + 1, role = helper ? 2 : (s < 2 ? 1 : 0);
                    const int site_e = role == 0 ? HH_SITE_ESC_HDG : (role == 1 ? HH_SITE_ESC_SPEED : HH_SITE_ESC_FIRE);
                    const int site_h = role == 0 ? HH_SITE_HC_SPEED1 : (role == 1 ? HH_SITE_HC_R : HH_SITE_HC_SPEED2);
                    const bool desc = (s & 1) ? escj[1] : escj[0];
                    int de = desc ? 1 : 0;
                    if (DUAL) { const int deh = q_down_i(de); de = helper ? deh : de; }
                    const double u = hh_rng_u01(tk1, (uint32_t)du, (uint32_t)(de ? site_e : site_h), 0u);
                    u0 = u; u1 = q_perm_d<HH_QP(0, 1, 0, 1)>(u);
                    if (DUAL) u2 = q_up_d(u);
                    else u2 = hh_rng_u01(tk1, (uint32_t)(s + 1), (uint32_t)(my_escaping ? HH_SITE_ESC_FIRE : HH_SITE_HC_SPEED2), 0u);
                }
                QTab tb;
                quad_tables<DUAL, false>(m, pub, s, helper, tb); /* the same expressions on the same operands as the simulation wave's own (reset ticks) */
#pragma unroll
                for (int k = 0; k < 3; k++) { mbx.tab.dist[k][tid] = tb.dist[k]; mbx.tab.foc[k][tid] = tb.foc[k]; mbx.tab.focr[k][tid] = tb.focr[k]; }
                mbx.tab.uc[tid] = pub.uc; mbx.tab.us[tid] = pub.us;
                mbx.tab.tk[tid] = tk1;
                mbx.tab.sx[tid] = rs_ahead; mbx.tab.sy[tid] = rc_ahead;
                /* _nearby_object on this wave's table with the alive mask the tick started with: what the simulation wave would compute behind Y whenever the
                 * tick changes no alive mask (QPre.spec), with the entries of the nearest one (QTgt) */
                tb.amask = am;
                Near2 nb;
                quad_nearby(c, tb, s, nb);
                const double foc_k0 = q_sel(tb.foc, nb.k0);
                mbx.tab.nr0[tid] = nb.r0; mbx.tab.nfoc[tid] = foc_k0; mbx.tab.nfocr[tid] = q_sel(tb.focr, nb.k0);
                mbx.tab.nbw[tid] = (nb.n & 3) | ((nb.j0 & 3) << 2) | ((nb.k0 & 3) << 4);
                if (l3) { /* the script itself, from this wave's own table */
                    QScriptOut so_;
                    quad_l3_script(c, m.lat, m.lon, m.hdg, m.ac_type, my_escaping, u0, u1, u2, nb, tb, foc_k0, rs_ahead, rc_ahead, so_);
                    mbx.tab.sp_hdg[tid] = so_.heading; mbx.tab.sp_spd[tid] = so_.speed;
                    mbx.tab.sp_w[tid] = (so_.fire & 1) | ((so_.fire_m & 1) << 1) | (((so_.opp + 1) & 7) << 2) | ((esc & 0xff) << 8) | ((esc_t & 0xff) << 16);
                }
                HH_OPROF(1);
                __syncthreads(); /* barrier Y: the table is there for the simulation wave; its integers are here */
                HH_OPROF(2);
                int valid = 0, done = 0;
                float rew = 0.0f;
                { /* episode statistics (the simulation wave's own code in the other forms): return summed in agent order, then length and outcome */
                    const int w7s = mbx.slim.w7[tid];
                    const double rv = ((w7s >> 24) & 1) ? mbx.slim.rew[tid] : 0.0;
                    const double r1 = q_rot_d<1>(rv); /* agent 2's, as lane s = 0 sees it */
                    const bool ran = ((w7s >> 26) & 1) != 0, dn = ((w7s >> 25) & 1) != 0;
                    const double e2 = (ep_ret + rv) + r1;
                    ep_ret = ran ? e2 : ep_ret;
                    if (HH_RARE(stat_lane && ran && dn)) { /* an episode ended */
                        const int am = (w7s >> 27) & 0xf, steps = steps_t;
                        const int ag = __popc(am & 3), op = __popc(am & 12);
                        P.last_ret[n] = (float)ep_ret;
                        P.last_len[n] = steps;
                        P.last_outcome[n] = (op <= 0 && steps < c.horizon) ? 1 : ((ag <= 0 && steps < c.horizon) ? -1 : 0);
                    }
                    if (dn && c.auto_reset) ep_ret = 0.0; /* the arena starts a new episode (K3 of the simulation wave) */
                }
                if (HH_RARE(mbx.slim.full)) { /* a reset in this tick (wave-uniform): the rows come from the simulation wave's own table of the new episodes */
                    if (row) mail_take(mbx.mail[0], g * 2 + s, tb, pub, m, valid, done, rew);
                } else {
                    pub.flags = mbx.slim.flags[tid];
                    quad_publish_norm(c, m, pub);
                    quad_tables_obs(pub, s, tb);
                    const int w5 = mbx.slim.w5[tid], w6 = mbx.slim.w6[tid], w7 = mbx.slim.w7[tid];
                    m.cannon_remain = w5 & 0xffff; m.cannon_max = (w5 >> 16) & 0xffff;
                    m.missile_remain = w6 & 0xff; m.rocket_max = (w6 >> 8) & 0xff; m.missile_wait = (w6 >> 16) & 0xff; m.burst = (w6 >> 24) & 0xff;
                    m.ac_type = w7 & 0xff; m.alive = (w7 >> 8) & 0xff; m.has_missile = (w7 >> 16) & 0xff;
                    valid = (w7 >> 24) & 1;
                    done = (w7 >> 25) & 1;
                    rew = (float)mbx.slim.rew[tid];
                }
                if (HH_USUAL(row)) {
                    quad_lowlevel_obs(c, tb, pub, s, c.agent_mode, m, &mbx.tile[(g * 2 + s) * D], D);
                    const size_t o = ((size_t)t * c.N + n) * 2 + s;
                    if (reward_out) reward_out[o] = rew;
                    if (valid_out) valid_out[o] = (uint8_t)valid;
                    if (s == 0 && done_out) done_out[(size_t)t * c.N + n] = (uint8_t)done;
                }
                q_wave_sync();
                if (obs_out) quad_store_rows<GPB>(obs_out, mbx.tile, t, c.N, D, tid);
                q_wave_sync(); /* the tile is free again */
                HH_OPROF(3);
            }
            if (stat_lane) P.ep_ret[n] = ep_ret;
#ifdef HH_PROFILE_PHASES
            if (tid == 0) for (int k_ = 0; k_ < 4; k_++) atomicAdd(&hh_prof_cycles[16 + k_], oacc_[k_]);
#endif
#undef HH_OPROF
            return;
        }
    } else if constexpr (TWO) {
        if (threadIdx.x >= 64) { /* ---------------- the output wave ---------------- */
            const int g = tid >> 1, s = tid & 1; /* lanes 0..31: one agent row each */
            const int n = blockIdx.x * GPB + g;
            const bool row = tid < 2 * GPB && n < c.N;
            for (int t = 0; t < T; t++) {
                __syncthreads(); /* mailbox t is posted */
                const ObsMail &mb = mbx.mail[t & 1];
                if (row) {
                    QTab tb; QPub pub; Unit m;
                    int valid, done; float rew;
                    mail_take(mb, tid, tb, pub, m, valid, done, rew);
                    quad_lowlevel_obs(c, tb, pub, s, c.agent_mode, m, &mbx.tile[tid * D], D);
                    const size_t o = ((size_t)t * c.N + n) * 2 + s;
                    if (reward_out) reward_out[o] = rew;
                    if (valid_out) valid_out[o] = (uint8_t)valid;
                    if (s == 0 && done_out) done_out[(size_t)t * c.N + n] = (uint8_t)done;
                }
                q_wave_sync();
                if (obs_out) quad_store_rows<GPB>(obs_out, mbx.tile, t, c.N, D, tid);
                q_wave_sync(); /* the tile is free again */
            }
            return;
        }
    }
    /* ---------------- the simulation wave (the only wave when !TWO) ---------------- */
    const bool helper = DUAL && tid >= 32;              /* lanes 32..63: helpers of lane - 32 (quad_tables, the moves) */
    const int mt = DUAL ? (tid & 31) : tid;
    const int g = mt >> 2, s = mt & 3;
    const int base = g * A;
    const int n = blockIdx.x * GPB + g;
    const bool active = !helper && g < GPB && n < c.N;
    const size_t U = (size_t)c.N * A;
    const size_t u = (size_t)n * A + s;
    HH_PROF_DECL;
    Unit m = Unit{};
    Arena ar = Arena{};
    double ep_ret = 0.0;
    if (active) {
        unit_load(P, U, u, m);
        arena_load(P, c, n, ar);
        ep_ret = P.ep_ret[n];
    } else {
        ar.done = 1;
    }
    uint32_t evm_last = 0;
    int act_fault = 0; /* a consumed action word was out of range and ran sanitised (hh_act_unpack) */
    {
        const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
        if (tid < 11) sh.rk_speed[tid] = speed_table[tid];
    }
    QPub pub;
    QTab tb;
    Near2 nbc;
    quad_publish(c, m, pub);
    quad_tables<DUAL>(m, pub, s, helper, tb);
    quad_nearby(c, tb, s, nbc);
    QTgt tg;
    { /* the lane's target entries (OWT).  An agent's opp_to_attack comes with the loaded state and need not be the nearest opponent (hh_set_state): the first
       * tick reads the entries of THAT slot; every later tick's target is the refreshed one = nbc.j0 */
        const int ks = (s < 2 && m.n_tgt) ? ((m.tgt0 - 1 - s) & 3) : nbc.k0;
        tg.foc = q_sel(tb.foc, ks); tg.focr = q_sel(tb.focr, ks); tg.dist = q_sel(tb.dist, ks);
    }
    /* Action words: vmcnt is one in-order counter for loads AND stores, so waiting for a load also waits for the
     * write acknowledgements of every store issued before it.  The word of tick t+1 is therefore taken (waited
     * for) right after tick t's compute and BEFORE tick t's output stores, when the load is a whole tick old and
     * the previous tick's stores have long been acknowledged; the request for tick t+2 goes out at the same place. */
    const bool has_act = active && s < c.n_ctrl;
    QPre pre;
    pre.ok = false; pre.spec = false; pre.sp_hdg = pre.sp_spd = 0.0; pre.sp_w = 0;
    pre.tkey = 0ULL; pre.sx = pre.sy = 0.0;
    const size_t act_stride = (size_t)c.N * c.n_ctrl * 4;
    const int8_t *act_ptr = has_act ? actions + ((size_t)n * c.n_ctrl + s) * 4 : actions; /* lanes without a row re-read row 0, unused */
    int act_cur = *reinterpret_cast<const int *>(act_ptr);
    int act_next = *reinterpret_cast<const int *>(act_ptr + (size_t)min(1, T - 1) * act_stride);
    q_wave_sync();
    for (int t = 0; t < T; t++) {
        StepOut so;
        int8_t act[4];
        const bool was_running = active && !ar.done;
        hh_act_unpack(act_cur, act, act_fault, has_act & was_running & (m.alive != 0));
        QPosMail *posmail = nullptr;
        if constexpr (OWT) posmail = &mbx.pos;
        const int amask_before = tb.amask; /* alive at tick start: QPre.spec holds when the tick leaves it as it is */
        tick_quad<(W >= 2), DUAL, OWT>(c, sh, tid, g, s, base, active, helper, m, ar, act, tb, pub, nbc, tg, so, evm_last, posmail, pre HH_PROF_PASS);
        const int done_now = ar.done;
        if constexpr (TWO && !OWT) { /* post the agents' rows as early as they exist: the LDS stores drain behind the work below */
            if (!helper && s < 2) mail_post(mbx.mail[t & 1], g * 2 + s, tb, pub, m, so, done_now);
        }
        asm volatile("" : "+v"(act_next)); /* take the word of tick t+1 HERE (see above) ... */
        act_cur = act_next;
        act_next = *reinterpret_cast<const int *>(act_ptr + (size_t)min(t + 2, T - 1) * act_stride); /* ... and request t+2 */
        if (!TWO && HH_USUAL(active && s < 2)) {
            size_t o = ((size_t)t * c.N + n) * 2 + s;
            if (reward_out) reward_out[o] = (float)so.reward;
            if (valid_out) valid_out[o] = (uint8_t)so.valid;
        }
        const int amask_stats = tb.amask; /* who is alive after the tick, before a reset */
        /* episode statistics in agent order (every lane of the arena keeps the same running sum); OWT: kept by the output wave */
        if constexpr (!OWT) {
            const double rv = so.valid ? so.reward : 0.0;
            const double r0 = q_bc_d<0>(rv), r1 = q_bc_d<1>(rv);
            {
                const double e2 = (ep_ret + r0) + r1;
                ep_ret = was_running ? e2 : ep_ret;
                const bool fin = was_running & (ar.done != 0) & (s == 0);
                if (HH_RARE(q_any(fin))) if (fin) { /* an episode ended: rare */
                    const int ag = __popc(tb.amask & 3), op = __popc(tb.amask & 12);
                    P.last_ret[n] = (float)ep_ret;
                    P.last_len[n] = ar.steps;
                    P.last_outcome[n] = (op <= 0 && ar.steps < c.horizon) ? 1 : ((ag <= 0 && ar.steps < c.horizon) ? -1 : 0);
                }
            }
        }
        if (!TWO && active && s == 0 && done_out) done_out[(size_t)t * c.N + n] = (uint8_t)ar.done;
        const bool need_reset = active && ar.done && c.auto_reset;
        const bool reset_tick = q_any(need_reset);
        if (HH_RARE(reset_tick)) { /* K3, wave-uniform */
            if (need_reset) {
                reset_arena_scalars(ar);
                reset_unit<A>(c, s, m, ar);
                ep_ret = 0.0;
            }
            quad_publish(c, m, pub);
            quad_tables<DUAL>(m, pub, s, helper, tb);
            quad_nearby(c, tb, s, nbc);
            tg.foc = q_sel(tb.foc, nbc.k0); tg.focr = q_sel(tb.focr, nbc.k0); tg.dist = nbc.r0; /* the target refresh below makes j0 everybody's target */
            if constexpr (OWT) { /* the rows of this tick come from THIS table (the first observation of the new episodes) */
                if (!helper && s < 2) mail_post(mbx.mail[0], g * 2 + s, tb, pub, m, so, done_now);
            } else if constexpr (TWO) { /* the first observation of the new episode replaces the posted rows */
                if (!helper && s < 2) mail_post(mbx.mail[t & 1], g * 2 + s, tb, pub, m, so, done_now);
            }
        }
        HH_PROF(9);
        if constexpr (OWT) {
            /* the integers of the agents' rows (every lane posts its own words: no exec-mask region), then barrier Y */
            mbx.slim.flags[tid] = pub.flags;
            mbx.slim.w5[tid] = (m.cannon_remain & 0xffff) | ((m.cannon_max & 0xffff) << 16);
            mbx.slim.w6[tid] = (m.missile_remain & 0xff) | ((m.rocket_max & 0xff) << 8) | ((m.missile_wait & 0xff) << 16) | ((m.burst & 0xff) << 24);
            mbx.slim.w7[tid] = (m.ac_type & 0xff) | ((m.alive & 0xff) << 8) | ((m.has_missile & 0xff) << 16) | ((so.valid & 1) << 24) | ((done_now & 1) << 25) |
                               ((was_running ? 1 : 0) << 26) | ((amask_stats & 0xf) << 27);
            mbx.slim.rew[tid] = so.reward;
            if (tid == 0) mbx.slim.full = reset_tick ? 1 : 0;
            HH_PROF(9);
            __syncthreads(); /* barrier Y */
            HH_PROF(13);
            pre.ok = !reset_tick;
            pre.spec = !reset_tick && !q_any(active && tb.amask != amask_before);
            if (HH_USUAL(!reset_tick)) { /* wave-uniform: take what the output wave built while this wave ran the envelope phases, and what it computed ahead */
                pre.tkey = mbx.tab.tk[tid]; pre.sx = mbx.tab.sx[tid]; pre.sy = mbx.tab.sy[tid];
                pub.uc = mbx.tab.uc[tid]; pub.us = mbx.tab.us[tid]; /* the exact heading vector (the tick carried a rotated one) */
                pre.sp_hdg = mbx.tab.sp_hdg[tid]; pre.sp_spd = mbx.tab.sp_spd[tid]; pre.sp_w = mbx.tab.sp_w[tid];
                if (HH_USUAL(pre.spec)) { /* wave-uniform: no alive mask of the wave changed — the output wave's _nearby_object and target entries are this tick's (QTgt) */
                    const int w = mbx.tab.nbw[tid];
                    const double r0 = mbx.tab.nr0[tid];
                    tg.foc = mbx.tab.nfoc[tid]; tg.focr = mbx.tab.nfocr[tid]; tg.dist = r0;
                    nbc.n = w & 3; nbc.j0 = (w >> 2) & 3; nbc.k0 = (w >> 4) & 3;
                    nbc.r0 = r0; nbc.d0 = c.inv_diag * r0; /* quad_nearby's own product (r0 = 0 without a live opponent) */
                    nbc.j1 = nbc.k1 = 0; nbc.d1 = nbc.r1 = 0.0; /* the second entry is read by escape-mode rows and the escape shaping only: never on this wave */
                } else { /* somebody was removed: the table's distances with the new alive mask */
                    QTab tf = tb;
#pragma unroll
                    for (int k = 0; k < 3; k++) { tf.dist[k] = mbx.tab.dist[k][tid]; tf.foc[k] = mbx.tab.foc[k][tid]; tf.focr[k] = mbx.tab.focr[k][tid]; }
                    quad_nearby(c, tf, s, nbc);
                    tg.foc = q_sel(tf.foc, nbc.k0); tg.focr = q_sel(tf.focr, nbc.k0); tg.dist = nbc.r0;
                }
            }
            { /* env_hetero.py:99-101: the observation refreshes opp_to_attack (straight-line on every lane, kept by the agents') */
                Unit mr = m;
                quad_target_refresh(nbc, mr);
                const bool keep = active & (s < 2);
                m.n_tgt = keep ? mr.n_tgt : m.n_tgt; m.tgt0 = keep ? mr.tgt0 : m.tgt0; m.tgt_d0 = keep ? mr.tgt_d0 : m.tgt_d0;
            }
            HH_PROF(10);
            continue;
        } else if constexpr (TWO) {
            /* hand the agents' rows to the output wave and go on */
            { /* straight-line on every lane, kept by the agents' */
                Unit mr = m;
                quad_target_refresh(nbc, mr);
                const bool keep = active & (s < 2);
                m.n_tgt = keep ? mr.n_tgt : m.n_tgt; m.tgt0 = keep ? mr.tgt0 : m.tgt0; m.tgt_d0 = keep ? mr.tgt_d0 : m.tgt_d0;
            }
            __syncthreads(); /* the one workgroup barrier of the tick */
            HH_PROF(10);
            continue;
        }
        /* K2: observation rows staged in LDS, then written with unit-stride 16-byte stores */
        if (HH_USUAL(active && s < 2)) quad_lowlevel_obs(c, tb, pub, s, c.agent_mode, m, &sh.u.obs[(g * 2 + s) * D], D);
        q_wave_sync();
        if (obs_out) quad_store_rows<GPB>(obs_out, sh.u.obs, t, c.N, D, tid);
        q_wave_sync();
        HH_PROF(10);
    }
    HH_PROF_FLUSH;
    if (active) {
        unit_store(P, U, u, m);
        if (s == 0) {
            arena_store(P, n, ar);
            if constexpr (!OWT) P.ep_ret[n] = ep_ret; /* OWT: the output wave keeps and stores it */
            P.ev_mask[n] = 0;
        }
    }
    q_wave_sync();
    if (active && evm_last) atomicOr(&P.ev_mask[n], evm_last);
    hh_act_fault_commit(P, n, active, act_fault);
}

#endif /* HH_KERNELS_QUAD_H */
