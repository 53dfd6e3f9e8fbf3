/*
 * hh_kernels_oct.h — HighLevelEnv (3-vs-3 commander, envs/env_hier.py) with REGISTER exchange.
 *
 * Same path, same arithmetic and same results as hh_k_hier_macro<6, 64, W, HLD> (hh_kernels_hier.h), which stays the A/B
 * reference (HH_NO_OCT=1):
 *   envs/env_hier.py:114-140 _take_action = _action_assess (142-190) + <= 16 x { every live unit: _take_base_action
 *   (env_base.py:214-238) } -> cmano_simulator.py:138-157 do_tick -> env_base.py:240-310 rewards / events -> env_hier.py:49-98 state.
 *
 * What differs is how the aircraft of an arena see each other.  An arena is one aligned group of EIGHT lanes laid out as two
 * quads of three:   [ a0 a1 a2 -- | o0 o1 o2 -- ]   (agents in the first quad, opponents in the second, lanes 3 and 7 idle).
 * gfx950's DPP has no 8-lane rotate (rows are 16 lanes, DPP8 is not in this ISA), but with this layout every neighbour is at
 * most two VALU moves away and no LDS round trip (write, s_waitcnt, read, s_waitcnt) is left on the tick's critical path:
 *   - the two FRIENDS are a rotation of the ring of three inside the quad:     one v_mov_b32_dpp quad_perm [1,2,0,3] / [2,0,1,3];
 *   - the three ENEMIES are the other quad: a quad swap = row_shl:4 into banks 0,2 + row_shr:4 into banks 1,3 (two moves for the
 *     enemy with the same index), then the same ring rotations for the other two;
 *   so a word of all five others costs six moves.  The pair table is indexed by RELATION r = 0,1: friends (index + 1, + 2 mod 3),
 *   r = 2,3,4: enemies (index + 0, + 1, + 2 mod 3); the relation under which Y sees X is a compile-time function of the one
 *   under which X sees Y (friends 0 <-> 1, enemies 2 <-> 2, 3 <-> 4), so symmetric quantities are computed once per pair and
 *   handed over, and "focus of the other at me" is a fetch of the other lane's table entry — as in the 2-vs-2 kernel;
 *   - kill resolution runs redundantly on all lanes of the arena from broadcast words (quad broadcast + one cross-quad move),
 *     and is skipped for the whole wave on the (usual) ticks where no lane holds a hit or a rocket fuse / end-of-life bit;
 *   - arena-wide counts (alive per side, launches in id order, out-of-bounds, the surrounding event) are wave ballots;
 *   - the envelope tests keep the workgroup queue in LDS (work compaction across arenas), entered only when it is not empty.
 * The macro step also builds the pair table only where it is consumed: a pilot-action TAPE needs no pilot observations, so
 * between _action_assess and the commander observation the table is read by the surrounding event alone (sub-steps 11..15,
 * env_hier.py:133-135); a launch test computes its one entry on demand (the same expression, the same bits).
 * DPP reads are only made from wave-uniform control flow (a disabled source lane would read as 0).
 */
#ifndef HH_KERNELS_OCT_H
#define HH_KERNELS_OCT_H

#include "hh_kernels_hier.h"

#define HH_DPP_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define HH_DPP_ROW_SHL(n) (0x100 + (n))
#define HH_DPP_ROW_SHR(n) (0x110 + (n))

template <int CTRL>
__device__ __forceinline__ int o_perm_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
/* the other quad's lane with the same index */
__device__ __forceinline__ int o_swap_i(int v) {
    const int a = __builtin_amdgcn_update_dpp(v, v, HH_DPP_ROW_SHL(4), 0xf, 0x5, false); /* lanes 0-3, 8-11  <- lane + 4 */
    return __builtin_amdgcn_update_dpp(a, v, HH_DPP_ROW_SHR(4), 0xf, 0xa, false);        /* lanes 4-7, 12-15 <- lane - 4 */
}
/* ring of three inside the quad: value of index (i + K) % 3 (lane 3 keeps its own) */
template <int K>
__device__ __forceinline__ int o_rot_i(int v) {
    static_assert(K == 1 || K == 2, "ring of three");
    return o_perm_i<(K == 1 ? HH_DPP_QP(1, 2, 0, 3) : HH_DPP_QP(2, 0, 1, 3))>(v);
}
/* value held by lane position P (0..7) of the lane's 8-lane group */
template <int P>
__device__ __forceinline__ int o_bc_i(int v) {
    const int t = o_perm_i<HH_DPP_QP(P & 3, P & 3, P & 3, P & 3)>(v);
    if (P < 4) return __builtin_amdgcn_update_dpp(t, t, HH_DPP_ROW_SHR(4), 0xf, 0xa, false);
    return __builtin_amdgcn_update_dpp(t, t, HH_DPP_ROW_SHL(4), 0xf, 0x5, false);
}
/* value held by index J of the lane's own quad */
template <int J>
__device__ __forceinline__ int o_qbc_i(int v) { return o_perm_i<HH_DPP_QP(J, J, J, J)>(v); }
#define HH_O_LIFT_D(NAME, TPL, ...)                                                              \
    TPL __device__ __forceinline__ double NAME##_d(double v) {                                   \
        int lo = __double2loint(v), hi = __double2hiint(v);                                      \
        lo = NAME##_i __VA_ARGS__(lo);                                                           \
        hi = NAME##_i __VA_ARGS__(hi);                                                           \
        return __hiloint2double(hi, lo);                                                         \
    }                                                                                            \
    TPL __device__ __forceinline__ float NAME##_f(float v) { return __int_as_float(NAME##_i __VA_ARGS__(__float_as_int(v))); }
HH_O_LIFT_D(o_swap, , )
HH_O_LIFT_D(o_rot, template <int K>, <K>)
HH_O_LIFT_D(o_bc, template <int P>, <P>)
HH_O_LIFT_D(o_qbc, template <int J>, <J>)
#undef HH_O_LIFT_D

/* a word of all five others by relation: [0] friend +1, [1] friend +2, [2] enemy +0, [3] enemy +1, [4] enemy +2 */
#define HH_O_FETCH5(T, dst, v)                                         \
    do {                                                               \
        (dst)[0] = o_rot_##T<1>(v); (dst)[1] = o_rot_##T<2>(v);        \
        const auto sw_ = o_swap_##T(v);                                \
        (dst)[2] = sw_; (dst)[3] = o_rot_##T<1>(sw_); (dst)[4] = o_rot_##T<2>(sw_); \
    } while (0)
/* select by relation (operands by VALUE: a select between array addresses would pin the table in scratch memory) */
template <class T>
__device__ __forceinline__ T o_sel5v(T a0, T a1, T a2, T a3, T a4, int r) {
    T x = a4;
    if (r == 3) x = a3;
    if (r == 2) x = a2;
    if (r == 1) x = a1;
    if (r == 0) x = a0;
    return x;
}
#define o_sel5(arr, r) o_sel5v((arr)[0], (arr)[1], (arr)[2], (arr)[3], (arr)[4], (r))
template <class T>
__device__ __forceinline__ T o_sel3v(T a0, T a1, T a2, int e) {
    T x = a2;
    if (e == 1) x = a1;
    if (e == 0) x = a0;
    return x;
}

/* who the lane is */
struct OLane {
    int g, p, q, i;   /* arena of the wave, lane position 0..7, side (0 agents / 1 opponents), index inside the side (3 = idle) */
    int s;            /* unit slot = id - 1 (agents 0..nA-1, opponents nA..nA+nO-1) */
    int base;         /* tid of lane position 0 */
    bool exists;      /* the arena is in range and the world has this aircraft */
};
__device__ __forceinline__ int o_mod3(int x) { return x >= 3 ? x - 3 : x; } /* x in 0..5 */
/* index inside its side, lane position and slot of relation r */
__device__ __forceinline__ int o_rel_index(const OLane &L, int r) { return o_mod3((L.i > 2 ? 0 : L.i) + (r < 2 ? r + 1 : r - 2)); }
__device__ __forceinline__ int o_rel_pos(const OLane &L, int r) { return ((r < 2 ? L.q : 1 - L.q) << 2) | o_rel_index(L, r); }
__device__ __forceinline__ int o_pos_slot(const DevCfg &c, int pos) { return pos < 4 ? pos : c.nA + (pos - 4); }
__device__ __forceinline__ int o_slot_pos(const DevCfg &c, int slot) { return slot < c.nA ? slot : 4 + (slot - c.nA); }
/* relation under which the lane sees lane position `pos` (pos != own position) */
__device__ __forceinline__ int o_pos_rel(const OLane &L, int pos) {
    const int d = o_mod3((pos & 3) + 3 - (L.i > 2 ? 0 : L.i)); /* index distance in the ring */
    return (pos >> 2) == L.q ? d - 1 : 2 + d;
}

/* what the other lanes read from this aircraft (QPub of the 2-vs-2 kernel) */
struct OPub {
    double uc, us, un;             /* heading unit vector and its norm (env_base.py:428) */
    float nlat, nlon, nspd, nhdg;  /* normalised observation entries (env_base.py:117-121) */
    int flags;                     /* FL_ALIVE | type << 1 | FL_SHOT */
};
/* the arena as seen from this lane, by relation */
struct OTab {
    double lat[5], lon[5];                  /* positions (valid after every tick) */
    double dist[5], foc[5], focr[5];        /* planar distance [deg], focus me -> r, focus r -> me */
    double hd[3];                           /* heading difference, enemies only (read by observations about opponents) */
    float nlat[5], nlon[5], nspd[5], nhdg[5];
    int fl[5];
    int amask;                              /* alive bits by lane POSITION (bits 0-2 agents, 4-6 opponents), own bit included */
};

__device__ __forceinline__ void oct_publish_vec(const Unit &m, OPub &p) {
    double sn, cs;
    hh_sincos(hh_pymod360(90.0 - m.hdg) * (HH_PI / 180.0), &sn, &cs);
    p.uc = cs;
    p.us = sn;
    p.un = hh_sqrt(cs * cs + sn * sn);
}
__device__ __forceinline__ void oct_publish_norm(const DevCfg &c, const Unit &m, OPub &p) {
    p.nlat = (float)hh_clip(hh_div_known(m.lat - HH_MAP_LAT0, c.ext_lat, c.inv_ext_lat), 0.0, 1.0);
    p.nlon = (float)hh_clip(hh_div_known(m.lon - HH_MAP_LON0, c.ext_lon, c.inv_ext_lon), 0.0, 1.0);
    p.nspd = (float)hh_clip(hh_div_known(m.spd, HH_AC_MAX_SPEED(m.ac_type), HH_AC_INV_MAX_SPEED(m.ac_type)), 0.0, 1.0);
    p.nhdg = (float)hh_clip(HH_DIVC(hh_pymod359(m.hdg), 359.0), 0.0, 1.0);
}
__device__ __forceinline__ void oct_publish_flags(const Unit &m, OPub &p) {
    const int shot = m.burst > 0 || (m.ac_type == 1 && m.has_missile);
    p.flags = (m.alive ? FL_ALIVE : 0) | ((m.ac_type & 3) << 1) | (shot ? FL_SHOT : 0);
}
__device__ __forceinline__ int oct_arena_bits(unsigned long long ballot, const OLane &L) { return (int)((ballot >> (L.g * 8)) & 0xffULL); }

/* positions of the others + the alive mask: all a tape-driven tick needs from the table.  WAVE-UNIFORM control flow only. */
__device__ __forceinline__ void oct_positions(const Unit &m, const OLane &L, OTab &t) {
    HH_O_FETCH5(d, t.lat, m.lat);
    HH_O_FETCH5(d, t.lon, m.lon);
    t.amask = oct_arena_bits(__ballot(m.alive != 0), L);
}

/* the pair table of pair_tables() in registers (hh_kernels.h), entry for entry the same expressions.  ALL = false: only what the
 * surrounding event reads (distance and focus both ways between enemies), from the positions already in t.  WAVE-UNIFORM only. */
template <bool ALL>
__device__ __forceinline__ void oct_tables(const Unit &m, const OPub &p, const OLane &L, OTab &t) {
    const double c1 = p.uc, s1 = p.us, n1 = p.un;
    if (ALL) oct_positions(m, L, t);
    /* planar distances: dx*dx + dy*dy does not change when both differences flip sign, so a pair's distance is the same bits
     * from either end: friend +1 and enemies +0, +1 are computed (three square roots cover the fifteen pairs), friend +2 and
     * enemy +2 are the other end's friend +1 / enemy +1 */
    {
        const double dx2 = t.lon[2] - m.lon, dy2 = t.lat[2] - m.lat;
        const double dx3 = t.lon[3] - m.lon, dy3 = t.lat[3] - m.lat;
        t.dist[2] = hh_sqrt(dx2 * dx2 + dy2 * dy2);
        t.dist[3] = hh_sqrt(dx3 * dx3 + dy3 * dy3);
        t.dist[4] = o_rot_d<2>(o_swap_d(t.dist[3]));
        if (ALL) {
            const double dx0 = t.lon[0] - m.lon, dy0 = t.lat[0] - m.lat;
            t.dist[0] = hh_sqrt(dx0 * dx0 + dy0 * dy0);
            t.dist[1] = o_rot_d<2>(t.dist[0]);
        }
    }
#pragma unroll
    for (int r = ALL ? 0 : 2; r < 5; r++) {
        const double dx = t.lon[r] - m.lon, dy = t.lat[r] - m.lat;
        const double dot = c1 * dx + s1 * dy;
        const double x = hh_clip(dot / (n1 * t.dist[r] + 1e-10), -1.0, 1.0);
        t.foc[r] = hh_acos(x) * (180.0 / HH_PI);
    }
    /* focus of the other at me = the other lane's entry about me */
    t.focr[2] = o_swap_d(t.foc[2]); t.focr[3] = o_rot_d<1>(o_swap_d(t.foc[4])); t.focr[4] = o_rot_d<2>(o_swap_d(t.foc[3]));
    if (ALL) {
        t.focr[0] = o_rot_d<1>(t.foc[1]); t.focr[1] = o_rot_d<2>(t.foc[0]);
        double ouc[3], ous[3], oun[3];
        {   /* enemies' heading vectors */
            const double a = o_swap_d(p.uc), b = o_swap_d(p.us), n = o_swap_d(p.un);
            ouc[0] = a; ouc[1] = o_rot_d<1>(a);
            ous[0] = b; ous[1] = o_rot_d<1>(b);
            oun[0] = n; oun[1] = o_rot_d<1>(n);
        }
        HH_O_FETCH5(f, t.nlat, p.nlat);
        HH_O_FETCH5(f, t.nlon, p.nlon);
        HH_O_FETCH5(f, t.nspd, p.nspd);
        HH_O_FETCH5(f, t.nhdg, p.nhdg);
        HH_O_FETCH5(i, t.fl, p.flags);
        /* heading difference (env_base.py:448-456), symmetric in its operands and read only about enemies: +0, +1 computed,
         * +2 handed over */
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double dot = c1 * ouc[e] + s1 * ous[e];
            const double x = hh_clip(dot / (n1 * oun[e] + 1e-10), -1.0, 1.0);
            t.hd[e] = hh_clip(HH_DIVC(hh_acos(x) * (180.0 / HH_PI), 180.0), 0.0, 1.0);
        }
        t.hd[2] = o_rot_d<2>(o_swap_d(t.hd[1]));
    }
}

/* one table entry computed on demand (launch_planar_direct of hh_kernels.h): focus and distance towards relation r */
__device__ __forceinline__ void oct_entry_direct(const Unit &m, const OPub &p, double t_lat, double t_lon, double &foc, double &dist) {
    const double dx = t_lon - m.lon, dy = t_lat - m.lat;
    const double n2 = hh_sqrt(dx * dx + dy * dy);
    const double x = hh_clip((p.uc * dx + p.us * dy) / (p.un * n2 + 1e-10), -1.0, 1.0);
    foc = hh_acos(x) * (180.0 / HH_PI);
    dist = n2;
}

/* LDS of the register-exchange kernel: the envelope queue's exchange area + the output staging tile */
struct OctShared {
    double lat0[64], lon0[64], hdg[64];
    int flags[64], res[64];
    unsigned long long g_tkey[8];
    double rk_speed[12];
    union alignas(16) {
        struct {
            double lat1[64], lon1[64], hdg1[64];
            double rk_lat[64], rk_lon[64];
            int q_code[64 * 8];
        } t;
        float obs[8 * 3 * HH_OBS_HL];
        float prow[8 * 6 * 30]; /* every unit's pilot row of the wave's eight arenas, contiguous like [N, 6, 30] */
    } u;
};

__device__ __forceinline__ void o_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
/* any lane of the wave, as a scalar the compiler cannot fold back into the lanes' own test (q_any of hh_kernels_quad.h; control-flow
 * notes there: a region under the exec mask costs ~45 - 55 cycles at one wave per SIMD, so short bodies are selects and rare bodies
 * sit behind a uniform branch) */
__device__ __forceinline__ bool o_any(bool x) {
    unsigned long long b = __ballot(x);
    asm volatile("" : "+s"(b));
    return b != 0ULL;
}

/* env_base.py:400-422 _nearby_object from the register table: live units of the other side (friendly = false) or the own
 * side, stable-sorted by normalised distance (ties keep id order).  Near3's ids are RELATIONS here. */
__device__ __forceinline__ void oct_nearby(const DevCfg &c, const OTab &t, const OLane &L, bool friendly, Near3 &o) {
    o.n = 0; o.i0 = o.i1 = o.i2 = 0; o.d0 = o.d1 = o.d2 = 0.0; o.r0 = o.r1 = o.r2 = 0.0;
    const int side = friendly ? L.q : 1 - L.q;
    const int own = L.i > 2 ? 0 : L.i;
#pragma unroll
    for (int jj = 0; jj < 3; jj++) { /* id order inside the side */
        if (friendly && jj == own) continue;
        const int d = o_mod3(jj + 3 - own);
        const int r = friendly ? d - 1 : 2 + d;
        if (!((t.amask >> ((side << 2) | jj)) & 1)) continue;
        const double dr = o_sel5(t.dist, r);
        const double dn = c.inv_diag * dr;
        const int pp = (o.n >= 1 && o.d0 <= dn) + (o.n >= 2 && o.d1 <= dn) + (o.n >= 3 && o.d2 <= dn);
        if (pp <= 1) { o.i2 = o.i1; o.d2 = o.d1; o.r2 = o.r1; }
        if (pp == 0) { o.i1 = o.i0; o.d1 = o.d0; o.r1 = o.r0; o.i0 = r; o.d0 = dn; o.r0 = dr; }
        else if (pp == 1) { o.i1 = r; o.d1 = dn; o.r1 = dr; }
        else if (pp == 2) { o.i2 = r; o.d2 = dn; o.r2 = dr; }
        if (o.n < 3) o.n++;
    }
}

/* env_base.py:185-212 opp_ac_values about relation r (an enemy: r = 2..4); mode 0 fight / 1 escape / 2 HighLevel */
__device__ __forceinline__ int oct_opp_block(const OTab &t, int mode, int r, double dist, float *out) {
    const double f_so = o_sel5(t.foc, r), f_os = o_sel5(t.focr, r);
    int n = 0;
    out[n++] = o_sel5(t.nlat, r);
    out[n++] = o_sel5(t.nlon, r);
    out[n++] = o_sel5(t.nspd, r);
    out[n++] = o_sel5(t.nhdg, r);
    out[n++] = (float)o_sel3v(t.hd[0], t.hd[1], t.hd[2], r - 2);
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
    if (mode != 2) out[n++] = (o_sel5(t.fl, r) & FL_SHOT) ? 1.0f : 0.0f;
    return n;
}
/* env_base.py:166-183 friendly_ac_values about relation r (a live friend) */
__device__ __forceinline__ void oct_friend_block(const DevCfg &c, const OTab &t, int r, float *out) {
    out[0] = o_sel5(t.nlat, r);
    out[1] = o_sel5(t.nlon, r);
    out[2] = (float)norm180(o_sel5(t.foc, r));
    out[3] = (float)norm180(o_sel5(t.focr, r));
    out[4] = (float)(c.inv_diag * o_sel5(t.dist, r));
}

/* env_hier.py:49-98 state(): commander observation (agents) and the stored sorted target lists (all units) */
__device__ __forceinline__ void oct_commander_obs(const DevCfg &c, const OTab &t, const OPub &p, const OLane &L, Unit &m, float *out) {
    const bool agent = L.q == 0;
    if (agent) for (int k = 0; k < HH_OBS_HL; k++) out[k] = 0.0f;
    m.n_tgt = 0; m.tgt0 = m.tgt1 = m.tgt2 = 0; m.tgt_d0 = m.tgt_d1 = m.tgt_d2 = 0.0;
    if (!m.alive) return;
    Near3 nb;
    oct_nearby(c, t, L, false, nb);
    /* unit ids of the sorted relations */
    const int id0 = o_pos_slot(c, o_rel_pos(L, nb.i0)) + 1, id1 = o_pos_slot(c, o_rel_pos(L, nb.i1)) + 1, id2 = o_pos_slot(c, o_rel_pos(L, nb.i2)) + 1;
    if (agent) {
        if (nb.n == 0) return;
        int n = 0;
        out[n++] = p.nlat;
        out[n++] = p.nlon;
        out[n++] = p.nspd;
        out[n++] = p.nhdg;
        oct_opp_block(t, 2, nb.i0, nb.d0, out + n);
        m.n_tgt = 1; m.tgt0 = id0; m.tgt_d0 = nb.d0;
        if (nb.n >= 2) {
            oct_opp_block(t, 2, nb.i1, nb.d1, out + n + 10);
            m.n_tgt = 2; m.tgt1 = id1; m.tgt_d1 = nb.d1;
        }
        n += HH_N_OPP_HL * 10;
        Near3 fr;
        oct_nearby(c, t, L, true, fr);
        if (fr.n >= 1) oct_friend_block(c, t, fr.i0, out + n);
        if (fr.n >= 2) oct_friend_block(c, t, fr.i1, out + n + 5);
    } else { /* opponents keep the full sorted list of agents (env_hier.py:97) */
        m.n_tgt = nb.n;
        if (nb.n >= 1) { m.tgt0 = id0; m.tgt_d0 = nb.d0; }
        if (nb.n >= 2) { m.tgt1 = id1; m.tgt_d1 = nb.d1; }
        if (nb.n >= 3) { m.tgt2 = id2; m.tgt_d2 = nb.d2; }
    }
}

/* env_hier.py:142-190 _action_assess for the lane's unit (hl_action_assess of hh_kernels_hier.h from the register table) */
__device__ __forceinline__ double oct_action_assess(const DevCfg &c, const OTab &t, const OLane &L, Unit &m, const Arena &ar, int cmd) {
    const int id = L.s + 1;
    double rew = 0.0;
    if (!m.alive) { m.cmd_act = 0; return 0.0; }
    if (L.q == 0) {
        int cc = cmd;
        if (cc > 0) {
            const int t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2;
            int opp = 0;
            if (cc - 1 < m.n_tgt) { opp = t2; if (cc == 2) opp = t1; if (cc == 1) opp = t0; }
            else cc = 1;
            if (!opp) rew = -0.1;
            if (c.hier_action_assess && opp) {
                const int r = o_pos_rel(L, o_slot_pos(c, opp - 1));
                rew = (o_sel5(t.dist, r) < 0.1 && o_sel5(t.foc, r) < 15.0 && o_sel5(t.focr, r) > 40.0) ? 0.1 : 0.0;
            }
        } else if (c.hier_action_assess) {
            /* the reference indexes the stored list with 0 - 1 = -1 when there is no target either: hl_action_assess reads slot
             * tgt0 - 1 = -1 there, which the generic kernel's LDS layout answers with the previous lane's entry.  An agent with an
             * empty list has no live opponent, so its arena's episode is over and the branch is unreachable while running. */
            const int r = m.tgt0 ? o_pos_rel(L, o_slot_pos(c, m.tgt0 - 1)) : 2;
            if (o_sel5(t.dist, r) < 0.1 && o_sel5(t.focr, r) < 15.0 && o_sel5(t.foc, r) > 40.0) rew = 0.1;
        }
        m.cmd_act = cc;
    } else {
        const int g = hl_gcd(c.hier_opp_fight_ratio, 100);
        const int num = c.hier_opp_fight_ratio / (g ? g : 1), den = 100 / (g ? g : 1);
        const double total = (double)den + 0.0;
        const int fight = d_rng(ar, id, HH_SITE_HL_FIGHT, 0) * total >= (double)(den - num);
        int ag = 0;
        if (fight) {
            const int possible = m.n_tgt;
            if (possible > 1 && (d_rng(ar, id, HH_SITE_HL_OTHER, 0) * 4.0 >= 1.0))
                ag = hh_rng_randint(d_rng(ar, id, HH_SITE_HL_PICK, 0), 2, possible);
            else
                ag = 1;
        }
        m.cmd_act = ag;
    }
    return rew;
}

/* workgroup envelope queue: slots by wave ballot (the count is a scalar, an empty queue costs nothing) */
#define HH_O_PUSH(cond, code_)                                                                                        \
    do {                                                                                                              \
        const bool c_ = (cond);                                                                                       \
        const unsigned long long bm_ = __ballot(c_);                                                                  \
        if (HH_RARE(bm_ != 0ULL)) {                                                                                   \
            if (c_) sh.u.t.q_code[q_total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm_, 0u))] = (code_); \
            q_total += __popcll(bm_);                                                                                 \
        }                                                                                                             \
    } while (0)

/* env_base.py:214-238 _take_base_action of the lanes with `acts` (act_phase<.., hl = true> of hh_kernels.h): command decode, cannon,
 * the missile launch with its envelope test, launch bookkeeping in unit id order, weapon flags.  TAB: the register table holds the
 * current geometry; otherwise the launch test computes its entry on demand.  WAVE-UNIFORM call sites only. */
template <bool IX, bool TAB>
__device__ __forceinline__ void act_oct(const DevCfg &c, OctShared &sh, int tid, const OLane &L, bool running, Unit &m, Arena &ar,
                                        const int8_t (&act)[4], bool acts, const OTab &tb, OPub &pub, uint32_t &evm) {
    const int id = L.s + 1;
    const bool snap = L.exists && running && m.alive && acts;
    int want_launch = 0, launch_pos = 0;
    bool base_gate = false;
    if (HH_USUAL(snap)) {
        double dd;
        const int t = hl_target_slot(m, dd);
        double nh = hh_pymod360(m.hdg + (double)(((int)act[0] - 6) * 15));
        if (nh >= 360.0 || nh < 0.0) nh = 0.0;
        m.cmd_hdg = nh;
        const double mx = HH_AC_MAX_SPEED(m.ac_type);
        m.cmd_spd = 100.0 + ((mx - 100.0) / 8.0) * (double)act[1];
        if (act[2] && m.cannon_remain > 0) arm_cannon(m);
        {
            const bool gate = (m.ac_type == 1) & (act[3] != 0) & (t != 0) & (m.missile_remain > 0) & (m.has_missile == 0) & (m.missile_wait == 0);
            const int lp = o_slot_pos(c, t ? t - 1 : 0);
            base_gate = gate;
            want_launch = gate ? 1 : 0;
            launch_pos = gate ? lp : 0;
        }
    }
    const bool try_launch = want_launch && !m.has_missile && m.missile_remain > 0;
    int launch_pre = -1;
    if (o_any(try_launch)) if (try_launch) { /* planar stage of the launch predicate (hh_envelope.h) */
        const int r = o_pos_rel(L, launch_pos);
        const double t_lat = o_sel5(tb.lat, r), t_lon = o_sel5(tb.lon, r);
        double foc, dist;
        if (TAB) { foc = o_sel5(tb.foc, r); dist = o_sel5(tb.dist, r); }
        else oct_entry_direct(m, pub, t_lat, t_lon, foc, dist);
        const double cross = pub.uc * (t_lat - m.lat) - pub.us * (t_lon - m.lon);
        launch_pre = hh_missile_cone_planar(m.lat, m.lon, t_lat, t_lon, foc, cross, dist);
    }
    int q_total = 0, myres = 0;
    HH_O_PUSH(try_launch && launch_pre < 0, tid | (0 << 8) | (launch_pos << 10));
    if (HH_RARE(q_total != 0)) { /* wave-uniform */
        sh.lat0[tid] = m.lat; sh.lon0[tid] = m.lon; sh.hdg[tid] = m.hdg;
        sh.flags[tid] = pub.flags;
        sh.res[tid] = 0;
        if (L.p == 0) sh.g_tkey[L.g] = ar.tkey;
        o_wave_sync();
        drain_envelope_queue_t<8, 64, IX>(sh, tid, q_total, c.nA + 1);
        o_wave_sync();
        myres = sh.res[tid];
    }
    int launched = 0;
    const bool launch_now = try_launch & (((myres & 1) != 0) | (launch_pre == 1));
    if (o_any(launch_now)) if (launch_now) { /* ac1.py:76-79 */
        launched = 1;
        m.rk_alive = 1; m.rk_lat = m.lat; m.rk_lon = m.lon; m.rk_hdg = m.hdg; m.rk_cmd = m.hdg;
        m.rk_target = o_pos_slot(c, launch_pos) + 1; m.rk_life = 0;
        m.has_missile = 1;
        m.missile_remain = m.missile_remain - 1 > 0 ? m.missile_remain - 1 : 0;
        evm |= 1u << (24 + L.s);
    }
    if (o_any(base_gate)) if (base_gate) m.missile_wait = hh_rng_randint(d_rng(ar, id, HH_SITE_MISSILE_WAIT, 0), 8, 12);
    {
        const bool dec = snap & (m.missile_wait > 0) & (m.has_missile == 0);
        const int w1 = m.missile_wait - 1;
        m.missile_wait = dec ? w1 : m.missile_wait;
    }
    {   /* rocket ids in unit id order (cmano_simulator.py:104-108) */
        const int lb = oct_arena_bits(__ballot(launched != 0), L);
        const int sq = ar.next_seq + __popc(lb & ((1 << L.p) - 1)) + 1;
        m.rk_seq = launched ? sq : m.rk_seq;
        ar.next_seq += __popc(lb);
    }
    oct_publish_flags(m, pub); /* weapon flags other lanes observe (env_base.py:208-211) */
}

/* do_tick of one HighLevelEnv sub-step (tick<6, 64>(tmode 1) of hh_kernels.h): commands and launches were applied by act_oct.
 * On entry tb.lat / tb.lon / tb.amask hold the pre-tick positions and alive bits and pub the heading vector; on return they hold
 * the post-tick ones, and with FULL the whole post-tick pair table + published observation entries. */
template <bool IX, bool FULL>
__device__ __forceinline__ void tick_oct(const DevCfg &c, OctShared &sh, int tid, const OLane &L, bool running, Unit &m, Arena &ar, OTab &tb,
                                         OPub &pub, StepOut &out, uint32_t &ev_mask_out) {
    const int id = L.s + 1, s = L.s;
    const bool agent = L.q == 0;
    uint32_t evm = 0;
    out.kill_event = 0;
    const bool snap = running && L.exists && m.alive;
    const int amask0 = tb.amask;
    const int rk_age0 = (m.rk_alive && m.rk_life <= HH_ROCKET_MAX_LIFE) ? m.rk_life : 0;
    const double rk_speed0 = sh.rk_speed[rk_age0];

    /* ---------------- phase B: aircraft kinematics + move (ac1.py:81-133) ---------------- */
    const double lat_old = m.lat, lon_old = m.lon, hdg_old = m.hdg;
    bool fired = false;
    const int rk_pre = m.rk_alive;
    if (HH_USUAL(snap)) {
        const int t = m.ac_type;
        {
            const double delta = d_signed_heading_diff(m.hdg, m.cmd_hdg);
            const double max_deg = HH_AC_TURN_RATE(t) * 1.0;
            const double stepped = hh_pymod360(m.hdg + (delta >= 0.0 ? max_deg : -max_deg));
            const double nh = hh_fabs(delta) <= max_deg ? m.cmd_hdg : stepped;
            m.hdg = m.hdg != m.cmd_hdg ? nh : m.hdg;
        }
        {
            const double delta = m.cmd_spd - m.spd;
            const double max_delta = HH_AC_ACCEL(t) * 1.0;
            const double stepped = m.spd + (delta >= 0.0 ? max_delta : -max_delta);
            const double ns = hh_fabs(delta) <= max_delta ? m.cmd_spd : stepped;
            m.spd = m.spd != m.cmd_spd ? ns : m.spd;
        }
        {
            const bool f = m.burst > 0;
            const int b1 = m.burst - 1 > 0 ? m.burst - 1 : 0, c1 = m.cannon_remain - 1 > 0 ? m.cannon_remain - 1 : 0;
            fired = f;
            m.burst = f ? b1 : m.burst;
            m.cannon_remain = f ? c1 : m.cannon_remain;
        }
    }
    { /* ac1.py:117-128 */
        const bool steer = snap & (m.has_missile != 0) & (m.rk_alive != 0);
        if (HH_USUAL(steer)) m.rk_cmd = hh_clip(m.rk_hdg * hh_rng_uniform(d_rng(ar, id, HH_SITE_ROCKET_NOISE, 0), 0.95, 1.05), 0.0, 359.0);
        m.has_missile = (snap & (m.has_missile != 0) & (m.rk_alive == 0)) ? 0 : m.has_missile;
    }
    const bool rk_spec = running && L.exists && rk_pre && m.rk_life <= HH_ROCKET_MAX_LIFE;
    double rk_nlat = 0.0, rk_nlon = 0.0, rk_nhdg = 0.0;
    {
        const bool mv_a = snap && m.spd > 0.0;
        const bool any_rk = __ballot(rk_spec) != 0ULL;
        if (HH_USUAL(any_rk)) {
            double r_hdg = m.rk_hdg;
            const double r_cmd = m.rk_cmd;
            {
                const double delta = d_signed_heading_diff(r_hdg, r_cmd);
                const double stepped = r_hdg + (delta >= 0.0 ? HH_ROCKET_TURN_RATE : -HH_ROCKET_TURN_RATE);
                const double nh = hh_fabs(delta) <= HH_ROCKET_TURN_RATE ? r_cmd : stepped;
                r_hdg = r_hdg != r_cmd ? nh : r_hdg;
            }
            rk_nhdg = r_hdg;
            double a_lat, a_lon;
            d_geo_move2(m.lat, m.lon, m.hdg, mv_a ? m.spd * HH_KNOTS_TO_MS * 1.0 : 0.0, a_lat, a_lon,
                        rk_spec ? m.rk_lat : 5.0, rk_spec ? m.rk_lon : 7.0, r_hdg, rk_speed0 * HH_KNOTS_TO_MS * 1.0, rk_nlat, rk_nlon);
            if (mv_a) { m.lat = a_lat; m.lon = a_lon; }
        } else {
            if (mv_a) d_geo_move(m.lat, m.lon, m.hdg, m.spd * HH_KNOTS_TO_MS * 1.0, m.lat, m.lon);
        }
    }
    const double rk0_lat = rk_pre ? m.rk_lat : lat_old, rk0_lon = rk_pre ? m.rk_lon : lon_old;
    OPub pn = pub;
    oct_publish_vec(m, pn); /* heading vector after the turn: cannon prefilter now, published with the post-tick state */
    if (FULL) oct_publish_norm(c, m, pn);
    /* positions of the others after their move */
    double lat1[5], lon1[5];
    HH_O_FETCH5(d, lat1, m.lat);
    HH_O_FETCH5(d, lon1, m.lon);

    /* ---------------- phase Q: envelope tests that survive the prefilter -> workgroup queue ---------------- */
    const bool rk_maybe = running && L.exists && rk_pre;
    int q_total = 0;
    {
        const int t = m.ac_type;
        int push[5], pjs[5], any_push = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int pj = o_rel_pos(L, r);
            const bool snap_j = running && ((amask0 >> pj) & 1); /* not alive at tick start -> can never be "currently alive" */
            const bool enemy = r >= 2;
            const bool lower = pj < L.p; /* target already moved iff its id is lower (cmano_simulator.py:142) */
            const double tl = lower ? lat1[r] : tb.lat[r];
            const double to = lower ? lon1[r] : tb.lon[r];
            push[r] = (int)fired & (int)snap_j & ((int)(c.friendly_kill != 0) | (int)enemy) & (int)d_maybe_within_km(lat_old, lon_old, tl, to, HH_AC_CANNON_KM(t)) &
                      (int)!hh_cannon_cone_planar_outside(lat_old, lon_old, tl, to, pn.uc, pn.us, t);
            pjs[r] = pj;
            any_push |= push[r];
        }
        if (HH_RARE(o_any(any_push != 0))) {
#pragma unroll
            for (int r = 0; r < 5; r++) HH_O_PUSH(push[r] != 0, tid | (1 << 8) | (pjs[r] << 10));
        }
        if (o_any(rk_maybe)) { /* wave-uniform */
            const int tp = rk_maybe ? o_slot_pos(c, m.rk_target - 1) : ((1 - L.q) << 2);
            const int rt = o_pos_rel(L, tp);
            HH_O_PUSH(rk_maybe && d_maybe_within_km(rk0_lat, rk0_lon, o_sel5(lat1, rt), o_sel5(lon1, rt), HH_ROCKET_FUSE_KM), tid | (2 << 8) | (tp << 10));
            const int fp = o_slot_pos(c, s == 1 ? 0 : 1); /* rocket_unit.py:46: 1 if source.id == 2 else 2 */
            const bool self = fp == L.p;                  /* never for a source of the 3-vs-3 layout; n-vs-m keeps the reference's arithmetic */
            const int rf = self ? 0 : o_pos_rel(L, fp);
            const double fl_ = self ? m.lat : o_sel5(lat1, rf), fo_ = self ? m.lon : o_sel5(lon1, rf);
            HH_O_PUSH(rk_maybe && c.friendly_kill && d_maybe_within_km(rk0_lat, rk0_lon, fl_, fo_, HH_ROCKET_FUSE_KM), tid | (3 << 8) | (fp << 10));
        }
    }
    /* ---------------- phase I: dense pass over the queue ---------------- */
    int myres = 0;
    if (HH_RARE(q_total != 0)) { /* wave-uniform */
        sh.lat0[tid] = lat_old; sh.lon0[tid] = lon_old; sh.hdg[tid] = hdg_old;
        sh.flags[tid] = pub.flags;
        sh.u.t.lat1[tid] = m.lat; sh.u.t.lon1[tid] = m.lon; sh.u.t.hdg1[tid] = m.hdg;
        sh.u.t.rk_lat[tid] = rk0_lat; sh.u.t.rk_lon[tid] = rk0_lon;
        sh.res[tid] = 0;
        if (L.p == 0) sh.g_tkey[L.g] = ar.tkey;
        o_wave_sync();
        drain_envelope_queue_t<8, 64, IX>(sh, tid, q_total, c.nA + 1);
        o_wave_sync();
        myres = sh.res[tid];
    }
    const int rk_at_start = m.rk_alive;
    const int aux = fired ? (myres >> 1) & 0xff : 0; /* cannon hits by target lane position */
    int rkw = 0; /* bit0 present, bit1 fuse on target, bit2 fuse on "friendly", bit3 end of life, bits4-6 target position, bits 8.. seq */
    if (HH_USUAL(running && L.exists && rk_at_start)) {
        const int eol = m.rk_life > HH_ROCKET_MAX_LIFE;
        rkw = 1 | (((myres >> 9) & 1) << 1) | (((myres >> 10) & 1) << 2) | (eol << 3) | (o_slot_pos(c, m.rk_target - 1) << 4) | (m.rk_seq << 8);
    }

    /* ---------------- phases C + D: id-ordered resolution, computed identically by the lanes of the arena ---------------- */
    int alive = amask0, nev = 0, dead = 0;
    unsigned long long evpack = 0; /* 7 bits per event: killer position | victim position << 3 | by rocket << 6 */
    if (HH_RARE(__ballot(aux != 0 || (rkw & 0xe) != 0) != 0ULL)) { /* wave-uniform: some arena of the wave has something to resolve */
        int aux_[8], res_[8];
        aux_[0] = o_bc_i<0>(aux); aux_[1] = o_bc_i<1>(aux); aux_[2] = o_bc_i<2>(aux);
        aux_[4] = o_bc_i<4>(aux); aux_[5] = o_bc_i<5>(aux); aux_[6] = o_bc_i<6>(aux);
        res_[0] = o_bc_i<0>(rkw); res_[1] = o_bc_i<1>(rkw); res_[2] = o_bc_i<2>(rkw);
        res_[4] = o_bc_i<4>(rkw); res_[5] = o_bc_i<5>(rkw); res_[6] = o_bc_i<6>(rkw);
        aux_[3] = aux_[7] = res_[3] = res_[7] = 0;
        if (running) {
            /* aircraft phase: shooter i (alive at tick start, even if killed earlier in this tick) hits the still-alive targets
             * in id order (ac1.py:106-115) */
            if (aux_[0] | aux_[1] | aux_[2] | aux_[4] | aux_[5] | aux_[6]) {
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    if (i == 3) continue;
                    const int ci = aux_[i];
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        if (j == 3) continue;
                        if (((ci >> j) & 1) && ((alive >> j) & 1)) {
                            alive &= ~(1 << j);
                            evpack |= (unsigned long long)(i | (j << 3)) << (7 * nev);
                            nev++;
                        }
                    }
                }
            }
            /* rocket phase in launch order (rocket_unit.py:37-58) */
            int done_mask = 0;
#pragma unroll 1
            for (int k = 0; k < 6; k++) {
                int best = -1, best_seq = 0x7fffffff, w = 0;
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    if (j == 3) continue;
                    const int wj = res_[j];
                    if ((wj & 1) && !((done_mask >> j) & 1) && (wj >> 8) < best_seq) { best = j; best_seq = wj >> 8; w = wj; }
                }
                if (best < 0) break;
                done_mask |= 1 << best;
                const int tg = (w >> 4) & 7;
                const int fid = o_slot_pos(c, o_pos_slot(c, best) == 1 ? 0 : 1);
                if (((w >> 1) & 1) && ((alive >> tg) & 1)) {
                    alive &= ~(1 << tg); dead |= 1 << best;
                    evpack |= (unsigned long long)(best | (tg << 3) | (1 << 6)) << (7 * nev);
                    nev++;
                } else if (c.friendly_kill && ((alive >> fid) & 1) && ((w >> 2) & 1)) {
                    alive &= ~(1 << fid); dead |= 1 << best;
                    evpack |= (unsigned long long)(best | (fid << 3) | (1 << 6)) << (7 * nev);
                    nev++;
                } else if ((w >> 3) & 1) {
                    dead |= 1 << best;
                }
            }
        }
    }
    if (HH_USUAL(running && L.exists && rk_at_start)) { /* (usually some rocket of the wave is in flight: no any-lane test in front) */
        if ((dead >> L.p) & 1) {
            m.rk_alive = 0; m.rk_target = 0; m.rk_life = 0; m.rk_seq = 0;
            m.rk_lat = m.rk_lon = m.rk_hdg = m.rk_cmd = 0.0;
        } else { /* commit the speculative turn + move (rocket_unit.py:61-73) */
            m.rk_hdg = rk_nhdg; m.rk_lat = rk_nlat; m.rk_lon = rk_nlon;
            m.rk_life += 1;
        }
    }

    /* ---------------- phase E: out of bounds, rewards (env_base.py:240-310) ---------------- */
    int oob = 0;
    {
        const int al = (alive >> L.p) & 1;
        const bool inb = (HH_MAP_LON0 <= m.lon) & (m.lon <= c.lon_hi) & (HH_MAP_LAT0 <= m.lat) & (m.lat <= c.lat_hi);
        oob = (L.exists & running & (al != 0) & !inb) ? 1 : 0;
        m.alive = L.exists ? (oob ? 0 : al) : m.alive;
    }
    const int oobm = oct_arena_bits(__ballot(oob != 0), L);
    double rews = 0.0;
    int destroyed = 0;
    if (HH_RARE(o_any((nev > 0) | (oob != 0)))) { /* kills and removals are rare */
    if (running && L.exists && agent) {
        const double sc = c.rew_scale;
        if (oob) { rews += -2.0 * sc; destroyed = 1; }
        for (int e = 0; e < nev; e++) {
            const int w = (int)(evpack >> (7 * e)) & 0x7f;
            const int kp = w & 7, dp = (w >> 3) & 7;
            if (kp < 4) {
                if (dp >= 4) { if (kp == L.p) rews += 1.0; }
            } else if (dp < 4) {
                if (dp == L.p) { rews += -1.0 * sc; destroyed = 1; }
            }
        }
    }
    for (int e = 0; e < nev; e++) { /* event masks for parity checks (by unit slot) */
        const int w = (int)(evpack >> (7 * e)) & 0x7f;
        const int ds = o_pos_slot(c, (w >> 3) & 7);
        evm |= ((w >> 6) & 1) ? (1u << (8 + ds)) : (1u << ds);
    }
    if (oob) evm |= 1u << (16 + s);
    }
    ev_mask_out = evm;
    /* post-tick state */
    oct_publish_flags(m, pn);
    pub = pn;
    if (FULL) oct_tables<true>(m, pub, L, tb);
    else {
#pragma unroll
        for (int r = 0; r < 5; r++) { tb.lat[r] = lat1[r]; tb.lon[r] = lon1[r]; }
        tb.amask = oct_arena_bits(__ballot(m.alive != 0), L);
    }
    {
        const int ke = ((nev > 0) | (oobm != 0)) ? 1 : 0;
        out.kill_event = running ? ke : out.kill_event;
    }
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    if (c.glob_frac > 0.0) { r0 = o_qbc_d<0>(rews); r1 = o_qbc_d<1>(rews); r2 = o_qbc_d<2>(rews); } /* uniform: the agents' quad */
    {
        double add = rews;
        if (c.glob_frac > 0.0) { /* wave-uniform: configuration */
            /* env_base.py:268-276 shared reward: the other agents' event rewards in id order */
            double other = 0.0;
            if (0 < c.nA && L.i != 0) other += r0;
            if (1 < c.nA && L.i != 1) other += r1;
            if (2 < c.nA && L.i != 2) other += r2;
            add = rews + c.glob_frac * other;
        }
        const double nr = out.reward + add;
        const bool give = running & L.exists & agent & ((m.alive != 0) | (destroyed != 0));
        out.reward = give ? nr : out.reward;
    }
}

/* ---- the phases of a commander step on the register table (hl_do_begin / hl_do_tick / hl_do_end of hh_kernels_hier.h) ---- */

/* HL_BEGIN: env_hier.py:142-190 _action_assess + the opponents' draws; needs the full table */
__device__ __forceinline__ void oct_do_begin(const DevCfg &c, const OTab &tb, const OLane &L, int n, bool active, HlLane &H, const int8_t *__restrict__ cmd) {
    H.ar.hl_s = 0;
    H.ar.hl_run = active && !H.ar.done;
    H.acc = 0.0;
    if (H.ar.hl_run && L.exists) {
        const int cc = L.q == 0 ? (int)cmd[(size_t)n * c.nA + L.s] : 0;
        const double r = oct_action_assess(c, tb, L, H.m, H.ar, cc);
        if (L.q == 0) H.acc = r;
    }
}

/* do_tick + rewards + kill / surrounding events + s += 1 (env_hier.py:125-138).  Returns 1 iff this lane's arena ran the tick. */
template <bool IX, bool FULL>
__device__ __forceinline__ int oct_do_tick(const DevPtrs &P, const DevCfg &c, OctShared &sh, int tid, const OLane &L, int n, bool active, HlLane &H,
                                           OTab &tb, OPub &pub) {
    StepOut so;
    so.reward = 0.0; so.valid = 0; so.opp_stat0 = 0.0;
    const bool was_running = active && H.ar.hl_run;
    uint32_t evm_tick = 0;
    tick_oct<IX, FULL>(c, sh, tid, L, was_running, H.m, H.ar, tb, pub, so, evm_tick);
    H.evm |= evm_tick;
    /* env_hier.py:133-135: the surrounding event is looked for only after min_sub_steps (s > 10).  Without FULL the table is
     * built here, on the ticks that read it (wave-uniform: some arena of the wave is that far into its macro step) */
    const bool look = was_running && H.ar.hl_s > 10;
#ifndef HHO_ABL_NO_EVT /* tuning builds only (tools/build_variant.sh): WRONG RESULTS on purpose — what does the surrounding event's table (sub-steps 11..15) cost? */
    if (__ballot(look)) {
        if (!FULL) oct_tables<false>(H.m, pub, L, tb);
    }
#endif
    int near_ = 0;
    if (look && L.exists && L.q == 0 && H.m.alive) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if (!((tb.amask >> (4 + o_rel_index(L, 2 + e))) & 1)) continue;
            if (tb.dist[2 + e] < 0.1 && (tb.foc[2 + e] < 15.0 || tb.focr[2 + e] < 15.0)) near_ = 1;
        }
    }
    const int situ = oct_arena_bits(__ballot(near_ != 0), L) != 0;
    if (was_running) {
        if (L.q == 0 && L.exists) H.acc += so.reward;
        H.ar.hl_s += 1;
        H.ar.steps += 1;
        arena_rekey(H.ar);
        H.ar.hl_run = (H.ar.hl_s <= 15 && !so.kill_event && !situ) ? 1 : 0;
        if (L.exists) trace_append(P, 6, n, L.s, H.m, H.ar, H.tcur);
    }
    return was_running ? 1 : 0;
}

/* HL_END: done, rewards out, episode statistics, eval counters, auto-reset, commander observation + stored target lists
 * (env_hier.py:49-98, env_base.py:91-107); the agents' rows are left staged in sh.u.obs.  `tb` must hold the full table. */
__device__ __forceinline__ void oct_do_end(const DevPtrs &P, const DevCfg &c, OctShared &sh, int tid, const OLane &L, int n, bool active, HlLane &H,
                                           OTab &tb, OPub &pub, int phase, float *__restrict__ reward_out, uint8_t *__restrict__ valid_out,
                                           uint8_t *__restrict__ done_out, const uint8_t *__restrict__ mask) {
    const bool agent = L.q == 0;
    Unit &m = H.m;
    Arena &ar = H.ar;
    const bool ending = phase == HH_HL_END && active && !ar.done; /* arena took part in this macro step */
    const int ag = __popc(tb.amask & 0x07), op = __popc(tb.amask & 0x70);
    if (ending) ar.done = (ag <= 0 || op <= 0 || ar.steps >= c.horizon) ? 1 : 0;
    {   /* episode return: the agents' accumulated rewards in id order (every lane of the agents' quad keeps the same sum) */
        const double a_ = (ending && agent && L.exists) ? H.acc : 0.0;
        const double a0 = o_qbc_d<0>(a_), a1 = o_qbc_d<1>(a_), a2 = o_qbc_d<2>(a_);
        if (ending && L.p == 0) {
            if (0 < c.nA) H.ep_ret += a0;
            if (1 < c.nA) H.ep_ret += a1;
            if (2 < c.nA) H.ep_ret += a2;
            if (ar.done) {
                P.last_ret[n] = (float)H.ep_ret;
                P.last_len[n] = ar.steps;
                P.last_outcome[n] = (op <= 0 && ar.steps < c.horizon) ? 1 : ((ag <= 0 && ar.steps < c.horizon) ? -1 : 0);
            }
        }
    }
    if (phase == HH_HL_END) { /* eval_info of this commander step (env_base.py:91-107): units that still exist, by assessed commander action */
        const bool ex = ending && L.exists && m.alive;
        const int al = oct_arena_bits(__ballot(ex), L);
        const int b1 = oct_arena_bits(__ballot(ex && (m.cmd_act & 1)), L), b2 = oct_arena_bits(__ballot(ex && (m.cmd_act & 2)), L);
        if (active && L.p == 0) {
            int e[HH_EVAL_K];
#pragma unroll
            for (int k = 0; k < HH_EVAL_K; k++) e[k] = 0;
            if (ending) {
                const int A_ = 0x07, O_ = 0x70, v = b1 | b2;
                e[0] = (op <= 0 && ar.steps < c.horizon) ? 1 : 0;
                e[1] = (ag <= 0 && ar.steps < c.horizon) ? 1 : 0;
                e[2] = (ar.steps >= c.horizon && ag > 0 && op > 0) ? 1 : 0;
                e[3] = __popc(al & A_ & v);
                e[4] = __popc(al & A_ & ~v);
                e[5] = __popc(al & O_ & v);
                e[6] = __popc(al & O_ & ~v);
                e[7] = __popc(al & A_);
                e[8] = __popc(al & O_);
                e[9] = __popc(al & A_ & b1 & ~b2);
                e[10] = __popc(al & A_ & b2 & ~b1);
                e[11] = __popc(al & A_ & b1 & b2);
            }
#pragma unroll
            for (int k = 0; k < HH_EVAL_K; k++) {
                P.eval_last[(size_t)n * HH_EVAL_K + k] = e[k];
                if (e[k]) P.eval_tot[(size_t)n * HH_EVAL_K + k] += e[k];
            }
        }
    }
    if (phase == HH_HL_END) {
        ar.hl_run = 0;
        if (active && agent && L.exists) {
            const size_t o = (size_t)n * c.nA + L.s;
            if (reward_out) reward_out[o] = ending ? (float)H.acc : 0.0f;
            if (valid_out) valid_out[o] = ending ? 1 : 0; /* every agent id has a reward key (env_hier.py:154,188) */
        }
        if (active && L.p == 0 && done_out) done_out[n] = (uint8_t)ar.done;
    }
    const bool need_reset = phase == HH_HL_END ? (active && ar.done && c.auto_reset)
                                               : (phase == HH_HL_RESET && active && (mask == nullptr || mask[n]));
    if (HH_RARE(__ballot(need_reset) != 0ULL)) { /* wave-uniform */
        if (need_reset) {
            reset_arena_scalars(ar);
            if (L.exists) {
                reset_unit<6>(c, L.s, m, ar);
                trace_append(P, 6, n, L.s, m, ar, H.tcur); /* first row of the new episode */
            }
            ar.hl_s = 0; ar.hl_run = 0;
            H.ep_ret = 0.0;
            H.acc = 0.0;
        }
        oct_publish_vec(m, pub);
        oct_publish_norm(c, m, pub);
        oct_publish_flags(m, pub);
        oct_tables<true>(m, pub, L, tb);
    }
    if (active && L.exists) { /* rows of the agents staged in LDS (opponents only refresh their stored target lists) */
        float *row = agent ? &sh.u.obs[(L.g * c.nA + L.s) * HH_OBS_HL] : &sh.u.obs[0];
        oct_commander_obs(c, tb, pub, L, m, row);
    }
    o_wave_sync();
}

/* ---- the persistent macro step (hh_hl_rollout): ONE launch per commander step, pilot actions from a tape [16][N][6][4] ---- */
template <int W, bool HLD>
__global__ __launch_bounds__(64, W) void hh_k_hier_macro_oct(DevPtrs P, DevCfg c_in, const int8_t *__restrict__ cmd, const int8_t *__restrict__ tape,
                                                           float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                           uint8_t *__restrict__ valid_out, uint8_t *__restrict__ done_out,
                                                           int *__restrict__ counters) {
    DevCfg c_hl = c_in;
    hh_cfg_set_hl_default(c_hl); /* HLD: the default HighLevelEnv configuration as literals (hh_device.h) */
    const DevCfg &c = HLD ? c_hl : c_in;
    __shared__ OctShared sh;
    const int tid = threadIdx.x;
    OLane L;
    L.g = tid >> 3; L.p = tid & 7; L.q = (tid >> 2) & 1; L.i = tid & 3;
    L.base = tid & ~7;
    const int n = blockIdx.x * 8 + L.g;
    const bool active = n < c.N;
    L.exists = active && L.i < (L.q ? c.nO : c.nA);
    L.s = L.q ? c.nA + L.i : L.i;
    const size_t U = (size_t)c.N * 6;
    const size_t u = (size_t)n * 6 + L.s;
    HlLane H;
    H.m = Unit{};
    H.m.ac_type = 2; H.m.cannon_max = 1; /* lanes without an aircraft: never alive, harmless operands */
    H.ar = Arena{};
    H.acc = 0.0; H.ep_ret = 0.0; H.evm = 0;
    H.tcur = (P.trace != nullptr && active && n < P.trace_K) ? P.trace_pos[n] : 0;
    if (L.exists) unit_load(P, U, u, H.m);
    if (active) {
        arena_load(P, c, n, H.ar);
        if (L.p == 0) H.ep_ret = P.ep_ret[n];
    } else {
        H.ar.done = 1;
    }
    {
        const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
        if (tid < 11) sh.rk_speed[tid] = speed_table[tid];
    }
    OPub pub;
    OTab tb;
    oct_publish_vec(H.m, pub);
    oct_publish_norm(c, H.m, pub);
    oct_publish_flags(H.m, pub);
    oct_tables<true>(H.m, pub, L, tb);
#ifndef HHO_ABL_NO_BEGIN
    oct_do_begin(c, tb, L, n, active, H, cmd);
#endif
    int ticks = 0;
    uint32_t evm_last = 0;
    int act_fault = 0; /* a consumed action word was out of range and ran sanitised (hh_act_unpack) */
    /* the action word of the next sub-step is requested a sub-step ahead (one wave per SIMD cannot hide the round trip) */
    int act_next = L.exists ? *reinterpret_cast<const int *>(tape + u * 4) : 0;
    o_wave_sync();
    bool tab = true; /* the table in registers describes the current positions (wave-uniform) */
    for (int sub = 0; sub < 16; sub++) {
        if (!__any(H.ar.hl_run)) break; /* nobody in this wave is inside a macro step any more */
        /* two waves per SIMD (256 registers): what derives from the lane index is recomputed per sub-step instead of living in registers across the loop —
         * 236 -> 206 spilled registers, +3 % at 65536 arenas; at one wave per SIMD the recomputation costs more than the AGPR copies it saves (-1.5 % at 8192) */
        int tid_l = threadIdx.x;
        if constexpr (W >= 2) asm volatile("" : "+v"(tid_l));
        OLane L;
        L.g = tid_l >> 3; L.p = tid_l & 7; L.q = (tid_l >> 2) & 1; L.i = tid_l & 3;
        L.base = tid_l & ~7;
        const int n = blockIdx.x * 8 + L.g;
        const bool active = n < c.N;
        L.exists = active && L.i < (L.q ? c.nO : c.nA);
        L.s = L.q ? c.nA + L.i : L.i;
        const size_t u = (size_t)n * 6 + L.s;
        const int tid = tid_l;
        const int w = act_next;
        if (L.exists && sub + 1 < 16) act_next = *reinterpret_cast<const int *>(tape + ((size_t)(sub + 1) * U + u) * 4);
        int8_t act[4];
        const bool running = active && H.ar.hl_run;
        hh_act_unpack(w, act, act_fault, running && L.exists && H.m.alive);
        if (running) H.evm = 0;
        /* both sides' _take_base_action in one pass: a side's action reads nothing the other side's action writes (positions do not
         * move, the missile_wait draws are keyed by unit), and launches are numbered in unit id order either way */
#ifndef HHO_ABL_NO_ACT0
        if (HH_RARE(tab)) act_oct<(W >= 2), true>(c, sh, tid, L, running, H.m, H.ar, act, true, tb, pub, H.evm); /* the first sub-step only */
        else
#endif
        act_oct<(W >= 2), false>(c, sh, tid, L, running, H.m, H.ar, act, true, tb, pub, H.evm);
        ticks += oct_do_tick<(W >= 2), false>(P, c, sh, tid, L, n, active, H, tb, pub);
        tab = false;
        if (running) evm_last = H.evm;
    }
#ifndef HHO_ABL_NO_END_TABLE /* tuning builds only: WRONG RESULTS on purpose — the full pair table in front of the commander observation (what an output wave could take) */
    oct_publish_norm(c, H.m, pub);
    oct_tables<true>(H.m, pub, L, tb);
#endif
#ifndef HHO_ABL_NO_END
    oct_do_end(P, c, sh, tid, L, n, active, H, tb, pub, HH_HL_END, reward_out, valid_out, done_out, nullptr);
#endif
    if (obs_out) { /* the workgroup's agent rows are contiguous in [N, nA, 34] */
        const int arenas = min(8, c.N - (int)blockIdx.x * 8);
        const int cnt = arenas * c.nA * HH_OBS_HL;
        float *dst = obs_out + (size_t)blockIdx.x * 8 * c.nA * HH_OBS_HL;
        for (int k = tid; k < cnt; k += 64) dst[k] = sh.u.obs[k];
    }
    if (L.exists) {
        unit_store(P, U, u, H.m);
        P.acc_rew[u] = H.acc;
    }
    if (active && L.p == 0) {
        if (P.trace != nullptr && n < P.trace_K) P.trace_pos[n] = H.tcur;
        arena_store(P, n, H.ar);
        P.ep_ret[n] = H.ep_ret;
    }
    {   /* cumulative arena-ticks (hh_hl_tick_count): one atomic per wave */
        int t = L.p == 0 ? ticks : 0;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (tid == 0 && t && counters) atomicAdd(reinterpret_cast<unsigned long long *>(counters + 2), (unsigned long long)t);
    }
    /* event masks of each arena's last sub-step, like the phase path leaves them */
    if (active && L.p == 0 && ticks) P.ev_mask[n] = 0;
    __syncthreads();
    if (L.exists && ticks && evm_last) atomicOr(&P.ev_mask[n], evm_last);
    hh_act_fault_commit(P, n, L.exists, act_fault);
}

/* which aircraft's weapon flags (env_base.py:208-211 "shot" entry of opp_ac_values) the lane's pilot row carries: relation of the (up to two)
 * observed aircraft, -1 = none, and the index of their flag entry in the row (the variant-row form patches exactly those entries) */
struct OVar { int ra, rb, fa, fb; };
/* env_hier.py:100-112 lowlevel_state of the lane's unit -> 30 floats (zero padded) + policy type (hl_pilot_obs of hh_kernels_hier.h) */
__device__ __forceinline__ int oct_pilot_obs(const DevCfg &c, const OTab &t, const OPub &p, const OLane &L, const Unit &m, float *out, OVar *ov = nullptr) {
    for (int k = 0; k < 30; k++) out[k] = 0.0f;
    Near3 fr;
    oct_nearby(c, t, L, true, fr);
    int n = 0;
    out[n++] = p.nlat;
    out[n++] = p.nlon;
    out[n++] = p.nspd;
    out[n++] = p.nhdg;
    int mode;
    const int cmd_act = m.cmd_act, n_tgt = m.n_tgt, t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2;
    const double d0 = m.tgt_d0, d1 = m.tgt_d1, d2 = m.tgt_d2;
    const int r0 = t0 ? o_pos_rel(L, o_slot_pos(c, t0 - 1)) : 2, r1 = t1 ? o_pos_rel(L, o_slot_pos(c, t1 - 1)) : 2;
    if (cmd_act != 0) { /* fight the commander-chosen target, with the stale stored distance */
        mode = 1;
        double dist = d0;
        int r = r0;
        if (cmd_act == 2) { dist = d1; r = r1; }
        if (cmd_act >= 3) { dist = d2; r = t2 ? o_pos_rel(L, o_slot_pos(c, t2 - 1)) : 2; }
        out[n++] = (float)norm180(o_sel5(t.foc, r));
        out[n++] = (float)aspect(o_sel5(t.focr, r));
        out[n++] = (float)o_sel3v(t.hd[0], t.hd[1], t.hd[2], r - 2);
        out[n++] = (float)dist;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) {
            out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
            out[n++] = m.missile_wait == 0 ? 1.0f : 0.0f;
            out[n++] = (m.has_missile || m.burst > 0) ? 1.0f : 0.0f;
        } else {
            out[n++] = m.burst > 0 ? 1.0f : 0.0f;
        }
        if (ov) { ov->ra = r; ov->fa = n + 8; ov->rb = -1; ov->fb = 0; }
        n += oct_opp_block(t, 0, r, dist, out + n);
    } else {
        mode = 2;
        out[n++] = (float)hh_clip((double)m.cannon_remain / (double)m.cannon_max, 0.0, 1.0);
        if (m.ac_type == 1) out[n++] = (float)hh_clip((double)m.missile_remain / (double)m.rocket_max, 0.0, 1.0);
        out[n++] = (p.flags & FL_SHOT) ? 1.0f : 0.0f;
        if (n_tgt >= 1) oct_opp_block(t, 1, r0, d0, out + n);
        if (n_tgt >= 2) oct_opp_block(t, 1, r1, d1, out + n + 9);
        if (ov) { ov->ra = n_tgt >= 1 ? r0 : -1; ov->fa = n + 8; ov->rb = n_tgt >= 2 ? r1 : -1; ov->fb = n + 17; }
        n += 18;
    }
    if (fr.n) oct_friend_block(c, t, fr.i0, out + n);
    return mode;
}

/* ---- the phases one launch each, for callers whose pilot networks run BETWEEN them (hh_k_hier of hh_kernels_hier.h on the
 * register table): HL_BEGIN / HL_AGENTS_ACT / HL_TICK / HL_END.  Same results, bit for bit; HL_REFRESH / HL_RESET stay on the
 * generic kernel (the state in HBM is the same). ---- */
/* one phase for the eight arenas grp * 8 .. grp * 8 + 7, executed by ONE wave (tid = lane): the body of hh_k_hier_oct */
/* VAR: the variant-row form (hh_k_hier_oct_v, phases HH_HL_BEGIN_V / HH_HL_ACT_TICK).  The reference lets the opponents' pilots observe the agents' weapon
 * flags of the SAME sub-step (env_base.py:208-211: units act in id order), which is why the standard path needs a launch and a policy call per side.
 * An agent's act can only RAISE its flag (arm the cannon, launch a missile), and an opponent's row carries the flag of at most two agents, so the
 * launch that ends a sub-step emits, next to every opponent's row, the copies with the observed agents' still-zero flags forced to one (up to three
 * variants, `vrow` slots 3 + 4 j + v); ONE policy call evaluates the agents' rows and all variants, and the next launch lets the agents act, looks at
 * which flags rose, takes each opponent's action from the matching variant and runs the tick.  Same rows through the same networks: same results. */
#ifndef HHV_IX
#define HHV_IX(W) false /* the variant-row kernels call the exact envelope test out of line at either occupancy (tuning builds: -DHHV_IX(W)=... ) */
#endif
template <int W, bool VAR = false, int PHASE = -1> /* PHASE >= 0: the phase is known at compile time (the variant-row kernels: one instance per phase, each holding only its own path) */
__device__ __forceinline__ void oct_phase_body(const DevPtrs &P, const DevCfg &c, int phase_rt, int grp, int tid, OctShared &sh, const int8_t *__restrict__ cmd,
                                               const int8_t *__restrict__ actions, float *__restrict__ pilot_obs, uint8_t *__restrict__ pilot_mode,
                                               float *__restrict__ obs_out, float *__restrict__ reward_out, uint8_t *__restrict__ valid_out,
                                               uint8_t *__restrict__ done_out, int *__restrict__ running_count,
                                               unsigned long long *__restrict__ tick_total, float *vrow = nullptr, HhTl *tl = nullptr) {
    const int phase = PHASE >= 0 ? PHASE : phase_rt;
    OLane L;
    L.g = tid >> 3; L.p = tid & 7; L.q = (tid >> 2) & 1; L.i = tid & 3;
    L.base = tid & ~7;
    const int n = grp * 8 + L.g;
    const bool active = n < c.N;
    L.exists = active && L.i < (L.q ? c.nO : c.nA);
    L.s = L.q ? c.nA + L.i : L.i;
    const bool agent = L.q == 0;
    const size_t U = (size_t)c.N * 6;
    const size_t u = (size_t)n * 6 + L.s;
    HlLane H;
    H.m = Unit{};
    H.m.ac_type = 2; H.m.cannon_max = 1;
    H.ar = Arena{};
    H.acc = 0.0; H.ep_ret = 0.0; H.evm = 0;
    H.tcur = (P.trace != nullptr && active && n < P.trace_K) ? P.trace_pos[n] : 0;
    Unit &m = H.m;
    Arena &ar = H.ar;
    if (L.exists) { unit_load(P, U, u, m); H.acc = P.acc_rew[u]; }
    if (active) {
        arena_load(P, c, n, ar);
        if (L.p == 0) H.ep_ret = P.ep_ret[n];
    } else {
        ar.done = 1;
    }
    {
        const double speed_table[11] = HH_ROCKET_SPEED_TABLE;
        if (tid < 11) sh.rk_speed[tid] = speed_table[tid];
    }
    OPub pub;
    OTab tb;
    oct_publish_vec(m, pub);
    oct_publish_norm(c, m, pub);
    oct_publish_flags(m, pub);
    if (phase == HH_HL_TICK || (VAR && phase == HH_HL_ACT_TICK)) oct_positions(m, L, tb); /* the tick builds its table afterwards; before it only a launch test may ask for an entry */
    else oct_tables<true>(m, pub, L, tb);
    o_wave_sync();
#ifdef HH_TIMELINE
    if (tl) hh_tl_mark(*tl, 0); /* state loaded, published, positions / table built */
#endif
    int obs_side = -1; /* which side's pilot observations this launch emits */
    int act_fault = 0;  /* a consumed action word was out of range and ran sanitised (hh_act_unpack) */
    int ran_tick = 0;   /* HL_ACT_TICK: the lane's arena ran the tick in this launch */
    /* a bound policy bank (hh_bind_policy): this launch's pilot rows are binned by network here (see hh_k_hier) */
    int pslot = 0;
    HhBinTicket bt{0, 0};
    auto bin_issue = [&](int side) {
        const bool mine_ = side == 0 ? agent : !agent;
        const int sb = (L.exists && ar.hl_run && m.alive && mine_) ? hl_selector(c, m.cmd_act != 0 ? 1 : 2, m.ac_type, agent) : 0;
        pslot = sb ? (int)P.pol_lut[sb] : 0;
        bt = hh_bin_rows_issue(P.pol_counts, pslot);
    };
    if (VAR && phase == HH_HL_BEGIN_V) {
        oct_do_begin(c, tb, L, n, active, H, cmd);
        obs_side = 2;
    } else if (VAR && phase == HH_HL_ACT_TICK) {
        const size_t row0 = (size_t)n * HH_HL_VROWS;
        const bool running = active && ar.hl_run;
        int8_t act[4];
        hl_load_act(actions, row0 + L.i, L.exists && agent, act, act_fault, running && m.alive && agent);
        int flb[5]; /* the flags the emitted rows were built from */
        HH_O_FETCH5(i, flb, pub.flags);
        act_oct<HHV_IX(W), false>(c, sh, tid, L, running, m, ar, act, agent, tb, pub, H.evm);
        HH_O_FETCH5(i, tb.fl, pub.flags);
        int v = 0; /* which of its rows describes what the opponent sees now: bit 0 / 1 = the first / second observed agent raised its flag */
        {
            const int cmd_act = m.cmd_act, n_tgt = m.n_tgt, t0 = m.tgt0, t1 = m.tgt1, t2 = m.tgt2;
            const int r0 = t0 ? o_pos_rel(L, o_slot_pos(c, t0 - 1)) : 2, r1 = t1 ? o_pos_rel(L, o_slot_pos(c, t1 - 1)) : 2;
            int ra, rb = -1;
            if (cmd_act != 0) { ra = r0; if (cmd_act == 2) ra = r1; if (cmd_act >= 3) ra = t2 ? o_pos_rel(L, o_slot_pos(c, t2 - 1)) : 2; }
            else { ra = n_tgt >= 1 ? r0 : -1; rb = n_tgt >= 2 ? r1 : -1; }
            if (ra >= 0 && ((o_sel5(tb.fl, ra) & ~o_sel5(flb, ra)) & FL_SHOT)) v |= 1;
            if (rb >= 0 && ((o_sel5(tb.fl, rb) & ~o_sel5(flb, rb)) & FL_SHOT)) v |= 2;
        }
        int8_t act2[4];
        hl_load_act(actions, row0 + 3 + 4 * L.i + v, L.exists && !agent, act2, act_fault, running && m.alive && !agent);
        act_oct<HHV_IX(W), false>(c, sh, tid, L, running, m, ar, act2, !agent, tb, pub, H.evm);
#ifdef HH_TIMELINE
        if (tl) hh_tl_mark(*tl, 1); /* both sides acted */
#endif
        const int ran = oct_do_tick<HHV_IX(W), true>(P, c, sh, tid, L, n, active, H, tb, pub);
#ifdef HH_TIMELINE
        if (tl) hh_tl_mark(*tl, 2); /* tick, rewards, events */
#endif
        ran_tick = ran;
        if (L.p == 0 && ran && ar.hl_run && running_count) atomicAdd(running_count, 1);
        {
            const unsigned long long rn = __ballot(ran && L.p == 0);
            if (rn && tid == 0 && tick_total) atomicAdd(tick_total, (unsigned long long)__popcll(rn));
        }
        obs_side = 2;
    } else if (phase == HH_HL_BEGIN) {
        oct_do_begin(c, tb, L, n, active, H, cmd);
        obs_side = 0;
        if (P.pol_lut && pilot_obs) bin_issue(0);
    } else if (phase == HH_HL_AGENTS_ACT) {
        int8_t act[4];
        hl_load_act(actions, u, L.exists, act, act_fault, active && ar.hl_run && m.alive && agent);
        if (P.pol_lut && pilot_obs) bin_issue(1);
        const bool running = active && ar.hl_run;
        act_oct<(W >= 2), true>(c, sh, tid, L, running, m, ar, act, agent, tb, pub, H.evm);
        HH_O_FETCH5(i, tb.fl, pub.flags); /* the opponents observe the agents' weapon flags of this sub-step (env_base.py:208-211) */
        obs_side = 1;
    } else if (phase == HH_HL_TICK) {
        int8_t act[4];
        hl_load_act(actions, u, L.exists, act, act_fault, active && ar.hl_run && m.alive && !agent);
        const bool running = active && ar.hl_run;
        act_oct<(W >= 2), false>(c, sh, tid, L, running, m, ar, act, !agent, tb, pub, H.evm);
        const int ran = oct_do_tick<(W >= 2), true>(P, c, sh, tid, L, n, active, H, tb, pub);
        if (L.p == 0 && ran && ar.hl_run && running_count) atomicAdd(running_count, 1);
        {   /* cumulative arena-ticks of this world (hh_hl_tick_count): one atomic per wave */
            const unsigned long long rn = __ballot(ran && L.p == 0);
            if (rn && tid == 0 && tick_total) atomicAdd(tick_total, (unsigned long long)__popcll(rn));
        }
        obs_side = 0;
        if (P.pol_lut && pilot_obs) bin_issue(0);
    } else { /* HH_HL_END */
        if (P.pol_lut && grp == 0 && tid <= 8) P.pol_counts[tid * HH_BIN_STRIDE] = 0; /* rows the last tick binned and nobody consumed */
        oct_do_end(P, c, sh, tid, L, n, active, H, tb, pub, HH_HL_END, reward_out, valid_out, done_out, nullptr);
        if (obs_out) {
            const int arenas = min(8, c.N - grp * 8);
            const int cnt = arenas * c.nA * HH_OBS_HL;
            float *dst = obs_out + (size_t)grp * 8 * c.nA * HH_OBS_HL;
            for (int k = tid; k < cnt; k += 64) dst[k] = sh.u.obs[k];
        }
    }
    if (VAR && obs_side == 2 && pilot_obs) {
        /* both sides' rows of the next sub-step: agents in slots 0..2, opponent j's variant v in slot 3 + 4 j + v */
        o_wave_sync(); /* the queue's exchange area is free */
        if (HH_RARE(c.nA + c.nO < 6)) {
            for (int k = tid; k < 8 * HH_HL_VROWS * 30; k += 64) vrow[k] = 0.0f;
            o_wave_sync();
        }
        const int slot0 = agent ? L.i : 3 + 4 * L.i;
        int sel = 0, vmask = 0;
        if (L.exists) {
            float *rw = &vrow[(L.g * HH_HL_VROWS + slot0) * 30];
            OVar ov{-1, -1, 0, 0};
            int mode = 0;
            if (ar.hl_run && m.alive) mode = oct_pilot_obs(c, tb, pub, L, m, rw, &ov);
            else for (int k = 0; k < 30; k++) rw[k] = 0.0f;
            sel = mode ? hl_selector(c, mode, m.ac_type, agent) : 0;
            vmask = sel ? 1 : 0;
            if (!agent) {
                /* a variant exists where the observed agent can still raise its flag: alive (it acts) and the flag down */
                const int fa_ = ov.ra >= 0 ? o_sel5(tb.fl, ov.ra) : 0, fb_ = ov.rb >= 0 ? o_sel5(tb.fl, ov.rb) : 0;
                const bool a0 = sel && ov.ra >= 0 && (fa_ & FL_ALIVE) && !(fa_ & FL_SHOT);
                const bool a1 = sel && ov.rb >= 0 && (fb_ & FL_ALIVE) && !(fb_ & FL_SHOT);
                vmask |= (a0 ? 2 : 0) | (a1 ? 4 : 0) | ((a0 && a1) ? 8 : 0);
                for (int k = 0; k < 30; k++) { /* slots without a variant carry the copy too: deterministic buffers, never listed */
                    const float x = rw[k];
                    rw[30 + k] = (a0 && k == ov.fa) ? 1.0f : x;
                    rw[60 + k] = (a1 && k == ov.fb) ? 1.0f : x;
                    rw[90 + k] = ((a0 && k == ov.fa) || (a1 && k == ov.fb)) ? 1.0f : x;
                }
            }
            if (pilot_mode) {
                uint8_t *pmq = pilot_mode + (size_t)n * HH_HL_VROWS + slot0;
                pmq[0] = (uint8_t)sel;
                if (!agent) { pmq[1] = (uint8_t)((vmask & 2) ? sel : 0); pmq[2] = (uint8_t)((vmask & 4) ? sel : 0); pmq[3] = (uint8_t)((vmask & 8) ? sel : 0); }
            }
        }
        if (HH_RARE(active && pilot_mode && c.nA + c.nO < 6 && L.p == 3)) { /* the bytes of the slots without an aircraft */
            for (int sl = c.nA; sl < 3; sl++) pilot_mode[(size_t)n * HH_HL_VROWS + sl] = 0;
            for (int sl = c.nO; sl < 3; sl++) for (int q = 0; q < 4; q++) pilot_mode[(size_t)n * HH_HL_VROWS + 3 + 4 * sl + q] = 0;
        }
        const int pslot_v = (P.pol_lut && sel) ? (int)P.pol_lut[sel] : 0;
        HhBinTicket4 bt4{0, {0, 0, 0, 0}};
        if (P.pol_lut) bt4 = hh_bin_rows_issue4(P.pol_counts, pslot_v, vmask);
        o_wave_sync();
        const int arenas = min(8, c.N - grp * 8);
        const int cnt = arenas * HH_HL_VROWS * 30; /* 450 floats per arena: a multiple of two */
        float *dst = pilot_obs + (size_t)grp * 8 * HH_HL_VROWS * 30;
        if (HH_USUAL((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (cnt & 3) == 0)) {
            const float4 *src4 = reinterpret_cast<const float4 *>(vrow);
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
            for (int k = tid; k < (cnt >> 2); k += 64) dst4[k] = src4[k];
        } else {
            for (int k = tid; k < cnt; k += 64) dst[k] = vrow[k];
        }
        if (P.pol_lut) hh_bin_rows_finish4(bt4, P.pol_lists, P.pol_max_rows, (int)((size_t)n * HH_HL_VROWS + slot0), pslot_v, vmask);
#ifdef HH_TIMELINE
        if (tl) hh_tl_mark(*tl, 3); /* rows built, stored, listed */
#endif
    } else if (obs_side >= 0 && pilot_obs) {
        const bool mine = obs_side == 0 ? agent : !agent;
        o_wave_sync(); /* the queue's exchange area (same LDS) is free */
        if (HH_RARE(c.nA + c.nO < 6)) { /* n-vs-m: the rows of the slots without an aircraft */
            for (int k = tid; k < 8 * 6 * 30; k += 64) sh.u.prow[k] = 0.0f;
            o_wave_sync();
        }
        if (L.exists) {
            float *row = &sh.u.prow[(L.g * 6 + L.s) * 30];
            int mode = 0;
            if (ar.hl_run && m.alive && mine) mode = oct_pilot_obs(c, tb, pub, L, m, row);
            else for (int k = 0; k < 30; k++) row[k] = 0.0f;
            if (pilot_mode) pilot_mode[u] = (uint8_t)(mode ? hl_selector(c, mode, m.ac_type, agent) : 0);
        }
        if (HH_RARE(active && pilot_mode && c.nA + c.nO < 6 && L.p == 3)) { /* the bytes of the slots without an aircraft */
            for (int sl = c.nA + c.nO; sl < 6; sl++) pilot_mode[(size_t)n * 6 + sl] = 0;
        }
        o_wave_sync();
        const int arenas = min(8, c.N - grp * 8);
        const int cnt = arenas * 6 * 30;
        float *dst = pilot_obs + (size_t)grp * 8 * 6 * 30;
        if (HH_USUAL((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (cnt & 3) == 0)) {
            const float4 *src4 = reinterpret_cast<const float4 *>(sh.u.prow);
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
            for (int k = tid; k < (cnt >> 2); k += 64) dst4[k] = src4[k];
        } else {
            for (int k = tid; k < cnt; k += 64) dst[k] = sh.u.prow[k];
        }
        if (P.pol_lut) hh_bin_rows_finish(bt, P.pol_lists, P.pol_max_rows, (int)u, pslot);
    }
    if (L.exists) {
        unit_store(P, U, u, m);
        P.acc_rew[u] = H.acc;
    }
    if (active && L.p == 0) {
        if (P.trace != nullptr && n < P.trace_K) P.trace_pos[n] = H.tcur;
        arena_store(P, n, ar);
        P.ep_ret[n] = H.ep_ret;
    }
    if (phase == HH_HL_AGENTS_ACT || phase == HH_HL_TICK || (VAR && phase == HH_HL_ACT_TICK)) {
        /* (HL_ACT_TICK: hl_run may have dropped in this very launch, so the arenas that ran the tick clear their mask) */
        if (active && L.p == 0 && ((phase == HH_HL_AGENTS_ACT && ar.hl_run) || (VAR && phase == HH_HL_ACT_TICK && ran_tick))) P.ev_mask[n] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the clear is acknowledged before this wave's atomics leave (one wave per arena group) */
        if (L.exists && H.evm) atomicOr(&P.ev_mask[n], H.evm);
        hh_act_fault_commit(P, n, L.exists, act_fault);
    }
}

/* one instance per phase (PHASE a compile-time constant: each holds its own path only — the form with the phase as a run-time argument carried every path's
 * registers: 337 at one wave per SIMD, 124 spilled at two) */
template <int W, int PHASE>
__global__ __launch_bounds__(64, W) void hh_k_hier_oct(DevPtrs P, DevCfg c, const int8_t *__restrict__ cmd, const int8_t *__restrict__ actions,
                                                     float *__restrict__ pilot_obs, uint8_t *__restrict__ pilot_mode, float *__restrict__ obs_out,
                                                     float *__restrict__ reward_out, uint8_t *__restrict__ valid_out, uint8_t *__restrict__ done_out,
                                                     int *__restrict__ running_count) {
    __shared__ OctShared sh;
    oct_phase_body<W, false, PHASE>(P, c, PHASE, (int)blockIdx.x, (int)threadIdx.x, sh, cmd, actions, pilot_obs, pilot_mode, obs_out, reward_out, valid_out, done_out,
                      running_count, running_count ? reinterpret_cast<unsigned long long *>(running_count + 2) : nullptr);
}

/* the variant-row form of the two phases that carry a sub-step (see oct_phase_body): HH_HL_BEGIN_V, then HH_HL_ACT_TICK per sub-step; HH_HL_END is the
 * standard kernel's.  pilot_obs [N, 15, 30], pilot_mode [N, 15], actions [N, 15, 4]. */
/* WPB independent waves per workgroup (no barrier between them: each has its own LDS slice and its own eight arenas).  One wave per workgroup lets the
 * dispatcher scatter a launch's waves over every CU of the chip, two or three to a CU — and a policy tile of the other sub-world's stream, which needs a
 * whole CU (eight waves x 256 registers, 141 KB of LDS), then finds none free until the launch is through (tools/timeline.py).  Eight waves x 16.6 KB fill a CU
 * the way a policy tile does: a launch of 512 waves touches 64 CUs and leaves the other 192 whole. */
#ifndef HHV_WPB
#define HHV_WPB 8
#endif
template <int W, int PHASE, int WPB>
__global__ __launch_bounds__(64 * WPB) void hh_k_hier_oct_v(DevPtrs P, DevCfg c, const int8_t *__restrict__ cmd, const int8_t *__restrict__ actions,
                                                       float *__restrict__ pilot_obs, uint8_t *__restrict__ pilot_mode, int *__restrict__ running_count) {
    /* the variant rows are staged where the envelope queue's exchange area lies (free once the tick is through: oct_phase_body waits before it writes them) */
    union VS {
        OctShared sh;
        struct { unsigned char head[offsetof(OctShared, u)]; alignas(16) float vrow[8 * HH_HL_VROWS * 30]; } v;
    };
    static_assert(offsetof(OctShared, u) % 16 == 0, "16-byte row stores");
    __shared__ VS vs_all[WPB];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = (int)blockIdx.x * WPB + wave;
    if (grp * 8 >= c.N) return;
    VS &vs = vs_all[wave];
    OctShared &sh = vs.sh;
    float *vrow = vs.v.vrow;
    HhTl tl;
    hh_tl_begin(tl);
    oct_phase_body<W, true, PHASE>(P, c, PHASE, grp, (int)threadIdx.x & 63, sh, cmd, actions, pilot_obs, pilot_mode, nullptr, nullptr, nullptr, nullptr,
                            running_count, running_count ? reinterpret_cast<unsigned long long *>(running_count + 2) : nullptr, vrow, &tl);
    hh_tl_end(tl, 1, (unsigned)c.arena_offset);
}

#endif /* HH_KERNELS_OCT_H */
