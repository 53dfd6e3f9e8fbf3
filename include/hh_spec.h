/*
 * hh_spec.h — every numeric constant of the air-combat world in one place, shared by the HIP
 * kernels and the C oracle (the Python side holds no copy of them: the spaces / observation widths it
 * needs are in hhmarl_2d_amd/env_hetero.py and env_hier.py).  Each block cites the reference lines the numbers come from.
 */
#ifndef HH_SPEC_H
#define HH_SPEC_H

#include <stdint.h>

/* warsim/simulator/cmano_simulator.py:21 */
#define HH_KNOTS_TO_MS 0.514444

/* envs/env_base.py:43 — MapLimits(7.0, 5.0, 7.0+map_size, 5.0+map_size) */
#define HH_MAP_LON0 7.0
#define HH_MAP_LAT0 5.0

/* Aircraft type tables, index = ac_type-1.
 * type 1 "Rafale"     warsim/simulator/ac1.py:24-36
 * type 2 "RafaleLong" warsim/simulator/ac2.py:23-32 */
#define HH_AC_TURN_RATE(t)   ((t) == 1 ? 5.0 : 3.5)    /* max_deg_sec */
#define HH_AC_MAX_SPEED(t)   ((t) == 1 ? 900.0 : 600.0) /* max_speed_knots */
#define HH_AC_INV_MAX_SPEED(t) ((t) == 1 ? (1.0 / 900.0) : (1.0 / 600.0))
#define HH_AC_ACCEL(t)       ((t) == 1 ? 35.0 : 28.0)   /* max_knots_sec */
#define HH_AC_CANNON_KM(t)   ((t) == 1 ? 2.0 : 4.5)     /* cannon_range_km */
#define HH_AC_CANNON_HALF(t) ((t) == 1 ? (10.0 / 2.0) : (7.0 / 2.0)) /* cannon_width_deg / 2.0 */
#define HH_AC_BURST(t)       ((t) == 1 ? 5 : 3)         /* cannon_burst_time_sec */
/* per-tick hit probability = cannon_hit_prob / (cannon_burst_time_sec / tick_secs), ac1.py:112-113, ac2.py:99-100 */
#define HH_AC_HIT_PROB(t)    ((t) == 1 ? (0.75 / (5.0 / 1.0)) : (0.9 / (3.0 / 1.0)))
#define HH_AC_CANNON_DEFAULT 200 /* cannon_max_time_sec (both) */
#define HH_AC1_MISSILES_DEFAULT 5 /* ac1.py:33 */
#define HH_MISSILE_RANGE_KM 111.0 /* ac1.py:34 */
#define HH_MISSILE_HALF_DEG 60.0  /* ac1.py:35,145-146: missile_width_deg / 2 */

/* Rocket, warsim/simulator/rocket_unit.py:14-21: quadratic spline through (0,500),(10,2000),
 * (20,1400),(30,600) kn, tabulated for life = 0..10 s with SciPy 1.15.3 (the closed form
 * 500 + 3250/12 t - 145/12 t^2 agrees to 2e-12 kn; the table keeps SciPy's last digits). */
#define HH_ROCKET_TURN_RATE 10.0
#define HH_ROCKET_MAX_LIFE 10 /* removed when life > speed_profile_time[1] */
#define HH_ROCKET_FUSE_KM 1.0
#define HH_ROCKET_SPEED_TABLE                                                                    \
    { 500.0, 758.7499999999999, 993.3333333333335, 1203.75, 1390.0, 1552.083333333333,           \
      1690.0000000000002, 1803.75, 1893.3333333333335, 1958.75, 2000.0 }

/* Observation sizes, envs/env_base.py:29-32, envs/env_hier.py:20-25 */
#define HH_OBS_FIGHT_AC1 26
#define HH_OBS_FIGHT_AC2 24
#define HH_OBS_ESC_AC1 30
#define HH_OBS_ESC_AC2 29
#define HH_OBS_HL 34
#define HH_N_OPP_HL 2

/* environment kind / agent mode */
#define HH_ENV_LOWLEVEL 0  /* envs/env_hetero.py LowLevelEnv */
#define HH_ENV_HIGHLEVEL 1 /* envs/env_hier.py HighLevelEnv */
#define HH_MODE_FIGHT 0
#define HH_MODE_ESCAPE 1

#define HH_MAX_AIRCRAFT 10 /* unit slots of an arena: 2v2 -> 4, up to 3 per side -> 6, up to 5 per side (evaluation.py's n-vs-m) -> 10 */
#define HH_SIDE_MAX 5      /* aircraft per side */
/* unit slots of a HighLevelEnv arena: agents in slots 0 .. n_agents-1, opponents behind them, the rest never alive */
#define HH_HL_SLOTS(n_agents, n_opps) (((n_agents) > 3 || (n_opps) > 3) ? 10 : 6)
/* entries of a unit's stored sorted target list (env_hier.py:52,97: an opponent keeps every live agent) */
#define HH_TGT_K_OF(n_agents, n_opps) (((n_agents) > 3 || (n_opps) > 3) ? 5 : 3)

/* event-mask bit of a unit slot (hh_get_event_masks): classes 0 killed by cannon, 1 killed by rocket, 2 out of bounds, 3 missile
 * launched.  Arenas of up to 8 slots: class c in bits 8c .. 8c+7.  Ten-slot arenas: classes 0..2 in bits 10c .. 10c+9, launches as
 * one bit per SIDE (30: an agent launched, 31: an opponent). */
#define HH_EV_BIT(slots, cls, slot, is_agent) \
    ((slots) <= 8 ? (1u << (8 * (cls) + (slot))) : ((cls) < 3 ? (1u << (10 * (cls) + (slot))) : (1u << ((is_agent) ? 30 : 31))))

/* Action words: MultiDiscrete([13, 9, 2, 2]) = relative heading, speed level, fire cannon, fire missile (envs/env_hetero.py:37-43,
 * env_base.py:214-238).  The reference's spaces never emit anything else, and its guards would raise on most of what lies outside
 * (ac1.py:58-66: set_speed refuses a speed outside [100, max]; set_heading is unreachable behind `% 360`).  A batched world cannot
 * raise for one arena, so an untrusted action word is SANITISED where it is loaded — heading component clamped to [0, 12], speed
 * component to [0, 8], the two fire components read as "non-zero = fire" (the reference's bool()) — and the caller is told through a
 * sticky per-arena flag (hh_action_faults, hh_abi.h) that the step ran on the sanitised action.  The kernels and the oracle share this
 * one definition.  w = the four int8 components packed little-endian (a0 in bits 0..7).  Returns the sanitised word; *bad is OR-ed
 * with 1 when any component was outside its range. */
#define HH_ACT_HEADING_MAX 12
#define HH_ACT_SPEED_MAX 8
#if defined(__HIPCC__)
#define HH_SPEC_FN __host__ __device__ __forceinline__
#else
#define HH_SPEC_FN static inline
#endif
HH_SPEC_FN uint32_t hh_action_sanitize(uint32_t w, int *bad) {
    const int a0 = (int)(int8_t)(w & 0xffu), a1 = (int)(int8_t)((w >> 8) & 0xffu);
    const int c0 = a0 < 0 ? 0 : (a0 > HH_ACT_HEADING_MAX ? HH_ACT_HEADING_MAX : a0);
    const int c1 = a1 < 0 ? 0 : (a1 > HH_ACT_SPEED_MAX ? HH_ACT_SPEED_MAX : a1);
    const uint32_t f2 = (w & 0x00ff0000u) != 0u, f3 = (w & 0xff000000u) != 0u;
    *bad |= (c0 != a0) | (c1 != a1) | ((w & 0xfefe0000u) != 0u);
    return (uint32_t)c0 | ((uint32_t)c1 << 8) | (f2 << 16) | (f3 << 24);
}

/* Keyed-RNG draw sites.  One id per reference call site family. */
enum {
    HH_SITE_RESET_SIDE = 1,   /* envs/env_base.py:555  randint(1,2)                       */
    HH_SITE_RESET_X = 2,      /* env_base.py:496-547 / env_hier.py:232-246  uniform (lon) */
    HH_SITE_RESET_Y = 3,      /*   "    uniform (lat)                                     */
    HH_SITE_RESET_HDG = 4,    /*   "    randint (heading)                                 */
    HH_SITE_RESET_TYPE = 5,   /* env_base.py:560  randint(1,2) for slots >= 2             */
    HH_SITE_RESET_L5K = 6,    /* env_hetero.py:57 randint(3,5)                            */
    HH_SITE_MISSILE_WAIT = 7, /* env_base.py:230  randint(7,17) LL / (8,12) HL            */
    HH_SITE_L12_COIN = 8,     /* env_hetero.py:119,132  randint(0,1)                      */
    HH_SITE_L2_PERIOD = 9,    /* env_hetero.py:127  randint(35,45)                        */
    HH_SITE_L2_TURN = 10,     /* env_hetero.py:128  randint(0,1)                          */
    HH_SITE_L2_SPEED = 11,    /* env_hetero.py:130  randint(0,4)                          */
    HH_SITE_L3_ESC_COIN = 12, /* env_hetero.py:140  randint(0,1)                          */
    HH_SITE_L3_ESC_TIME = 13, /* env_hetero.py:142  uniform(20,30)                        */
    HH_SITE_ESC_HDG = 14,     /* env_hetero.py:235-242  uniform                           */
    HH_SITE_ESC_SPEED = 15,   /* env_hetero.py:243  uniform(300,600)                      */
    HH_SITE_ESC_FIRE = 16,    /* env_hetero.py:245  randint(0,1)                          */
    HH_SITE_HC_SPEED1 = 17,   /* env_hetero.py:255  uniform(100,400)                      */
    HH_SITE_HC_R = 18,        /* env_hetero.py:259  uniform(0.7,1.3)                      */
    HH_SITE_HC_SPEED2 = 19,   /* env_hetero.py:265  uniform(500,800) | uniform(100,500)   */
    HH_SITE_ROCKET_NOISE = 20,/* warsim/simulator/ac1.py:127  uniform(0.95,1.05)          */
    HH_SITE_CANNON = 21,      /* ac1.py:112, ac2.py:99  rnd_gen.random(); sub = target id */
    HH_SITE_HL_FIGHT = 22,    /* env_hier.py:176  choices([0,1], weights)                 */
    HH_SITE_HL_OTHER = 23,    /* env_hier.py:179  choices([0,1], [1,3])                   */
    HH_SITE_HL_PICK = 24,     /* env_hier.py:181  randint(2, k)                           */
    HH_SITE_POLICY_SAMPLE = 25, /* hh_policy_sample (hh_policy.h): the Categorical draw RLlib's sampler makes per action component
                                   (TorchMultiCategorical.sample -> torch.multinomial, unseeded in the reference); sub = component */
    HH_SITE_ACTION_TAPE = 26,   /* hh_action_tape_uniform (hh_abi.h): the synthetic "random actions" of the benchmark workloads (SURVEY.md 8d:
                                   i.i.d. uniform over MultiDiscrete([13,9,2,2]) keyed by (arena, step, agent)); sub = component; episode key 0 */
    HH_SITE_COUNT = 27
};

#endif /* HH_SPEC_H */
