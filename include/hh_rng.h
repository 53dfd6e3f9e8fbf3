/*
 * hh_rng.h — counter-based keyed random numbers.
 *
 * The reference is unseeded and its draw ORDER is data dependent (global `random` at ~40 sites,
 * envs/env_base.py:62-77 ignores `seed`, warsim/simulator/cmano_simulator.py:88 seeds with None),
 * so "identical seeds" parity is defined on a keyed tape instead of a stream:
 *
 *     u = U(seed, arena, episode, tick, unit, site, sub)  in [0, 1)
 *
 * - arena   : global arena id (rank offset + local index) -> sharding does not change results
 * - episode : number of resets of this arena (1 for the first episode)
 * - tick    : value of the env's `steps` counter when the reference would draw
 * - unit    : 1-based unit id the draw belongs to (0 for arena-level draws)
 * - site    : HH_SITE_* (hh_spec.h), sub: extra index (cannon target id)
 *
 * uniform(a,b) = a + (b-a)*u            (CPython random.uniform)
 * randint(a,b) = a + floor(u*(b-a+1))   (harness-defined; CPython's own randint is rejection
 *                                        sampling on getrandbits and is patched at function level)
 * The same functions are patched into the imported reference by oracle/ref_harness.py when the
 * golden traces are generated, so the reference, the C oracle and the HIP kernels all see the
 * same numbers.  Mixer: the 64-bit finalizer of splitmix64 (Steele, Lea, Flood 2014).
 */
#ifndef HH_RNG_H
#define HH_RNG_H

#include <stdint.h>
#include "hh_math.h"
#include "hh_spec.h"

HH_HD uint64_t hh_mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

/* per-arena stream key (depends on seed and global arena id) */
HH_HD uint64_t hh_rng_arena_key(uint64_t seed, uint64_t arena) {
    return hh_mix64(seed ^ (0x9e3779b97f4a7c15ULL * (arena + 1ULL)));
}

/* per-(episode, tick) key */
HH_HD uint64_t hh_rng_tick_key(uint64_t arena_key, uint32_t episode, uint32_t tick) {
    return hh_mix64(arena_key ^ (((uint64_t)episode << 32) | (uint64_t)tick));
}

HH_HD double hh_rng_u01(uint64_t tick_key, uint32_t unit, uint32_t site, uint32_t sub) {
    uint64_t h = hh_mix64(tick_key + (((uint64_t)unit << 32) | ((uint64_t)site << 16) | (uint64_t)sub));
    return (double)(h >> 11) * 0x1.0p-53;
}

HH_HD double hh_rng_uniform(double u, double a, double b) { return a + (b - a) * u; }
HH_HD int hh_rng_randint(double u, int a, int b) { return a + (int)hh_floor(u * (double)(b - a + 1)); }

/* envs/env_hetero.py:55-59: at level 5 (fight mode) every reset() draws k = randint(3,5): the opponents fly
 * self.policies[k] for that episode and observe in escape mode when k == 5.  The draw is keyed by (arena, episode,
 * tick 0, unit 0), so nothing needs to be stored: whoever knows the arena's episode counter recomputes it. */
HH_HD int hh_l5_policy_pick(uint64_t arena_key, uint32_t episode) {
    return hh_rng_randint(hh_rng_u01(hh_rng_tick_key(arena_key, episode, 0u), 0u, HH_SITE_RESET_L5K, 0u), 3, 5);
}

/* One word of the synthetic action tape (BASELINE configs[1]: "random actions"): the four components of agent `unit`
 * (1-based) of global arena `arena` at step index `step`, i.i.d. uniform over MultiDiscrete([13, 9, 2, 2]), keyed — the same word on any
 * device, shard or host.  Returns the packed little-endian int8 word (a0 in bits 0..7). */
HH_HD uint32_t hh_rng_action_word(uint64_t seed, uint64_t arena, uint32_t step, uint32_t unit) {
    const uint64_t tk = hh_rng_tick_key(hh_rng_arena_key(seed, arena), 0u, step);
    const uint32_t a0 = (uint32_t)hh_rng_randint(hh_rng_u01(tk, unit, HH_SITE_ACTION_TAPE, 0u), 0, 12);
    const uint32_t a1 = (uint32_t)hh_rng_randint(hh_rng_u01(tk, unit, HH_SITE_ACTION_TAPE, 1u), 0, 8);
    const uint32_t a2 = (uint32_t)hh_rng_randint(hh_rng_u01(tk, unit, HH_SITE_ACTION_TAPE, 2u), 0, 1);
    const uint32_t a3 = (uint32_t)hh_rng_randint(hh_rng_u01(tk, unit, HH_SITE_ACTION_TAPE, 3u), 0, 1);
    return a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
}

#endif /* HH_RNG_H */
