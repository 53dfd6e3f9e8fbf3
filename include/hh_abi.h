/*
 * hh_abi.h — C ABI of the MI355X batched air-combat world (libhh_world.so).
 *
 * The reference has no FFI: its boundary is RLlib's Python MultiAgentEnv protocol
 * (envs/env_hetero.py:20-63 LowLevelEnv, envs/env_hier.py:27-47 HighLevelEnv,
 * envs/env_base.py:62-109 reset/step).  This header is what a binding for that boundary
 * binds to; hhmarl_2d_amd/{env_hetero,env_hier}.py are the ctypes bindings that re-expose the
 * reference's dict protocol on top of it (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 on success or a negative HH_E_* code and never throws.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call that launches or copies makes the
 * world's device current for its duration and restores the caller's current device before it returns.  Buffers marked [dev] are
 * caller-owned device memory (e.g. torch tensors); [host] are host memory.  One host thread per
 * world.  The world owns only its struct-of-arrays state.
 *
 * Units are addressed 1..A in the reference (agents 1..n_agents, opponents after); arrays here
 * are 0-based slots in the same order.  A HighLevelEnv world has A = 6 unit slots, or A = 10 when a side has more than 3 aircraft
 * (HH_HL_SLOTS of hh_spec.h; up to 5 per side): with fewer aircraft than slots (n-vs-m evaluation scenarios) the slots behind the
 * last opponent are never alive and their rows read as zeros.
 */
#ifndef HH_ABI_H
#define HH_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HH_OK 0
#define HH_E_ARG (-1)      /* bad argument / unsupported configuration */
#define HH_E_HIP (-2)      /* HIP runtime error (hh_last_error() has the text) */
#define HH_E_NODEV (-3)    /* no usable GPU */

/* replaces the fields envs read from config.py's Namespace (config.py:17-54, 94-107) */
typedef struct hh_config {
    int32_t n_arenas;         /* arenas held by THIS world (this rank's shard) */
    int32_t env_kind;         /* HH_ENV_LOWLEVEL | HH_ENV_HIGHLEVEL */
    int32_t n_agents;         /* args.num_agents (2 low level; 1..5 high level) */
    int32_t n_opps;           /* args.num_opps   (2 low level; 1..5 high level: evaluation.py's n-vs-m scenarios) */
    int32_t level;            /* args.level 1..5 */
    int32_t agent_mode;       /* HH_MODE_FIGHT | HH_MODE_ESCAPE (args.agent_mode) */
    int32_t horizon;          /* args.horizon */
    int32_t friendly_kill;    /* args.friendly_kill */
    int32_t friendly_punish;  /* args.friendly_punish */
    int32_t esc_dist_rew;     /* args.esc_dist_rew */
    int32_t hier_action_assess;   /* args.hier_action_assess */
    int32_t hier_opp_fight_ratio; /* args.hier_opp_fight_ratio [%] */
    int32_t auto_reset;       /* 1: a finished arena is re-sampled inside hh_step */
    int32_t ext_opp_actions;  /* 1: opponents take actions from the caller (levels 4-5 frozen policies) */
    int32_t opp_side_selector; /* HighLevelEnv, evaluation.py with eval_hl = False: the opponents fly their OWN fight policies
                                  ("fight_{1,2}_opp" = L{eval_level_opp}, env_base.py:343-346,387-390): pilot_mode of an opponent's
                                  fight row carries HH_SEL_OPP_SIDE on top of policy type | aircraft type << 2 */
    int32_t reserved0;        /* keeps the doubles 8-byte aligned; must be 0 (hh_world_create refuses anything else: a caller built against the
                                 layout before opp_side_selector was added would otherwise be misread silently) */
    double map_size;          /* args.map_size */
    double glob_frac;         /* args.glob_frac */
    double rew_scale;         /* args.rew_scale */
    uint64_t seed;            /* keyed-RNG seed (hh_rng.h) */
    uint64_t arena_offset;    /* global id of local arena 0 (rank * n_arenas when sharded) */
} hh_config;

/* World snapshot used by hh_get_state / hh_set_state (parity tests, golden injection,
 * checkpointing).  All arrays are [host], arena-major: index = (arena * A + slot) * K + k. */
#define HH_ACF_K 6  /* lat, lon, hdg, spd, cmd_hdg, cmd_spd                       (a1, a6) */
#define HH_ACI_K 10 /* alive, ac_type, cannon_remain, cannon_burst, cannon_max,
                       missile_remain, rocket_max, missile_wait, has_missile, target      */
#define HH_RKF_K 4  /* lat, lon, hdg, cmd_hdg                                     (a10)   */
#define HH_RKI_K 4  /* alive, target, life, seq                                           */
#define HH_ARI_K 6  /* steps, alive_agents, alive_opps, escaping, escaping_time, episode  */
#define HH_TGT_K 3  /* stored sorted target list per unit (high level), ids / norm. distances; HighLevelEnv worlds with more than 3
                       aircraft on a side keep HH_TGT_K_WIDE entries (hh_spec.h: HH_TGT_K_OF) and their views are [N, A, 5] */
#define HH_TGT_K_WIDE 5
typedef struct hh_state_view {
    double *ac_f;   /* [N, A, HH_ACF_K] */
    int32_t *ac_i;  /* [N, A, HH_ACI_K] */
    double *rk_f;   /* [N, A, HH_RKF_K]  rocket slot s belongs to launcher slot s */
    int32_t *rk_i;  /* [N, A, HH_RKI_K] */
    int32_t *ar_i;  /* [N, HH_ARI_K] */
    int32_t *tgt_id; /* [N, A, K]  (0 = none), K = HH_TGT_K_OF(n_agents, n_opps) */
    double *tgt_d;   /* [N, A, K] */
} hh_state_view;

#define HH_SEL_OPP_SIDE 64 /* selector bit of an opponent's fight row when hh_config.opp_side_selector is set */

typedef struct hh_world hh_world;

/* ctor of LowLevelEnv / HighLevelEnv (env_hetero.py:20-51, env_hier.py:31-42) for N arenas */
int hh_world_create(const hh_config *cfg, int device, hh_world **out);
int hh_world_destroy(hh_world *w);
const char *hh_last_error(void);

/* observation width D (floats per controlled agent, zero padded): 26 fight / 30 escape / 34 commander */
int hh_obs_dim(const hh_world *w);
/* number of units whose actions the caller supplies per step: n_agents, or n_agents+n_opps
 * when ext_opp_actions */
int hh_n_ctrl(const hh_world *w);

/* reset() (env_base.py:62-77 + _reset_scenario 551-585 + state()): re-samples the arenas whose
 * mask byte is non-zero (mask [dev] u8[N]; NULL = all), episode += 1, and refreshes the
 * observation of those arenas into obs [dev] f32[N, n_agents, D] (may be NULL). */
int hh_reset(hh_world *w, const uint8_t *mask, float *obs, void *stream);

/* step() (env_base.py:79-109 -> env_hetero.py:105-186 _take_action -> cmano_simulator.py:138-157
 * do_tick -> env_hetero.py:188-225 rewards -> env_hetero.py:65-103 state).
 *   actions      [dev] i8 [N, n_ctrl, 4]   MultiDiscrete([13,9,2,2]); 4th ignored for type 2.  Components outside their ranges (the
 *                                          reference's spaces never emit them, env_hetero.py:37-43; its guards would raise, ac1.py:58-66)
 *                                          are SANITISED where the word is loaded — heading component clamped to [0, 12], speed component
 *                                          to [0, 8], fire components read as non-zero = fire (hh_spec.h: hh_action_sanitize) — and the
 *                                          arena's sticky fault flag is set (hh_action_faults).  Only CONSUMED words count: rows of dead
 *                                          units and of finished arenas may hold anything
 *   obs          [dev] f32[N, n_agents, D] observation after the tick (all agents, zeros if dead)
 *   reward       [dev] f32[N, n_agents]
 *   reward_valid [dev] u8 [N, n_agents]    1 iff the reference's rewards dict has the key
 *   done         [dev] u8 [N]              terminateds["__all__"]
 * Arenas already done (and not auto-reset) are left untouched and report reward_valid = 0. */
int hh_step(hh_world *w, const int8_t *actions, float *obs, float *reward, uint8_t *reward_valid,
            uint8_t *done, void *stream);

/* T consecutive steps from a pre-resident action tape in ONE launch sequence with no host
 * round trip: actions [dev] i8[T, N, n_ctrl, 4]; outputs are [T, ...] stacked like hh_step. */
int hh_rollout(hh_world *w, int32_t n_steps, const int8_t *actions, float *obs, float *reward,
               uint8_t *reward_valid, uint8_t *done, void *stream);

/* Levels 4-5 (ext_opp_actions): the opponents fly frozen policies that the reference evaluates INSIDE step(),
 * in unit id order after the agents acted (env_hetero.py:160-172, env_base.py:349-398).  Faithful split:
 *   hh_step_begin(agent_actions [dev] i8[N, n_agents, 4], opp_mode 0 fight / 1 escape)
 *       -> opp_obs [dev] f32[N, n_opps, 30]: lowlevel_state(opp_mode, i) of each live opponent (zeros if dead)
 *   (caller runs its frozen policies)
 *   hh_step_finish(opp_actions [dev] i8[N, n_opps, 4]) -> outputs exactly like hh_step.
 * hh_step with n_ctrl = n_agents + n_opps remains for callers that do not need the opponents' observations.
 * opp_mode = HH_OPP_MODE_EPISODE: level 5 in fight mode draws, per arena and per episode, which frozen policy set the
 * opponents fly (k = randint(3,5), env_hetero.py:55-59) and k == 5 means "escape": each arena then observes in the mode of
 * its own draw; hh_opp_policy reports k so that the caller evaluates policies[k]. */
#define HH_OPP_MODE_EPISODE (-1)
int hh_step_begin(hh_world *w, const int8_t *agent_actions, int32_t opp_mode, float *opp_obs, void *stream);
int hh_step_finish(hh_world *w, const int8_t *opp_actions, float *obs, float *reward, uint8_t *reward_valid,
                   uint8_t *done, void *stream);

/* level 5 / fight mode: k in {3,4,5} of every arena's CURRENT episode -> [dev] i8[N] (0 for every other configuration) */
int hh_opp_policy(hh_world *w, int8_t *k_out, void *stream);

/* current observation of every arena without stepping (state(): env_hetero.py:62-63 / env_hier.py:49-98) */
int hh_observe(hh_world *w, float *obs, void *stream);

/* HighLevelEnv.step (envs/env_hier.py:114-140) for 3-vs-3 worlds.  One commander step =
 *   hh_hl_begin(commander actions)            _action_assess + opponents' draws (142-190)
 *   up to 16 x { pilots(agents) -> hh_hl_agents_act -> pilots(opponents) -> hh_hl_tick }
 *   hh_hl_end                                 done, rewards, commander observation (49-98)
 * The frozen pilot policies (env_base.py:349-398) run in the caller between the launches on the
 * observations these calls emit: pilot_obs [dev] f32 [N, 6, 30] = lowlevel_state (env_hier.py:100-112)
 * of the side that acts next (agents after begin/tick, opponents after agents_act), zero elsewhere;
 * pilot_mode [dev] u8 [N, 6]: 0 = no action needed, else (policy type: 1 fight / 2 escape) | (aircraft type 1 / 2) << 2, i.e.
 * 5 = Fight1, 6 = Esc1, 9 = Fight2, 10 = Esc2 — the selector byte hh_policy_act (hh_policy.h) maps to a network.
 * actions [dev] i8 [N, 6, 4] (rows of the side that acts).  commander_actions [dev] i8 [N, 3] in {0,1,2}.
 * `running` (host, nullable): number of arenas still inside their macro step after this tick. */
int hh_hl_begin(hh_world *w, const int8_t *commander_actions, float *pilot_obs, uint8_t *pilot_mode, void *stream);
int hh_hl_agents_act(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, void *stream);
int hh_hl_tick(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, int32_t *running, void *stream);
int hh_hl_end(hh_world *w, float *obs, float *reward, uint8_t *reward_valid, uint8_t *done, void *stream);

/* The same commander step with ONE launch and ONE policy call per sub-step (worlds of up to 3 aircraft per side):
 *   hh_hl_begin_variants(commander actions) -> up to 16 x { pilots(all rows) -> hh_hl_act_tick } -> hh_hl_end
 * The standard path needs a launch per SIDE because the opponents' pilots observe the agents' weapon flags of the same sub-step (env_base.py:208-211,
 * units act in id order).  An agent's action can only RAISE its flag, and an opponent's row carries the flag of at most two agents, so these calls emit
 * each opponent's row together with its VARIANTS — the copies with the observed agents' still-zero flags forced to one — and hh_hl_act_tick, after
 * letting the agents act, takes each opponent's action from the variant that matches what happened.  Same rows through the same networks: the same
 * trajectories as the standard path, bit for bit — WHEN both run the same forward form of the policy kernel (a row's logits do not depend on its
 * tile, but the tile forms and the weights-through-LDS forms differ in the last bits, and the two paths' call sizes can fall on different sides of
 * the form threshold: a near-tie arg-max could then differ; pin the form with HH_POLICY_W / hh_policy_set_tile_rows for a bit-for-bit A/B).  Row slots per arena: HH_HL_VROWS = 15 = agents 0..2, then opponent j's variant v at 3 + 4 j + v
 * (v bit 0 / bit 1 = the first / second observed agent's flag forced; v = 0 is the row as the world stands).
 * pilot_obs [dev] f32 [N, 15, 30], pilot_mode [dev] u8 [N, 15] (0 = slot not in use), actions [dev] i8 [N, 15, 4] (the policy's output for every listed row).
 * A bound policy bank (hh_bind_policy) needs max_rows >= 15 * N. */
#define HH_HL_VROWS 15
int hh_hl_begin_variants(hh_world *w, const int8_t *commander_actions, float *pilot_obs, uint8_t *pilot_mode, void *stream);
int hh_hl_act_tick(hh_world *w, const int8_t *actions, float *pilot_obs, uint8_t *pilot_mode, int32_t *running, void *stream);

/* HighLevelEnv.step in ONE launch for callers whose pilot actions exist before the step starts (a recorded / scripted tape,
 * env-only throughput runs): pilot_tape [dev] i8 [16, N, 6, 4] = the actions of sub-step k in slice k (each side's rows are
 * read at its turn).  Same result as hh_hl_begin + 16 x {hh_hl_agents_act, hh_hl_tick} + hh_hl_end with those actions, bit
 * for bit; state stays in registers across the sub-steps and a workgroup stops as soon as none of its arenas is still inside
 * its macro step.  Pilot observations are not emitted (nobody reads them when the actions are already known). */
int hh_hl_rollout(hh_world *w, const int8_t *commander_actions, const int8_t *pilot_tape, float *obs, float *reward,
                  uint8_t *reward_valid, uint8_t *done, void *stream);

/* commander_actions of the current macro step after _action_assess (env_hier.py:142-190): agents' validated
 * actions and the opponents' drawn ones, [host] i8 [N, 6]; read by the eval_info counters (env_base.py:91-107) */
int hh_hl_commands(hh_world *w, int8_t *out);

/* cumulative arena-ticks run by HighLevelEnv macro steps on this world ([host] u64; synchronises `stream`): arenas leave
 * a macro step early (kill / surrounding event, env_hier.py:125), so ticks/s must be counted, not assumed 16 per step */
int hh_hl_tick_count(hh_world *w, uint64_t *out, void *stream);

/* name of the kernel instance hh_rollout / hh_step (or the hh_hl_* phases) launch for this world on this device */
int hh_rollout_kernel_name(hh_world *w, char *buf, int32_t len);

/* the same instance as a profiler prints it (demangled template arguments, e.g. "hh_k_world_quad<1, 1, true, 8>"): bench.py only
 * quotes counter evidence (profiles/latest_*.json) whose kernel name contains this string.  which = 0: the kernel of hh_rollout /
 * hh_step (LowLevelEnv) or of the hh_hl_* phase launches (HighLevelEnv); 1: the kernel of hh_hl_rollout */
int hh_kernel_instance(hh_world *w, int32_t which, char *buf, int32_t len);

/* Evaluation counters of HighLevelEnv.step (envs/env_base.py:91-107, read by evaluation.py:66-82), computed on the device
 * for every arena inside hh_hl_end: columns = agents_win, opps_win, draw (flags of the step that ended the episode),
 * agent_fight, agent_escape, opp_fight, opp_escape, agent_steps, opp_steps, opp1, opp2, opp3 (units that still exist after
 * the step, by their assessed commander action; oppK = agents fighting their K-th stored target).
 *   last  [dev] i32 [N, HH_EVAL_K]  nullable: the info dict of the most recent commander step of every arena
 *   total [dev] i32 [N, HH_EVAL_K]  nullable: summed over every commander step since the world was created / last cleared
 * clear_total != 0 zeroes the sums after copying them. */
#define HH_EVAL_K 12
int hh_eval_info(hh_world *w, int32_t *last, int32_t *total, int32_t clear_total, void *stream);

/* steps, alive_agents, alive_opps, done of every arena -> [dev] i32 [N, 4] (what step({}) needs: env_base.py:87-90) */
int hh_arena_status(hh_world *w, int32_t *out, void *stream);

/* Synthetic action tape of the benchmark workloads (BASELINE configs[1] "random actions": i.i.d. uniform over
 * MultiDiscrete([13,9,2,2]) from the keyed RNG, key = (seed, global arena, step, agent)): fills out [dev] i8 [T, N, n_units, 4]: the word at (t, n, s) is
 * hh_rng.h's `hh_rng_action_word` of seed, global arena arena_offset + n, step step0 + t, unit s + 1.  No world needed: the tape of a shard is the
 * slice of the global tape its arena_offset selects.  n_units = 2 for LowLevelEnv agents, 6 for a HighLevelEnv pilot tape. */
int hh_action_tape_uniform(uint64_t seed, uint64_t arena_offset, int32_t step0, int32_t T, int32_t N, int32_t n_units, int8_t *out, void *stream);

/* The device-side replacement of the reference's raising guards (ac1.py:58-66 set_heading / set_speed; SURVEY.md section 5 "invalid-state
 * flag per arena instead of raising"): out [dev] u8 [N] (nullable) = 1 for every arena in which, since the flags were last cleared, a step
 * (hh_step / hh_rollout / hh_step_begin / hh_step_finish / hh_hl_agents_act / hh_hl_tick / hh_hl_act_tick / hh_hl_rollout) consumed an action word with a
 * component outside MultiDiscrete([13,9,2,2]); that step ran on the sanitised word (see hh_step).  clear != 0 zeroes the flags after
 * copying them.  Resets do not clear them.  Ordered on `stream`, no host synchronisation. */
int hh_action_faults(hh_world *w, uint8_t *out, int32_t clear, void *stream);

/* Trajectory ring buffer on the device (rendering / trace export; cmano_simulator.py:125-130,159-162 record_unit_trace,
 * env_base.py:587-645 plot): after hh_trace_enable the first n_arenas arenas append one row per unit after every reset and every
 * tick, HH_TRACE_F floats = lat, lon, heading, speed, alive, rocket lat, rocket lon, rocket alive + 16 * episode; the ring keeps
 * the last `capacity` rows.  hh_trace_read copies it out: rows [host] f32 [capacity, n_arenas, A, HH_TRACE_F] (slot = row index
 * % capacity), count [host] i32 [n_arenas] = rows written so far.  While tracing is on, 2-vs-2 rollouts run on the generic kernel. */
#define HH_TRACE_F 8
int hh_trace_enable(hh_world *w, int32_t n_arenas, int32_t capacity);
int hh_trace_read(hh_world *w, float *rows, int32_t *count);

/* per-arena statistics of the most recently FINISHED episode (logging; this is what the
 * multi-GPU all-gather moves): ret [dev] f32[N] (sum of agent rewards), len [dev] i32[N],
 * outcome [dev] i8[N] (1 agents win, -1 opponents win, 0 draw, 2 none finished yet) */
int hh_episode_stats(hh_world *w, float *ret, int32_t *len, int8_t *outcome, void *stream);
/* the same three statistics as one [dev] f32[N, 3] block (return, length, outcome as floats), written by one small
 * launch: the block each rank contributes to the logging all-gather (hhmarl_2d_amd/sharding.py) */
int hh_episode_stats_packed(hh_world *w, float *out, void *stream);

/* Generalised advantage estimation on the stacked outputs of hh_rollout (what RLlib's postprocessing does
 * for the reference: train_hetero.py:216 gamma=0.99, lambda_=0.95, complete episodes).  All [dev]:
 * reward, valid, adv, ret [T, N, n_agents]; value [T+1, N, n_agents] (critic incl. bootstrap row); done [T, N].
 * Not tied to a world: launches on the calling thread's current device, which must own the buffers. */
int hh_gae(int32_t T, int32_t N, int32_t n_agents, const float *reward, const float *value, const uint8_t *valid,
           const uint8_t *done, float gamma, float lam, float *adv, float *ret, void *stream);

/* The same scan with the semantics RLlib 2.4 gives the reference's step stream (what train_hetero.py / train_hier.py actually
 * train on): no reward key = reward 0.0 and the row stays in the trajectory (the reference returns observations for dead agents,
 * env_hetero.py:65-103,217-223, and RLlib's sampler reads rewards[env_id].get(agent_id, 0.0)); last_r = 0.0 after every episode
 * end; delta and the discounted sum in float64 (np.concatenate with last_r promotes, scipy.signal.lfilter inside discount_cumsum
 * (ray/rllib/evaluation/postprocessing.py).  `reward` must hold 0.0 where hh_rollout reported reward_valid = 0 (it does).
 * train_hetero.py:216: gamma 0.99, lambda 0.95; train_hier.py:186: gamma 0.99 and RLlib's default lambda 1.0. */
int hh_gae_rllib(int32_t T, int32_t N, int32_t n_agents, const float *reward, const float *value, const uint8_t *done, double gamma,
                 double lam, float *adv, float *ret, void *stream);

/* Test probe: the shared math of include/hh_math.h / hh_geodesic.h evaluated ON THE DEVICE over arrays of operands, so that a
 * -m gpu test can compare the kernels' arithmetic with the CPU oracle's bit for bit (tests/test_gpu_math.py), not only through
 * trajectories.  fn: 0 sincos(a) -> o0, o1 | 1 atan2(a, b) | 2 acos(a) | 3 sincosd(a) -> o0, o1 | 4 atan2d(a, b) | 5 pymod(a, b) |
 * 6 remainder(a, b) | 7 fmod(a, b) | 8 round3(a) | 9 a / b by hh_div_known | 10 pymod_turn(a, b) | 11 sqrt(a) | 12 clip(a, 0, 1) |
 * 13 clip(a, -b, b) | 14 geo_move(lat = a, lon = b, azi = o0_in, s = o1_in) -> o0, o1 (the outputs carry the third and fourth
 * operand in).  a, b, o0, o1: device arrays of n doubles. */
int hh_math_eval(int32_t fn, int32_t n, const double *a, const double *b, double *o0, double *o1, void *stream);

/* host snapshot in / out (synchronises the stream) */
int hh_get_state(hh_world *w, hh_state_view *view);
int hh_set_state(hh_world *w, const hh_state_view *view);

/* integer event masks of the last step, for bit-exact parity checks ([host]; ordered after the work queued on
 * `stream` and synchronises that stream only):
 * per arena u32: bits 0..7 units killed by cannon, 8..15 killed by rocket, 16..23 out of bounds,
 * 24..31 missile launched this step (by unit slot); ten-slot arenas (more than 3 aircraft on a side): hh_spec.h HH_EV_BIT */
int hh_get_event_masks(hh_world *w, uint32_t *masks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HH_ABI_H */
