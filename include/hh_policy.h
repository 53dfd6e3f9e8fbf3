/*
 * hh_policy.h — C ABI of the frozen pilot / opponent policy networks (part of libhh_world.so).
 *
 * The reference evaluates frozen PyTorch policies INSIDE its environments: envs/env_base.py:312-347 `_get_policies`
 * (torch.load of policies/L*_AC*_{fight,escape}.pt) and 349-398 `_policy_actions` (one forward per live unit per tick with
 * dummy centralised-critic inputs, batch of one, seq_lens [1], arg-max per action component).  This header is what a
 * binding for that part binds to: the ACTOR half of the four architectures of models/ac_models_hetero.py
 * (Esc1 29-103, Esc2 105-180, Fight1 181-291, Fight2 293-404) as one fused gfx950 kernel over all units of all
 * arenas at once, writing int8 actions straight into the buffer hh_step / hh_step_finish / hh_hl_agents_act / hh_hl_tick read.
 *
 * Conventions as in hh_abi.h: 0 on success or a negative HH_E_* code, never throws; [dev] = caller-owned device memory,
 * [host] = host memory; `stream` is a hipStream_t passed as void*.
 */
#ifndef HH_POLICY_H
#define HH_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HH_NET_FIGHT1 0 /* models/ac_models_hetero.py:181-291, obs 26 -> 26 logits */
#define HH_NET_FIGHT2 1 /* 293-404, obs 24 -> 24 logits */
#define HH_NET_ESC1 2   /* 29-103,  obs 30 -> 26 logits */
#define HH_NET_ESC2 3   /* 105-180, obs 29 -> 24 logits */
#define HH_POLICY_MAX_NETS 8
#define HH_POLICY_LOGITS 32 /* row width of the optional logits output (zero padded) */

/* One network, as the reference's state_dict() holds it ([host], row-major [out, in] like nn.Linear.weight):
 * inp_w/inp_b[k] = inp{k+1}._model.0.{weight,bias}; att_* = att_act.{in_proj_weight [300,100], in_proj_bias [300],
 * out_proj.weight [100,100], out_proj.bias [100]} (fight nets only, NULL otherwise); shared_* = shared_layer._model.0
 * [500,500]; out_* = act_out._model.0 [26|24, 500].  The value branch is not used for acting. */
typedef struct hh_net_weights {
    int32_t kind;
    const float *inp_w[3];
    const float *inp_b[3];
    const float *att_in_proj_w, *att_in_proj_b, *att_out_w, *att_out_b;
    const float *shared_w, *shared_b;
    const float *out_w, *out_b;
} hh_net_weights;

typedef struct hh_policy hh_policy;

/* a bank of up to HH_POLICY_MAX_NETS networks on `device`; max_rows = the largest n_rows hh_policy_act will be given */
int hh_policy_create(int device, int32_t max_rows, hh_policy **out);
int hh_policy_destroy(hh_policy *p);

/* load (or replace) network `slot`: weights are repacked on the host into the kernel's k-interleaved layout (the two
 * attention projections folded into one 100x100 matrix in double precision: with a sequence of length 1 the softmax is 1
 * and the attention output is out_proj(v_proj(x))) and copied to the device.  Synchronous. */
int hh_policy_set_net(hh_policy *p, int32_t slot, const hh_net_weights *w);

/* selector byte -> network: lut[256] [host], value 0 = "no action for this row" (its action is written as zeros), s + 1 =
 * network slot s.  The world kernels emit selector bytes as pilot_mode (hh_hl_*: policy type | ac_type << 2); callers of the
 * LowLevelEnv split step build them from the unit's slot and the arena's level-5 draw. */
int hh_policy_set_lut(hh_policy *p, const uint8_t *lut);
/* Rows per workgroup tile of the forward kernel: 0 = chosen per call from the row count (the default: 64-row tiles when they come in whole
 * rounds of one per CU, 32-row tiles otherwise), 32 or 64 = that instance always.  Same results either way (bit-identical logits); a caller
 * that keeps several banks busy on concurrent streams (bench.py --workload hier --pilot net --streams K) prefers 64: the other streams'
 * kernels fill the CUs a partial round leaves idle.  The environment variable HH_POLICY_TILE, read at hh_policy_create, sets the same.
 * Calls with more than 40 x (number of CUs) rows that carry a network (10240 on an MI355X) run another form when the width is 0: the weights
 * streamed through LDS once per 64 rows, the activations resident in registers (hh_policy_kernel_w16.h: hh_k_policy_w16<4>; when 128-row tiles come in
 * whole rounds of one per CU its eight-wave instance hh_k_policy_w16<8>: one pass over the weights per 128 rows.  HH_POLICY_W=0 keeps the tile forms,
 * 2 / 3 force the <4> / <8> instance).  Its logits differ from the tile forms' in the last bits (the
 * output layer is summed in one k-ordered accumulator): same 1e-5 bound; <4> and <8> compute every row with the same operation order. */
int hh_policy_set_tile_rows(hh_policy *p, int32_t rows);

/* name of the forward kernel instance a call of n_rows rows launches on this bank, as a profiler prints it ("hh_k_policy_h<1>",
 * "hh_k_policy_h<2>", "hh_k_policy_w16<4>", "hh_k_policy_w16<8>"; sampler != 0: hh_policy_sample's
 * "hh_k_policy_ppo" / "hh_k_policy_w16_ppo"): bench.py quotes counter
 * evidence only for the instance it actually ran */
int hh_policy_kernel_name(hh_policy *p, int32_t n_rows, int32_t sampler, char *buf, int32_t len);

/* greedy actions of n_rows units in one launch sequence (row binning by network + the fused forward):
 *   obs     [dev] f32 [n_rows, obs_stride]   zero-padded observation rows (hh_step's obs, hh_step_begin's opp_obs, pilot_obs)
 *   sel     [dev] u8  [n_rows]               selector bytes (through the LUT); NULL = the same selectors as the previous call (a fixed
 *                                            network per unit slot, e.g. LowLevelEnv agents 1 / 2): the row lists are re-used, no binning pass
 *   actions [dev] i8  [n_rows, 4]            MultiDiscrete([13,9,2,2]) arg-max per component (4th = 0 for type-2 nets)
 *   logits  [dev] f32 [n_rows, 32]           optional (NULL): the actor's logits, zero padded; rows without a network untouched
 * Everything is ordered on `stream`; no host synchronisation (HIP-graph capturable). */
int hh_policy_act(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, const uint8_t *sel, int8_t *actions,
                  float *logits, void *stream);

/* The networks inside the env step (env_base.py:349-398 within env_hier.py:114-140 and env_hetero.py:160-172): bind a bank to a world
 * and the kernels that emit policy rows bin them by network themselves — through this bank's LUT into its row lists — while they
 * still hold the selector in a register: hh_hl_begin / hh_hl_agents_act / hh_hl_tick for the HighLevelEnv pilots (rows [N, 6]; the variant-row
 * launches hh_hl_begin_variants / hh_hl_act_tick: rows [N, 15], max_rows >= 15 N),
 * hh_step_begin for the frozen opponents of LowLevelEnv levels 4-5 (rows [N, n_opps]; selector = policy type | aircraft type << 2,
 * + 16 (k - 3) for the policy set k of the arena's level-5 draw, i.e. 5 / 9 at level 4 and 5 / 9, 21 / 25, 38 / 42 at level 5).
 * hh_policy_act_binned then runs the forward over those lists with no binning pass, and the last workgroup to read the row counters
 * clears them for the next phase (hh_hl_end / hh_reset drop rows nobody consumed).  Contract: after binding, follow every emitting
 * launch whose rows are wanted with ONE hh_policy_act_binned (obs = the rows that launch wrote, n_rows = n_arenas x 6 | n_arenas x
 * n_opps, obs_stride = 30) before the next emitting launch; rows without a network (dead units, finished arenas) keep whatever their
 * action bytes held (the world ignores them).  The bank must have its networks and LUT loaded, live on the world's device and have
 * max_rows >= the rows of one call; one world per bank (binding a second world moves the binding).  hh_bind_policy(w, NULL)
 * unbinds; destroying either side unbinds too.  Mixing hh_policy_act (with selectors) on a bound bank is allowed between steps. */
struct hh_world;
int hh_bind_policy(struct hh_world *w, hh_policy *p);
int hh_policy_act_binned(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, int8_t *actions, float *logits, void *stream);
/* the same for the variant-row phases (hh_abi.h: hh_hl_begin_variants / hh_hl_act_tick; n_rows = n_arenas x 15): live_rows = the caller's estimate of
 * the rows that carry a network (< 0: n_rows), which picks the kernel form */
int hh_policy_act_binned_live(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, int8_t *actions, float *logits, int32_t live_rows, void *stream);

/* ---- The TRAINABLE policies inside a PPO rollout (configs[2]).  What RLlib's sampler does per env step for each of
 * train_hetero.py's two policies (train_hetero.py:206-243): model.forward on the observer's dict (central_critic_observer, 162-181:
 * own observation, the other agent's observation, and ZERO action inputs while sampling), TorchMultiCategorical.sample() over the split
 * logits [13,9,2,2] / [13,9,2], its logp (ACTION_LOGP), and model.value_function() (VF_PREDS).  The value branch
 * (models/ac_models_hetero.py Fight1 232-255 + 277-289, Fight2 344-367 + 390-402; Esc1 72-83 + 97-103, Esc2 148-159 + 173-179):
 *     fight:  y = cat(tanh(v1 [own obs | own act]), tanh(v2 [other obs | other act]), y3),  y3 = normalize(t + att_val(t)), t = tanh(v3 [all 57]),
 *             att_val = MultiheadAttention(150, 2) over a sequence of length 1 = out_proj(v_proj(t));   value = val_out(shared_layer(y))
 *     escape: value = val_out(shared_layer(tanh(inp1_val [own obs | own act | other obs | other act])))
 * goes through the SAME shared_layer as the actor; it is evaluated as a second tile kind of the same fused kernel (hh_k_policy_ppo). */
typedef struct hh_critic_weights {
    int32_t kind;          /* HH_NET_*: the network whose value branch this is (goes into the same slot as its actor) */
    /* fight nets: v_w[0] = v1._model.0.weight [175, D_own + A_own], v_w[1] = v2 [175, D_other + A_other], v_w[2] = v3 [150, 57];
     * escape nets: v_w[0] = inp1_val._model.0.weight [500, 66], v_w[1] = v_w[2] = NULL.  Biases alike. */
    const float *v_w[3];
    const float *v_b[3];
    const float *att_in_proj_w, *att_in_proj_b, *att_out_w, *att_out_b; /* att_val: [450,150], [450], [150,150], [150]; NULL for escape nets */
    const float *shared_w, *shared_b;                                  /* shared_layer._model.0 [500,500], [500] (the tensor the actor uses) */
    const float *val_w, *val_b;                                        /* val_out._model.0 [1,500], [1] */
} hh_critic_weights;

/* load (or replace) the value branch of network `slot` (its actor must be loaded: hh_policy_set_net).  Synchronous.
 * hh_policy_set_net and hh_policy_set_critic are a PAIR: the value branch keeps its own permuted copy of the shared layer (the reference
 * has ONE module-level SHARED_LAYER, models/ac_models_hetero.py:21) and the input widths of its kind, so loading a new actor into the slot
 * invalidates it — hh_policy_sample then refuses a value output (HH_E_ARG) until the value branch is loaded again. */
int hh_policy_set_critic(hh_policy *p, int32_t slot, const hh_critic_weights *w);

/* One sampler step for n_rows units = [n_arenas, 2] rows of LowLevelEnv agents (row r's partner — "the other agent" of
 * central_critic_observer — is row r ^ 1), in one launch sequence (binning as in hh_policy_act + the fused forward over actor AND
 * critic tiles):
 *   obs, n_rows, obs_stride, sel   as hh_policy_act (sel == NULL re-uses the row lists of the previous call)
 *   w         the world whose agents these rows are, or NULL.  With a world, component c of row (arena n, slot s) is drawn by inverse
 *             CDF from  u = U(seed, arena_offset + n, episode, steps, unit s + 1, HH_SITE_POLICY_SAMPLE, c)  (hh_rng.h) with the
 *             arena's CURRENT episode / steps counters, i.e. the draw belongs to the step the action is about to be used in.
 *   uniforms  [dev] f64 [n_rows, 4]  overrides the keyed draw (tests; w may then be NULL).  Exactly one of w / uniforms unless greedy.
 *   crit_act  [dev] f32 [n_rows, 4]  each row's OWN action as the critic's act_1_own input, already scaled as
 *             on_postprocess_trajectory does (a0 / 12, a1 / 8, a2, a3: train_hetero.py:138-160; the partner's comes from row r ^ 1);
 *             NULL = zeros, which is what the sampler sees (train_hetero.py:168-177).
 *   greedy    != 0: arg-max instead of a draw (evaluation; logp is then the log-probability of the arg-max).
 *   actions   [dev] i8  [n_rows, 4]   the sampled MultiDiscrete action (4th = 0 for type-2 nets)
 *   logp      [dev] f32 [n_rows]      sum over the components of log softmax(logits_c)[a_c]   (nullable)
 *   vf        [dev] f32 [n_rows]      value_function()                                         (nullable: the critic tiles are skipped)
 *   logits    [dev] f32 [n_rows, 32]  optional, as hh_policy_act
 * Inverse CDF of a component with logits l[0..n): m = max l, e_i = exp(l_i - m), S = sum e_i in index order, t = (float)u * S, the action
 * is the first i whose running sum exceeds t (the last index if none does); logp_c = (l_a - m) - log S.  Rows without a network: action
 * 0, logp / vf untouched.  Everything is ordered on `stream`; no host synchronisation (HIP-graph capturable). */
int hh_policy_sample(hh_policy *p, const float *obs, int32_t n_rows, int32_t obs_stride, const uint8_t *sel, struct hh_world *w,
                     const double *uniforms, const float *crit_act, int32_t greedy, int8_t *actions, float *logp, float *vf, float *logits,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HH_POLICY_H */
