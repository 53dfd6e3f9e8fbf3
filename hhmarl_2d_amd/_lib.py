"""ctypes binding of libhh_world.so (include/hh_abi.h).  Fails loudly when the HIP library is
missing: there is NO CPU fallback in the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HH_WORLD_LIB") or os.path.join(HERE, "lib", "libhh_world.so")  # override only for A/B experiments

ENV_LOWLEVEL, ENV_HIGHLEVEL = 0, 1
MODE_FIGHT, MODE_ESCAPE = 0, 1
OPP_MODE_EPISODE = -1  # hh_step_begin: every arena observes in the mode of its own level-5 draw
ACF_K, ACI_K, RKF_K, RKI_K, ARI_K, TGT_K = 6, 10, 4, 4, 6, 3


def hl_slots(n_agents, n_opps):
    """unit slots of a HighLevelEnv arena (include/hh_spec.h: HH_HL_SLOTS)"""
    return 10 if max(n_agents, n_opps) > 3 else 6


def tgt_k_of(n_agents, n_opps):
    """entries of a unit's stored target list in the state views (include/hh_spec.h: HH_TGT_K_OF)"""
    return 5 if max(n_agents, n_opps) > 3 else TGT_K
EVAL_KEYS = ("agents_win", "opps_win", "draw", "agent_fight", "agent_escape", "opp_fight", "opp_escape", "agent_steps", "opp_steps",
             "opp1", "opp2", "opp3")  # columns of hh_eval_info (env_base.py:104-106)


class HHConfig(C.Structure):
    """hh_config (include/hh_abi.h); field order is ABI."""
    _fields_ = [
        ("n_arenas", C.c_int32), ("env_kind", C.c_int32), ("n_agents", C.c_int32), ("n_opps", C.c_int32),
        ("level", C.c_int32), ("agent_mode", C.c_int32), ("horizon", C.c_int32), ("friendly_kill", C.c_int32),
        ("friendly_punish", C.c_int32), ("esc_dist_rew", C.c_int32), ("hier_action_assess", C.c_int32),
        ("hier_opp_fight_ratio", C.c_int32), ("auto_reset", C.c_int32), ("ext_opp_actions", C.c_int32),
        ("opp_side_selector", C.c_int32), ("reserved0", C.c_int32),
        ("map_size", C.c_double), ("glob_frac", C.c_double), ("rew_scale", C.c_double),
        ("seed", C.c_uint64), ("arena_offset", C.c_uint64),
    ]


class HHStateView(C.Structure):
    _fields_ = [
        ("ac_f", C.POINTER(C.c_double)), ("ac_i", C.POINTER(C.c_int32)), ("rk_f", C.POINTER(C.c_double)),
        ("rk_i", C.POINTER(C.c_int32)), ("ar_i", C.POINTER(C.c_int32)), ("tgt_id", C.POINTER(C.c_int32)),
        ("tgt_d", C.POINTER(C.c_double)),
    ]


class HHNetWeights(C.Structure):
    """hh_net_weights (include/hh_policy.h): host pointers to one network's tensors, nn.Linear layout"""
    _fields_ = [("kind", C.c_int32), ("inp_w", C.c_void_p * 3), ("inp_b", C.c_void_p * 3),
                ("att_in_proj_w", C.c_void_p), ("att_in_proj_b", C.c_void_p), ("att_out_w", C.c_void_p), ("att_out_b", C.c_void_p),
                ("shared_w", C.c_void_p), ("shared_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


class HHCriticWeights(C.Structure):
    """hh_critic_weights (include/hh_policy.h): host pointers to one network's value branch, nn.Linear layout"""
    _fields_ = [("kind", C.c_int32), ("v_w", C.c_void_p * 3), ("v_b", C.c_void_p * 3),
                ("att_in_proj_w", C.c_void_p), ("att_in_proj_b", C.c_void_p), ("att_out_w", C.c_void_p), ("att_out_b", C.c_void_p),
                ("shared_w", C.c_void_p), ("shared_b", C.c_void_p), ("val_w", C.c_void_p), ("val_b", C.c_void_p)]


EXPORTS = ["hh_world_create", "hh_world_destroy", "hh_last_error", "hh_obs_dim", "hh_n_ctrl", "hh_reset", "hh_step",
           "hh_rollout", "hh_episode_stats", "hh_get_state", "hh_set_state", "hh_get_event_masks", "hh_observe",
           "hh_hl_begin", "hh_hl_agents_act", "hh_hl_tick", "hh_hl_end", "hh_step_begin", "hh_step_finish", "hh_gae", "hh_hl_commands",
           "hh_episode_stats_packed", "hh_hl_tick_count", "hh_rollout_kernel_name", "hh_opp_policy", "hh_eval_info", "hh_arena_status", "hh_hl_rollout", "hh_trace_enable", "hh_trace_read",
           "hh_policy_create", "hh_policy_destroy", "hh_policy_set_net", "hh_policy_set_lut", "hh_policy_set_tile_rows", "hh_policy_act",
           "hh_bind_policy", "hh_policy_act_binned", "hh_kernel_instance", "hh_gae_rllib", "hh_math_eval",
           "hh_policy_set_critic", "hh_policy_sample", "hh_policy_kernel_name", "hh_action_faults", "hh_action_tape_uniform",
           "hh_hl_begin_variants", "hh_hl_act_tick", "hh_policy_act_binned_live"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The MI355X world has no CPU fallback.")
        import torch  # noqa: F401  — BEFORE the library: torch ships its own HIP runtime; a process that loads libhh_world.so (linked against the system's) first and
        #                  imports torch afterwards ends up with two, and hh_world_create then reports "no HIP device"
        L = C.CDLL(LIB_PATH)
        L.hh_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.hh_world_create.argtypes = [C.POINTER(HHConfig), C.c_int, C.POINTER(vp)]
        L.hh_world_destroy.argtypes = [vp]
        L.hh_obs_dim.argtypes = [vp]
        L.hh_n_ctrl.argtypes = [vp]
        L.hh_reset.argtypes = [vp, vp, vp, vp]
        L.hh_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.hh_rollout.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.hh_episode_stats.argtypes = [vp, vp, vp, vp, vp]
        L.hh_get_state.argtypes = [vp, C.POINTER(HHStateView)]
        L.hh_set_state.argtypes = [vp, C.POINTER(HHStateView)]
        L.hh_get_event_masks.argtypes = [vp, vp, vp]
        L.hh_episode_stats_packed.argtypes = [vp, vp, vp]
        L.hh_hl_tick_count.argtypes = [vp, C.POINTER(C.c_uint64), vp]
        L.hh_rollout_kernel_name.argtypes = [vp, C.c_char_p, C.c_int32]
        L.hh_kernel_instance.argtypes = [vp, C.c_int32, C.c_char_p, C.c_int32]
        L.hh_observe.argtypes = [vp, vp, vp]
        L.hh_hl_begin.argtypes = [vp, vp, vp, vp, vp]
        L.hh_hl_agents_act.argtypes = [vp, vp, vp, vp, vp]
        L.hh_hl_tick.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
        L.hh_hl_end.argtypes = [vp, vp, vp, vp, vp, vp]
        L.hh_step_begin.argtypes = [vp, vp, C.c_int32, vp, vp]
        L.hh_step_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.hh_hl_commands.argtypes = [vp, vp]
        L.hh_trace_enable.argtypes = [vp, C.c_int32, C.c_int32]
        L.hh_trace_read.argtypes = [vp, vp, vp]
        L.hh_hl_rollout.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.hh_opp_policy.argtypes = [vp, vp, vp]
        L.hh_eval_info.argtypes = [vp, vp, vp, C.c_int32, vp]
        L.hh_arena_status.argtypes = [vp, vp, vp]
        L.hh_gae.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
        L.hh_gae_rllib.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp]
        L.hh_math_eval.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
        L.hh_policy_create.argtypes = [C.c_int, C.c_int32, C.POINTER(vp)]
        L.hh_policy_destroy.argtypes = [vp]
        L.hh_policy_set_net.argtypes = [vp, C.c_int32, C.POINTER(HHNetWeights)]
        L.hh_policy_set_lut.argtypes = [vp, vp]
        L.hh_policy_set_tile_rows.argtypes = [vp, C.c_int32]
        L.hh_policy_act.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.hh_bind_policy.argtypes = [vp, vp]
        L.hh_policy_act_binned.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp]
        L.hh_policy_act_binned_live.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp]
        L.hh_hl_begin_variants.argtypes = [vp, vp, vp, vp, vp]
        L.hh_hl_act_tick.argtypes = [vp, vp, vp, vp, vp, vp]
        L.hh_policy_set_critic.argtypes = [vp, C.c_int32, C.POINTER(HHCriticWeights)]
        L.hh_policy_sample.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
        L.hh_policy_kernel_name.argtypes = [vp, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
        L.hh_action_faults.argtypes = [vp, vp, C.c_int32, vp]
        L.hh_action_tape_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libhh_world error {rc}: {lib().hh_last_error().decode()}")
