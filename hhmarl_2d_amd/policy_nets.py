"""The frozen pilot / opponent policy networks the reference evaluates INSIDE its environments
(envs/env_base.py:312-398 `_get_policies` / `_policy_actions`): architecture description, weight
containers keyed like the reference's `state_dict()`.  (The plain PyTorch fp32 restatement of the forward passes — the
numerics reference for the fused HIP kernels — is test infrastructure and lives in oracle/policy_ref.py.)

Architectures (models/ac_models_hetero.py; only the ACTOR half is needed for acting):

    Esc1   (29-103)   obs 30: inp1 [0,7)->150 | inp2 [7,25)->250 | inp3 [25,30)->100   -> cat 500
    Esc2   (105-180)  obs 29: inp1 [0,6)->150 | inp2 [6,24)->250 | inp3 [24,29)->100   -> cat 500
    Fight1 (181-291)  obs 26: inp1 [0,12)->200 | inp2 [12,26)->200 | inp3 [0,26)->100 (+ attention) -> cat 500
    Fight2 (293-404)  obs 24: inp1 [0,10)->200 | inp2 [10,24)->200 | inp3 [0,24)->100 (+ attention) -> cat 500
    every FC is tanh; then shared_layer 500->500 tanh; act_out 500 -> 26 (type 1) | 24 (type 2) logits.

The fight nets pass inp3's output through `nn.MultiheadAttention(100, 2, batch_first=True)` over a sequence whose
length is 1 whenever the env calls them (`seq_lens=torch.tensor([1])`, env_base.py:393): the softmax over a single
key is 1, so the attention output is out_proj(v_proj(x)); then x_full = normalize(x_full + att) (L2, eps 1e-12).
`get_torch_action` (env_base.py:373-382) splits the logits [13,9,2,2] / [13,9,2] and takes the arg-max of each
Categorical's probs = arg-max of the logits (first maximum).
"""
import numpy as np

FIGHT1, FIGHT2, ESC1, ESC2 = 0, 1, 2, 3
KIND_NAMES = {FIGHT1: "Fight1", FIGHT2: "Fight2", ESC1: "Esc1", ESC2: "Esc2"}
OBS_DIM = {FIGHT1: 26, FIGHT2: 24, ESC1: 30, ESC2: 29}
N_OUT = {FIGHT1: 26, FIGHT2: 24, ESC1: 26, ESC2: 24}
# (first obs column, last+1, width) of inp1 / inp2 / inp3
INPUTS = {
    FIGHT1: ((0, 12, 200), (12, 26, 200), (0, 26, 100)),
    FIGHT2: ((0, 10, 200), (10, 24, 200), (0, 24, 100)),
    ESC1: ((0, 7, 150), (7, 25, 250), (25, 30, 100)),
    ESC2: ((0, 6, 150), (6, 24, 250), (24, 29, 100)),
}
HAS_ATT = {FIGHT1: True, FIGHT2: True, ESC1: False, ESC2: False}
ACTION_SPLIT = (13, 9, 2, 2)


def actor_keys(kind):
    """state_dict keys (reference module names; SlimFC wraps nn.Linear as `_model.0`) of the actor half -> shapes"""
    (a0, a1, w1), (b0, b1, w2), (c0, c1, w3) = INPUTS[kind]
    keys = {
        "inp1._model.0.weight": (w1, a1 - a0), "inp1._model.0.bias": (w1,),
        "inp2._model.0.weight": (w2, b1 - b0), "inp2._model.0.bias": (w2,),
        "inp3._model.0.weight": (w3, c1 - c0), "inp3._model.0.bias": (w3,),
        "shared_layer._model.0.weight": (500, 500), "shared_layer._model.0.bias": (500,),
        "act_out._model.0.weight": (N_OUT[kind], 500), "act_out._model.0.bias": (N_OUT[kind],),
    }
    if HAS_ATT[kind]:
        keys.update({"att_act.in_proj_weight": (300, 100), "att_act.in_proj_bias": (300,),
                     "att_act.out_proj.weight": (100, 100), "att_act.out_proj.bias": (100,)})
    return keys


def random_weights(kind, seed):
    """Deterministic synthetic weights (the reference's policies/*.pt are not shipped): numpy PCG64 streams, identical
    on every machine, so fixtures store a seed instead of megabytes of matrices.  N(0, 1/fan_in) weights keep the tanh
    layers in their responsive range; N(0, 0.1^2) biases make every bias path matter."""
    rng = np.random.default_rng([int(seed), int(kind)])
    sd = {}
    for k, shp in actor_keys(kind).items():
        if k.endswith("weight"):
            sd[k] = (rng.standard_normal(shp) / np.sqrt(shp[-1])).astype(np.float32)
        else:
            sd[k] = (0.1 * rng.standard_normal(shp)).astype(np.float32)
    return sd


def from_torch_module(module):
    """weights of a loaded reference policy (`torch.load('policies/L3_AC1_fight.pt')`) -> (kind, dict of numpy arrays)"""
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in module.state_dict().items()}
    att = "att_act.in_proj_weight" in sd
    cols1 = sd["inp1._model.0.weight"].shape[1]
    kind = {(True, 12): FIGHT1, (True, 10): FIGHT2, (False, 7): ESC1, (False, 6): ESC2}[(att, cols1)]
    return kind, {k: sd[k] for k in actor_keys(kind)}


# ---------------------------------------------------------------------------------------------------------------------
# The value branch and the sampler step of the TRAINABLE policies (train_hetero.py:206-243; hh_policy_sample, include/hh_policy.h)
#
#   Fight1 (232-255, 277-289) / Fight2 (344-367, 390-402):
#       y  = cat(tanh(v1 [obs_own | act_own]), tanh(v2 [obs_2 | act_2]))                       175 + 175
#       t  = tanh(v3 [obs_own | act_own | obs_2 | act_2]);  y3 = normalize(t + att_val(t))     150   (MultiheadAttention(150, 2), sequence length 1)
#       value = val_out(shared_layer(cat(y, y3)))
#   Esc1 (72-83, 97-103) / Esc2 (148-159, 173-179):  value = val_out(shared_layer(tanh(inp1_val [obs_own | act_own | obs_2 | act_2])))
# own / other observation and action widths per kind (train_hetero.py:183-198)
CRITIC_DIMS = {FIGHT1: (26, 4, 24, 3), FIGHT2: (24, 3, 26, 4), ESC1: (30, 4, 29, 3), ESC2: (29, 3, 30, 4)}


def critic_keys(kind):
    """state_dict keys of the value branch -> shapes (the shared layer is the actor's tensor)"""
    d1, a1, d2, a2 = CRITIC_DIMS[kind]
    if HAS_ATT[kind]:
        return {"v1._model.0.weight": (175, d1 + a1), "v1._model.0.bias": (175,), "v2._model.0.weight": (175, d2 + a2), "v2._model.0.bias": (175,),
                "v3._model.0.weight": (150, d1 + a1 + d2 + a2), "v3._model.0.bias": (150,),
                "att_val.in_proj_weight": (450, 150), "att_val.in_proj_bias": (450,), "att_val.out_proj.weight": (150, 150), "att_val.out_proj.bias": (150,),
                "val_out._model.0.weight": (1, 500), "val_out._model.0.bias": (1,)}
    return {"inp1_val._model.0.weight": (500, d1 + a1 + d2 + a2), "inp1_val._model.0.bias": (500,),
            "val_out._model.0.weight": (1, 500), "val_out._model.0.bias": (1,)}


def random_critic_weights(kind, seed):
    """synthetic value-branch weights, a stream of their own (random_weights' actor tensors stay what the committed fixtures were made with)"""
    rng = np.random.default_rng([int(seed), int(kind), 7])
    sd = {}
    for k, shp in critic_keys(kind).items():
        if k.endswith("weight"):
            sd[k] = (rng.standard_normal(shp) / np.sqrt(shp[-1])).astype(np.float32)
        else:
            sd[k] = (0.1 * rng.standard_normal(shp)).astype(np.float32)
    return sd


def critic_from_torch_module(module, kind):
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in module.state_dict().items()}
    return {k: sd[k] for k in critic_keys(kind)}


def scale_actions(act):
    """on_postprocess_trajectory's scaling of an action into the critic's act inputs (train_hetero.py:138-160): a0 / 12, a1 / 8, a2, a3"""
    a = np.asarray(act, dtype=np.float32).copy()
    a[..., 0] /= 12.0
    a[..., 1] /= 8.0
    return a


def critic_flops_per_row(kind):
    d1, a1, d2, a2 = CRITIC_DIMS[kind]
    n = d1 + a1 + d2 + a2
    macs = ((d1 + a1) * 175 + (d2 + a2) * 175 + n * 150 + 2 * 150 * 150 if HAS_ATT[kind] else n * 500) + 500 * 500 + 500
    return 2 * macs


def flops_per_row(kind):
    (a0, a1, w1), (b0, b1, w2), (c0, c1, w3) = INPUTS[kind]
    macs = (a1 - a0) * w1 + (b1 - b0) * w2 + (c1 - c0) * w3 + 500 * 500 + 500 * N_OUT[kind] + (100 * 100 * 2 if HAS_ATT[kind] else 0)
    return 2 * macs
