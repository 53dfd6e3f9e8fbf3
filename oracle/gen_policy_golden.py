"""
TEST INFRASTRUCTURE — golden vectors for the frozen pilot / opponent networks.

Instantiates the REAL reference model classes (models/ac_models_hetero.py: Fight1, Fight2, Esc1, Esc2, imported
unchanged from /root/reference) behind import stand-ins for the parts of ray.rllib they subclass, loads deterministic
synthetic weights into them (hhmarl_2d_amd.policy_nets.random_weights: the shipped repo has no policies/*.pt), and
calls them exactly the way the environment does (envs/env_base.py:349-398 `_policy_actions`: batch of one, dummy
centralised-critic inputs, seq_lens = [1], arg-max per action component).  Records (obs -> logits, action) per net into
tests/golden/policy_nets.npz; the weights are NOT stored (4 x 1 MB of noise), only their seed.

What is stubbed: ray.rllib's TorchModelV2 / RecurrentNetwork (constructor bookkeeping only), SlimFC (= nn.Linear +
activation, ray/rllib/models/torch/misc.py), add_time_dimension ([B*T, F] -> [B, T, F]), override, try_import_torch.
The forward() code that runs is the reference's own.

Round 4 adds tests/golden/policy_value.npz — the sampler's view of the TRAINABLE policies (train_hetero.py:206-243): the same
reference classes, loaded with synthetic actor AND value-branch weights, called with central_critic_observer's full dict (the other
agent's observation; action inputs zero as while sampling on half of the rows, scaled actions as on_postprocess_trajectory writes them
on the other half), `forward()` then `value_function()`.  Recorded per row: logits, value, the log-probability of a given action (the
sum of the components' Categorical(logits).log_prob, which is what RLlib's TorchMultiCategorical.logp computes), and for a recorded
uniform tape the inverse-CDF action of hh_policy_sample's definition evaluated in float64 on the REFERENCE's logits.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_policy_golden.py [--check]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from hhmarl_2d_amd import policy_nets as PN  # noqa: E402
import policy_ref as PR  # noqa: E402

REF_ROOT = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "policy_nets.npz")
OUT_VALUE = os.path.join(ROOT, "tests", "golden", "policy_value.npz")
VALUE_ROWS = 64
SEED, ROWS = 20240917, 96


def install_ray_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ModelV2:
        pass

    class TorchModelV2(ModelV2):
        def __init__(self, obs_space, action_space, num_outputs, model_config, name):
            self.obs_space, self.action_space, self.num_outputs, self.model_config, self.name = obs_space, action_space, num_outputs, model_config, name

    class RecurrentNetwork(TorchModelV2):
        pass

    class SlimFC(nn.Module):   # ray/rllib/models/torch/misc.py: Linear (+ activation) in self._model
        def __init__(self, in_size, out_size, initializer=None, activation_fn=None, use_bias=True, bias_init=0.0):
            super().__init__()
            lin = nn.Linear(in_size, out_size, bias=use_bias)
            if initializer is not None:
                initializer(lin.weight)
            if use_bias:
                nn.init.constant_(lin.bias, bias_init)
            layers = [lin] + ([activation_fn()] if activation_fn is not None else [])
            self._model = nn.Sequential(*layers)

        def forward(self, x):
            return self._model(x)

    def add_time_dimension(padded_inputs, *, seq_lens, framework="torch", time_major=False):
        b = seq_lens.shape[0]
        t = padded_inputs.shape[0] // b
        return padded_inputs.reshape((b, t) + tuple(padded_inputs.shape[1:]))

    def override(cls):
        return lambda f: f

    for n in ("ray", "ray.rllib", "ray.rllib.models", "ray.rllib.models.torch", "ray.rllib.utils", "ray.rllib.policy"):
        mod(n)
    mod("ray.rllib.models.modelv2", ModelV2=ModelV2)
    mod("ray.rllib.models.torch.misc", SlimFC=SlimFC)
    mod("ray.rllib.models.torch.torch_modelv2", TorchModelV2=TorchModelV2)
    mod("ray.rllib.models.torch.recurrent_net", RecurrentNetwork=RecurrentNetwork)
    mod("ray.rllib.utils.annotations", override=override)
    mod("ray.rllib.utils.framework", try_import_torch=lambda: (torch, nn))
    mod("ray.rllib.policy.rnn_sequencing", add_time_dimension=add_time_dimension)


def reference_models():
    sys.dont_write_bytecode = True
    install_ray_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from models import ac_models_hetero as M
    return M


def env_style_call(model, kind, obs_row):
    """envs/env_base.py:357-396: dummy critic inputs, batch of one, seq_lens [1]; get_torch_action's arg-max"""
    ac1 = kind in (PN.FIGHT1, PN.ESC1)
    fight = kind in (PN.FIGHT1, PN.FIGHT2)
    other = (24 if fight else 29) if ac1 else (26 if fight else 30)
    inp = {"obs_1_own": torch.tensor(np.expand_dims(obs_row, axis=0)), "obs_2": torch.zeros((1, other)),
           "act_1_own": torch.zeros((1, 4 if ac1 else 3)), "act_2": torch.zeros((1, 3 if ac1 else 4))}
    with torch.no_grad():
        out = model(input_dict={"obs": inp}, state=[torch.tensor(0)], seq_lens=torch.tensor([1]))
    logits = out[0]
    in_lens = (13, 9, 2, 2) if ac1 else (13, 9, 2)
    cats = [torch.distributions.categorical.Categorical(logits=p) for p in logits.split(in_lens, dim=1)]
    action = torch.stack([torch.argmax(c.probs, -1) for c in cats], dim=1)[0].numpy()
    return logits[0].numpy(), action


def synth_obs(rng, kind, rows):
    """observation-like rows: entries in [0,1], binary flags, a few all-zero blocks (dead friend / missing second opponent)"""
    d = PN.OBS_DIM[kind]
    x = rng.random((rows, d)).astype(np.float32)
    x[:, d - 5:] *= (rng.random((rows, 1)) > 0.25)          # friend block zeroed
    flags = rng.integers(0, d, size=(rows, 3))
    for r in range(rows):
        x[r, flags[r]] = rng.integers(0, 2, size=3)
    return x


def generate():
    M = reference_models()
    classes = {PN.FIGHT1: M.Fight1, PN.FIGHT2: M.Fight2, PN.ESC1: M.Esc1, PN.ESC2: M.Esc2}
    out = {"seed": np.array(SEED)}
    for kind, cls in classes.items():
        model = cls(None, None, PN.N_OUT[kind], {}, PN.KIND_NAMES[kind])
        sd = PN.random_weights(kind, SEED)
        full = model.state_dict()
        for k, v in sd.items():
            assert full[k].shape == v.shape, (k, full[k].shape, v.shape)
            full[k] = torch.from_numpy(v)
        model.load_state_dict(full)
        model.eval()
        rng = np.random.default_rng([SEED, 100 + kind])
        obs_rows, logit_rows, act_rows = [], [], []
        while len(obs_rows) < ROWS:
            o = synth_obs(rng, kind, 1)[0]
            lg, ac = env_style_call(model, kind, o)
            # near-ties cannot be pinned across summation orders: keep rows whose every arg-max wins by > 1e-3
            parts = np.split(lg, np.cumsum((13, 9, 2, 2)[: 4 if PN.N_OUT[kind] == 26 else 3])[:-1])
            if min(np.sort(p)[-1] - np.sort(p)[-2] for p in parts) <= 1e-3:
                continue
            a4 = np.zeros(4, dtype=np.int8)
            a4[: len(ac)] = ac
            obs_rows.append(o); logit_rows.append(lg); act_rows.append(a4)
        name = PN.KIND_NAMES[kind].lower()
        out[f"obs_{name}"] = np.stack(obs_rows)
        out[f"logits_{name}"] = np.stack(logit_rows).astype(np.float32)
        out[f"act_{name}"] = np.stack(act_rows)
        print(f"{name}: {ROWS} rows, logits in [{out[f'logits_{name}'].min():.3f}, {out[f'logits_{name}'].max():.3f}], "
              f"distinct actions {len({tuple(a) for a in act_rows})}")
    return out


def random_actions(rng, rows, ncomp):
    a = np.zeros((rows, 4), dtype=np.int8)
    for i, hi in enumerate((13, 9, 2, 2)[:ncomp]):
        a[:, i] = rng.integers(0, hi, rows)
    return a


def generate_value():
    """forward() + value_function() of the reference classes on the observer's full dict; logp of given actions; inverse-CDF draws"""
    M = reference_models()
    classes = {PN.FIGHT1: M.Fight1, PN.FIGHT2: M.Fight2, PN.ESC1: M.Esc1, PN.ESC2: M.Esc2}
    partner = {PN.FIGHT1: PN.FIGHT2, PN.FIGHT2: PN.FIGHT1, PN.ESC1: PN.ESC2, PN.ESC2: PN.ESC1}
    out = {"seed": np.array(SEED)}
    for kind, cls in classes.items():
        model = cls(None, None, PN.N_OUT[kind], {}, PN.KIND_NAMES[kind])
        sd, csd = PN.random_weights(kind, SEED), PN.random_critic_weights(kind, SEED)
        full = model.state_dict()
        for k, v in {**sd, **csd}.items():
            assert full[k].shape == v.shape, (k, full[k].shape, v.shape)
            full[k] = torch.from_numpy(v)
        assert set(full) == set(sd) | set(csd), sorted(set(full) ^ (set(sd) | set(csd)))   # nothing of the module is left at its random init
        model.load_state_dict(full)
        model.eval()
        d1, a1, d2, a2 = PN.CRITIC_DIMS[kind]
        rng = np.random.default_rng([SEED, 200 + kind])
        own = synth_obs(rng, kind, VALUE_ROWS)
        oth = synth_obs(rng, partner[kind], VALUE_ROWS)
        act_own, act_oth = random_actions(rng, VALUE_ROWS, a1), random_actions(rng, VALUE_ROWS, a2)
        zero = np.arange(VALUE_ROWS) % 2 == 0                      # the sampler's rows: action inputs are zeros (train_hetero.py:168-177)
        ca_own, ca_oth = PN.scale_actions(act_own), PN.scale_actions(act_oth)
        ca_own[zero] = 0.0
        ca_oth[zero] = 0.0
        given = random_actions(rng, VALUE_ROWS, a1)
        u = rng.random((VALUE_ROWS, 4))
        logits, value = [], []
        for r in range(VALUE_ROWS):
            inp = {"obs_1_own": torch.from_numpy(own[r:r + 1]), "obs_2": torch.from_numpy(oth[r:r + 1]),
                   "act_1_own": torch.from_numpy(ca_own[r:r + 1, :a1]), "act_2": torch.from_numpy(ca_oth[r:r + 1, :a2])}
            with torch.no_grad():
                lg = model(input_dict={"obs": inp}, state=[torch.tensor(0)], seq_lens=torch.tensor([1]))[0]
                vf = model.value_function()
            logits.append(lg[0].numpy())
            value.append(float(vf[0]))
        logits = np.stack(logits).astype(np.float32)
        logp = PR.multicategorical_logp(logits, given, PN.N_OUT[kind]).numpy().astype(np.float32)
        drawn, drawn_logp, margin = PR.inverse_cdf_actions(logits, u, PN.N_OUT[kind])
        name = PN.KIND_NAMES[kind].lower()
        out.update({f"obs_own_{name}": own, f"obs_other_{name}": oth, f"crit_act_own_{name}": ca_own, f"crit_act_other_{name}": ca_oth,
                    f"logits_{name}": logits, f"value_{name}": np.asarray(value, dtype=np.float32), f"given_{name}": given, f"logp_given_{name}": logp,
                    f"u_{name}": u, f"drawn_{name}": drawn, f"drawn_logp_{name}": drawn_logp.astype(np.float32), f"margin_{name}": margin})
        print(f"{name}: value in [{min(value):.3f}, {max(value):.3f}], logp(given) in [{logp.min():.2f}, {logp.max():.2f}], "
              f"smallest draw margin {margin.min():.2e}")
    return out


def _check(path, data):
    old = np.load(path)
    return [k for k in data if k not in old.files or not np.array_equal(old[k], data[k])]


if __name__ == "__main__":
    data, vdata = generate(), generate_value()
    if "--check" in sys.argv:
        bad = _check(OUT, data) + _check(OUT_VALUE, vdata)
        print("policy fixtures reproduce" if not bad else f"DIFFERENT: {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **data)
    np.savez_compressed(OUT_VALUE, **vdata)
    for pth in (OUT, OUT_VALUE):
        print("wrote", pth, os.path.getsize(pth), "bytes")
