"""
TEST INFRASTRUCTURE — golden vectors for the centralised-critic input packing.

Runs the REAL reference callbacks — `central_critic_observer` and `CustomCallback.on_postprocess_trajectory` of
train_hetero.py:113-181 (2-vs-2 low level) and train_hier.py:100-165 (3-vs-3 commander), obtained by calling the
reference's own `get_policy(args)` with a recording stand-in for RLlib's PPOConfig builder — on synthetic episode
batches cut from the committed environment traces, and stores (observations, actions) -> the CUR_OBS rows the critic
finally sees.  What is RLlib's and therefore re-stated here: the flattening of the observer's Dict into one row
(gymnasium sorts Dict keys, RLlib's DictFlatteningPreprocessor concatenates them in that order).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_critic_golden.py [--check]
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_policy_golden as GP  # noqa: E402  (ray model stand-ins)
import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "critic_packing.npz")


class _Recorder:
    """fluent stand-in for PPOConfig(): remembers what the reference hands to RLlib"""

    def __init__(self):
        self.kw = {}

    def __getattr__(self, name):
        def f(*a, **k):
            self.kw[name] = (a, k)
            return self
        return f

    def build(self):
        return self


def install_trainer_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class SampleBatch(dict):
        CUR_OBS, ACTIONS = "obs", "actions"

    class DefaultCallbacks:
        pass

    class ModelCatalog:
        @staticmethod
        def register_custom_model(*a, **k):
            pass

    mod("tensorboard", program=types.SimpleNamespace(TensorBoard=object))
    for n in ("ray.rllib.algorithms", "ray.rllib.policy"):
        if n not in sys.modules:
            mod(n)
    mod("ray.rllib.algorithms.ppo", PPOConfig=_Recorder)
    mod("ray.rllib.algorithms.callbacks", DefaultCallbacks=DefaultCallbacks)
    mod("ray.rllib.policy.policy", PolicySpec=lambda *a, **k: (a, k))
    mod("ray.rllib.policy.sample_batch", SampleBatch=SampleBatch)
    sys.modules["ray.rllib.models"].ModelCatalog = ModelCatalog
    return SampleBatch


def flatten(d):
    """RLlib DictFlatteningPreprocessor over a gymnasium Dict space: keys in sorted order"""
    return np.concatenate([np.asarray(d[k], dtype=np.float32).ravel() for k in sorted(d)])


def reference_callbacks(module_name, args):
    mod = __import__(module_name)
    rec = mod.get_policy(args)
    cb = rec.kw["callbacks"][0][0]()
    observer = rec.kw["multi_agent"][1]["observation_fn"]
    return cb, observer, mod


def generate():
    H.load_reference()            # env stand-ins (gymnasium, ray env base, geographiclib)
    GP.install_ray_stubs()        # model stand-ins
    SampleBatch = install_trainer_stubs()
    if H.REF_ROOT not in sys.path:
        sys.path.insert(0, H.REF_ROOT)
    out = {}
    # ---------------- low level, fight and escape widths
    for mode, trace in (("fight", "env_l3_fight_random.npz"), ("escape", "env_l3_escape_shaping.npz")):
        g = np.load(os.path.join(ROOT, "tests", "golden", trace))
        rows = np.nonzero(g["kind"] == 1)[0][:48]
        d1, d2 = (26, 24) if mode == "fight" else (30, 29)
        args = types.SimpleNamespace(agent_mode=mode, num_workers=1, gpu=0, env_config={}, batch_size=1, mini_batch_size=1)
        cb, observer, _ = reference_callbacks("train_hetero", args)
        obs = g["obs"][rows]                               # [T, 2, D] zero padded rows as the env emits them
        act = g["actions"][rows][:, :2].astype(np.float32)  # [T, 2, 4]
        flat = {}
        for ag in (1, 2):
            flat[ag] = np.stack([flatten(observer({1: obs[t, 0, :d1], 2: obs[t, 1, :d2]})[ag]) for t in range(len(rows))])
        batches = {1: (None, SampleBatch({SampleBatch.ACTIONS: act[:, 0, :4].copy()})), 2: (None, SampleBatch({SampleBatch.ACTIONS: act[:, 1, :3].copy()}))}
        for ag in (1, 2):
            post = SampleBatch({SampleBatch.CUR_OBS: flat[ag].copy()})
            cb.on_postprocess_trajectory(worker=None, episode=None, agent_id=ag, policy_id=None, policies=None, postprocessed_batch=post,
                                         original_batches=batches)
            out[f"ll_{mode}_rows_agent{ag}"] = post[SampleBatch.CUR_OBS].astype(np.float32)
        out[f"ll_{mode}_obs"] = obs.astype(np.float32)
        out[f"ll_{mode}_act"] = act.astype(np.int8)
        print(f"low level {mode}: {len(rows)} steps, row widths {out[f'll_{mode}_rows_agent1'].shape[1]} / {out[f'll_{mode}_rows_agent2'].shape[1]}")
    # ---------------- commander
    g = np.load(os.path.join(ROOT, "tests", "golden", "env_hl_random_pilots.npz"))
    rows = np.nonzero(g["kind"] == 1)[0][:24]
    args = types.SimpleNamespace(num_workers=1, gpu=0, env_config={}, batch_size=1, mini_batch_size=1)
    cb, observer, _ = reference_callbacks("train_hier", args)
    obs = g["obs"][rows]
    cmd = g["cmd"][rows].astype(np.float32)
    flat = {ag: np.stack([flatten(observer({i + 1: obs[t, i] for i in range(3)})[ag]) for t in range(len(rows))]) for ag in (1, 2, 3)}
    batches = {ag: (None, SampleBatch({SampleBatch.ACTIONS: cmd[:, ag - 1].copy()})) for ag in (1, 2, 3)}
    for ag in (1, 2, 3):
        post = SampleBatch({SampleBatch.CUR_OBS: flat[ag].copy()})
        cb.on_postprocess_trajectory(worker=None, episode=None, agent_id=ag, policy_id=None, policies=None, postprocessed_batch=post,
                                     original_batches=batches)
        out[f"hl_rows_agent{ag}"] = post[SampleBatch.CUR_OBS].astype(np.float32)
    out["hl_obs"] = obs.astype(np.float32)
    out["hl_act"] = cmd.astype(np.int8)
    print(f"commander: {len(rows)} steps, row width {out['hl_rows_agent1'].shape[1]}")
    return out


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    data = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in data if k not in old.files or not np.array_equal(old[k], data[k])]
        print("critic fixtures reproduce" if not bad else f"DIFFERENT: {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
