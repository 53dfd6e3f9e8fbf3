"""The env-facing part of the reference's config.py (Config.args, config.py:17-54 and the derived
fields of set_metrics, config.py:94-107).  Only the fields the environments read are kept; the
trainer/checkpoint plumbing of the reference is out of scope."""
from types import SimpleNamespace

HORIZON_BY_LEVEL = {1: 150, 2: 200, 3: 300, 4: 350, 5: 400}  # config.py:94-95


def make_args(mode=0, **kw):
    """mode 0 = low-level (2-vs-2) training, 1 = high-level (3-vs-3), 2 = evaluation (config.py:12-16)."""
    d = dict(
        level=1, horizon=500, agent_mode="fight",
        num_agents=2 if mode == 0 else 3, num_opps=2 if mode == 0 else 3,
        hier_opp_fight_ratio=75, map_size=0.3 if mode == 0 else 0.5,
        glob_frac=0.0, rew_scale=1, esc_dist_rew=False, hier_action_assess=True,
        friendly_kill=True, friendly_punish=False, eval_info=(mode == 2), eval_hl=True,
        eval_level_ag=5, eval_level_opp=4,
    )
    d.update(kw)
    if "horizon" not in kw:
        d["horizon"] = HORIZON_BY_LEVEL[d["level"]] if mode == 0 else 500
    d["total_num"] = d["num_agents"] + d["num_opps"]
    args = SimpleNamespace(**d)
    args.env_config = {"args": args}
    return args
