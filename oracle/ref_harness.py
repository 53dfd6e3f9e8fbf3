"""
TEST INFRASTRUCTURE — runs ONLY in the build container (needs /root/reference).

Imports the *unchanged* reference environment (envs/env_hetero.py LowLevelEnv,
envs/env_hier.py HighLevelEnv and the warsim simulator) behind minimal stand-ins for the
third-party packages this image lacks, and replaces its unseeded randomness by the keyed
tape of include/hh_rng.h.  Used by oracle/gen_env_golden.py to record golden step traces.

What is stubbed, and why that does not weaken the pin:
  * ray.rllib MultiAgentEnv, gymnasium.spaces, cairo, cartopy: import-time only; none of them
    computes anything on the step/reset/observation path.
  * ScenarioPlotter.__init__: rendering only (needs network for coastlines).
  * geographiclib.geodesic.Geodesic.WGS84.{Inverse,Direct}: the package is absent, so the
    arithmetic comes from oracle/geodesic_ref.py (Karney 2013, pinned against an independent
    mpmath ODE integration).  The reference's env/simulator LOGIC is the real reference.
  * random.uniform / randint / choices and CmanoSimulator.rnd_gen: mapped by call site
    (file, line) -> HH_SITE_* and the current unit id to the keyed generator.
Nothing of the reference is copied: it is imported from where it lies, with
sys.dont_write_bytecode set so no __pycache__ is written into the read-only tree.
"""
import bisect
import itertools
import linecache
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
MASK64 = (1 << 64) - 1

# ---------------------------------------------------------------- keyed RNG (mirror of hh_rng.h)
SITES = dict(
    RESET_SIDE=1, RESET_X=2, RESET_Y=3, RESET_HDG=4, RESET_TYPE=5, RESET_L5K=6, MISSILE_WAIT=7,
    L12_COIN=8, L2_PERIOD=9, L2_TURN=10, L2_SPEED=11, L3_ESC_COIN=12, L3_ESC_TIME=13, ESC_HDG=14,
    ESC_SPEED=15, ESC_FIRE=16, HC_SPEED1=17, HC_R=18, HC_SPEED2=19, ROCKET_NOISE=20, CANNON=21,
    HL_FIGHT=22, HL_OTHER=23, HL_PICK=24,
)


def mix64(z):
    z &= MASK64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & MASK64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & MASK64
    z ^= z >> 31
    return z


def arena_key(seed, arena):
    return mix64(seed ^ ((0x9E3779B97F4A7C15 * (arena + 1)) & MASK64))


def tick_key(akey, episode, tick):
    return mix64(akey ^ ((episode << 32) | (tick & 0xFFFFFFFF)))


def u01(tkey, unit, site, sub=0):
    h = mix64((tkey + ((unit << 32) | (site << 16) | sub)) & MASK64)
    return (h >> 11) * (2.0 ** -53)


class KeyedTape:
    """Resolves a draw made somewhere inside the reference to a keyed uniform."""

    def __init__(self, seed, arena):
        self.akey = arena_key(seed, arena)
        self.episode = 0
        self.env = None
        self.log = []  # (tick, unit, site, sub) of every draw, for collision checks

    def u(self, unit, site, sub=0):
        tick = self.env.steps
        self.log.append((self.episode, tick, unit, site, sub))
        return u01(tick_key(self.akey, self.episode, tick), unit, site, sub)


def _locate(tape):
    """Walk up from the patched function to the reference frame that drew; -> (site, unit, sub)."""
    f = sys._getframe(2)
    while f is not None and not f.f_code.co_filename.startswith(REF_ROOT):
        f = f.f_back
    if f is None:
        raise RuntimeError("random draw from outside the reference")
    fn = os.path.basename(f.f_code.co_filename)
    line = f.f_lineno
    L = f.f_locals
    S = SITES
    if fn == "env_base.py":
        if line == 555:
            return S["RESET_SIDE"], 0, 0
        if line == 560:
            n_ag = L["self"].args.num_agents
            return S["RESET_TYPE"], (L["i"] + 1) + (0 if L["group"] == "agent" else n_ag), 0
        if line == 230:
            return S["MISSILE_WAIT"], L["unit_id"], 0
        if f.f_code.co_name == "_sample_state":
            return _sample_site(f)
    if fn == "env_hier.py":
        if f.f_code.co_name == "_sample_state":
            return _sample_site(f)
        if line == 176:
            return S["HL_FIGHT"], L["i"], 0
        if line == 179:
            return S["HL_OTHER"], L["i"], 0
        if line == 181:
            return S["HL_PICK"], L["i"], 0
    if fn == "env_hetero.py":
        if line == 57:
            return S["RESET_L5K"], 0, 0
        if line in (119, 132):
            return S["L12_COIN"], L["unit_id"], 0
        if line == 127:
            return S["L2_PERIOD"], L["unit_id"], 0
        if line == 128:
            return S["L2_TURN"], L["unit_id"], 0
        if line == 130:
            return S["L2_SPEED"], L["unit_id"], 0
        if line == 140:
            return S["L3_ESC_COIN"], L["unit_id"], 0
        if line == 142:
            return S["L3_ESC_TIME"], L["unit_id"], 0
        if line in (235, 237, 240, 242):
            return S["ESC_HDG"], L["unit"].id, 0
        if line == 243:
            return S["ESC_SPEED"], L["unit"].id, 0
        if line == 245:
            return S["ESC_FIRE"], L["unit"].id, 0
        if line == 255:
            return S["HC_SPEED1"], L["opp_id"], 0
        if line == 259:
            return S["HC_R"], L["opp_id"], 0
        if line == 265:
            return S["HC_SPEED2"], L["opp_id"], 0
    if fn == "ac1.py":
        if line == 127:
            return S["ROCKET_NOISE"], L["self"].id, 0
        if line in (112, 113):
            return S["CANNON"], L["self"].id, L["unit"].id
    if fn == "ac2.py" and line in (99, 100):
        return S["CANNON"], L["self"].id, L["unit"].id
    raise RuntimeError(f"unmapped random draw at {fn}:{line}")


def _sample_site(f):
    L = f.f_locals
    txt = linecache.getline(f.f_code.co_filename, f.f_lineno).strip()
    n_ag = L["self"].args.num_agents
    unit = (L["i"] + 1) + (0 if L["agent"] == "agent" else n_ag)
    if txt.startswith("x ="):
        return SITES["RESET_X"], unit, 0
    if txt.startswith("y ="):
        return SITES["RESET_Y"], unit, 0
    if txt.startswith("a ="):
        return SITES["RESET_HDG"], unit, 0
    raise RuntimeError(f"unmapped _sample_state draw: {txt}")


class RandomProxy:
    """Stands in for the `random` module inside reference modules."""

    def __init__(self, tape):
        self._tape = tape

    def _u(self):
        site, unit, sub = _locate(self._tape)
        return self._tape.u(unit, site, sub)

    def uniform(self, a, b):
        return a + (b - a) * self._u()

    def randint(self, a, b):
        return a + int(np.floor(self._u() * (b - a + 1)))

    def random(self):
        return self._u()

    def choices(self, population, weights=None, *, cum_weights=None, k=1):
        assert k == 1 and weights is not None
        cum = list(itertools.accumulate(weights))
        total = cum[-1] + 0.0
        return [population[bisect.bisect(cum, self._u() * total, 0, len(population) - 1)]]

    # the reference only constructs Random(None) in CmanoSimulator.__init__; its instance is
    # replaced after every reset (see RefEnv.reset)
    def Random(self, seed=None):
        return self


# ---------------------------------------------------------------- stubs + import
def _install_stubs():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import geodesic_ref

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class MultiAgentEnv:
        def __init__(self):
            pass

    mod("ray")
    mod("ray.rllib")
    mod("ray.rllib.env")
    mod("ray.rllib.env.multi_agent_env", MultiAgentEnv=MultiAgentEnv)

    class _Space:
        def __init__(self, *a, **k):
            self.args = a
            self.kwargs = k

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = shape if shape is not None else np.shape(low)

    class Dict(_Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec)

    class Discrete(_Space):
        def __init__(self, n):
            self.n = n

    spaces = mod("gymnasium.spaces", Box=Box, Dict=Dict, MultiDiscrete=MultiDiscrete, Discrete=Discrete)
    mod("gymnasium", spaces=spaces)
    mod("cairo", FONT_SLANT_NORMAL=0, FONT_WEIGHT_NORMAL=0)
    mod("cartopy")

    class _WGS84:
        @staticmethod
        def Inverse(lat1, lon1, lat2, lon2, outmask=0):
            s12, azi1 = geodesic_ref.inverse(lat1, lon1, lat2, lon2)
            return {"s12": s12, "azi1": azi1}

        @staticmethod
        def Direct(lat1, lon1, azi1, s12, outmask=0):
            lat2, lon2 = geodesic_ref.direct(lat1, lon1, azi1, s12)
            return {"lat2": lat2, "lon2": lon2}

    class Geodesic:
        DISTANCE, AZIMUTH, LATITUDE, LONGITUDE = 1, 2, 4, 8
        WGS84 = _WGS84()

    mod("geographiclib")
    mod("geographiclib.geodesic", Geodesic=Geodesic)


_loaded = {}


def load_reference():
    """Import the reference env classes (once). Returns a dict of the objects the harness needs."""
    if _loaded:
        return _loaded
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (this harness only runs in the build container)")
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    from warsim.scenplotter import scenario_plotter
    scenario_plotter.ScenarioPlotter.__init__ = lambda self, *a, **k: None
    from envs import env_hetero, env_hier, env_base
    _loaded.update(env_hetero=env_hetero, env_hier=env_hier, env_base=env_base)
    return _loaded


def _patch_random(tape):
    proxy = RandomProxy(tape)
    for name, m in list(sys.modules.items()):
        fn = getattr(m, "__file__", None)
        if fn and fn.startswith(REF_ROOT) and hasattr(m, "random"):
            m.random = proxy
    return proxy


def make_args(**kw):
    """The fields the env reads from config.py's Namespace (config.py:17-54,94-107)."""
    level = kw.get("level", 1)
    mode = kw.get("mode", 0)  # 0 = low level, 1 = high level
    d = dict(
        level=level, agent_mode="fight",
        num_agents=2 if mode == 0 else 3, num_opps=2 if mode == 0 else 3,
        map_size=0.3 if mode == 0 else 0.5, glob_frac=0.0, rew_scale=1, esc_dist_rew=False,
        hier_action_assess=True, friendly_kill=True, friendly_punish=False, eval_info=False,
        eval_hl=True, eval_level_ag=5, eval_level_opp=4, hier_opp_fight_ratio=75,
    )
    d.update({k: v for k, v in kw.items() if k != "mode"})
    d["total_num"] = d["num_agents"] + d["num_opps"]
    if "horizon" not in kw:
        d["horizon"] = {1: 150, 2: 200, 3: 300, 4: 350, 5: 400}[level] if mode == 0 else 500
    return types.SimpleNamespace(**d)


# ---------------------------------------------------------------- state extraction
AC_F = ("lat", "lon", "hdg", "spd", "cmd_hdg", "cmd_spd")
AC_I = ("alive", "ac_type", "cannon_remain", "cannon_burst", "cannon_max", "missile_remain", "rocket_max",
        "missile_wait", "has_missile", "target")
RK_F = ("lat", "lon", "hdg", "cmd_hdg", "spd")
RK_I = ("alive", "target", "life", "seq")
AR_I = ("steps", "alive_agents", "alive_opps", "escaping", "escaping_time")


def dump_state(env, n_units):
    """World snapshot in the layout of hh_get_state (include/hh_abi.h)."""
    sim = env.sim
    ac_f = np.zeros((n_units, len(AC_F)))
    ac_i = np.zeros((n_units, len(AC_I)), dtype=np.int32)
    rk_f = np.zeros((n_units, len(RK_F)))
    rk_i = np.zeros((n_units, len(RK_I)), dtype=np.int32)
    units = env._hh_units  # id -> unit object (kept after removal)
    for i in range(1, n_units + 1):
        u = units[i]
        t = env.opp_to_attack.get(i)
        if isinstance(t, list):  # HighLevelEnv keeps sorted target lists; first id is recorded
            t = t[0][0] if t else 0
        ac_f[i - 1] = (u.position.lat, u.position.lon, u.heading, u.speed, u.new_heading, u.new_speed)
        ac_i[i - 1] = (int(sim.unit_exists(i)), u.ac_type, u.cannon_remain_secs, u.cannon_current_burst_secs,
                       u.cannon_max, u.missile_remain, u.rocket_max, env.missile_wait[i],
                       int(bool(u.actual_missile)), t or 0)
        m = u.actual_missile
    for uid, r in sim.active_units.items():
        if uid > n_units:
            s = r.source.id - 1
            rk_f[s] = (r.position.lat, r.position.lon, float(r.heading), float(r.new_heading), float(r.speed))
            rk_i[s] = (1, r.target.id, (sim.utc_time - r.firing_time).seconds, uid - n_units)
    ar_i = np.array([env.steps, env.alive_agents, env.alive_opps, int(env.hardcoded_opps_escaping),
                     env.opps_escaping_time], dtype=np.int32)
    K = 5 if max(env.args.num_agents, env.args.num_opps) > 3 else 3   # include/hh_spec.h: HH_TGT_K_OF
    tgt_id = np.zeros((n_units, K), dtype=np.int32)
    tgt_d = np.zeros((n_units, K))
    for i in range(1, n_units + 1):
        t = env.opp_to_attack.get(i)
        if isinstance(t, list):
            for k, e in enumerate(t[:K]):
                tgt_id[i - 1, k] = e[0]
                tgt_d[i - 1, k] = e[1]
        elif t:
            tgt_id[i - 1, 0] = t
    return dict(ac_f=ac_f, ac_i=ac_i, rk_f=rk_f, rk_i=rk_i, ar_i=ar_i, tgt_id=tgt_id, tgt_d=tgt_d)


class RefEnv:
    """One reference arena driven through the keyed tape."""

    def __init__(self, kind, args, seed, arena, keep_policies=False):
        ref = load_reference()
        self.kind = kind
        self.args = args
        self.tape = KeyedTape(seed, arena)
        _patch_random(self.tape)
        if kind == "low":
            self.env = ref["env_hetero"].LowLevelEnv({"args": args})
        else:
            cls = ref["env_hier"].HighLevelEnv
            if not keep_policies:   # taped pilots: nothing to load (the exported policies are not shipped)
                cls._get_policies = lambda self_, mode: None
            self.env = cls({"args": args})
        self.tape.env = self.env
        self.n_units = args.total_num
        self.n_agents = args.num_agents

    def reset(self):
        self.tape.episode += 1
        self.env.steps = 0  # so that reset draws are keyed at tick 0 (env_base.py:66 does the same first)
        _patch_random(self.tape)
        obs, _ = self.env.reset()
        self.env.sim.rnd_gen = RandomProxy(self.tape)
        self.env._hh_units = {i: self.env.sim.get_unit(i) for i in range(1, self.n_units + 1)}
        return obs

    def step(self, action_dict):
        _patch_random(self.tape)
        return self.env.step(action_dict)

    def inject(self, units=None, rockets=(), steps=None):
        """Put the live reference objects into a hand-built situation (edge-case fixtures): `units` maps
        unit id -> dict of fields (lat, lon, hdg, spd, cmd_hdg, cmd_spd, burst, cannon_remain, missile_remain,
        missile_wait; alive=False removes the unit the way a kill would); `rockets` spawns Rocket objects the way
        Rafale.fire_missile does (ac1.py:76-79) with a given age.  Afterwards the env's own state() refreshes
        opp_to_attack exactly as the end of a step would."""
        from datetime import timedelta
        e, sim = self.env, self.env.sim
        ref_sim = sys.modules["simulator.cmano_simulator"]
        Rocket = sys.modules["simulator.rocket_unit"].Rocket
        for uid, f in (units or {}).items():
            u = e._hh_units[uid]
            if "lat" in f: u.position.lat = float(f["lat"])
            if "lon" in f: u.position.lon = float(f["lon"])
            if "hdg" in f: u.heading = float(f["hdg"]); u.new_heading = float(f.get("cmd_hdg", f["hdg"]))
            if "spd" in f: u.speed = float(f["spd"]); u.new_speed = float(f.get("cmd_spd", f["spd"]))
            if "cmd_hdg" in f: u.new_heading = float(f["cmd_hdg"])
            if "cmd_spd" in f: u.new_speed = float(f["cmd_spd"])
            if "burst" in f: u.cannon_current_burst_secs = f["burst"]
            if "cannon_remain" in f: u.cannon_remain_secs = f["cannon_remain"]
            if "missile_remain" in f: u.missile_remain = f["missile_remain"]
            if "missile_wait" in f: e.missile_wait[uid] = f["missile_wait"]
            if f.get("alive", True) is False and sim.unit_exists(uid):
                sim.remove_unit(uid)
                if uid <= self.n_agents: e.alive_agents -= 1
                else: e.alive_opps -= 1
        for r in rockets:
            src, tgt = e._hh_units[r["source"]], e._hh_units[r["target"]]
            m = Rocket(ref_sim.Position(float(r["lat"]), float(r["lon"]), src.position.alt), float(r["hdg"]),
                       sim.utc_time - timedelta(seconds=int(r.get("life", 0))), tgt, src, src.friendly_check)
            m.new_heading = float(r.get("cmd_hdg", r["hdg"]))
            sim.add_unit(m)
            src.actual_missile = m
        if steps is not None:
            e.steps = int(steps)
        return e.state()

    def obs_array(self, obs, dim):
        out = np.zeros((self.n_agents, dim), dtype=np.float32)
        for i in range(1, self.n_agents + 1):
            v = obs[i]
            out[i - 1, : len(v)] = v
        return out

    def state(self):
        return dump_state(self.env, self.n_units)
