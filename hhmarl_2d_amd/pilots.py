"""Stand-ins for the frozen low-level pilot policies that HighLevelEnv runs inside the environment
(envs/env_base.py:312-398).  The reference torch.load()s `policies/L*_AC*_{fight,escape}.pt`, which
are not shipped (.gitignore:5); a deployment passes its own loaded policies as the `pilot` callable.

A pilot is `callable(pilot_obs f32 [N,A,30], pilot_mode u8 [N,A]) -> int8 actions [N,A,4]`
(MultiDiscrete([13,9,2,2]); rows with mode 0 are ignored).  Greedy arg-max per action component is
what the reference does (env_base.py:373-382)."""
import torch


class RandomPilot:
    """uniform actions from a device generator (env-only throughput measurements)"""

    def __init__(self, device, seed=0):
        torch.manual_seed(seed)  # default generator: usable inside a captured HIP graph
        self.hi = torch.tensor([13, 9, 2, 2], device=device)

    def __call__(self, pilot_obs, pilot_mode):
        n, a = pilot_mode.shape
        return (torch.rand((n, a, 4), device=pilot_obs.device) * self.hi).to(torch.int8)


class TapePilot:
    """uniform actions from a tape that is resident in HBM before the step starts (the way bench.py feeds LowLevelEnv):
    `bank` holds `chunks` x `depth` sub-steps of int8 actions [N, A, 4]; `load(k)` copies chunk k into the static buffer the
    calls hand out, one slice per sub-step (the agents' call and the opponents' call of a sub-step get the same tensor —
    each side's rows are read by its own launch).  No kernel runs inside the macro step."""

    def __init__(self, device, n_arenas, n_units, seed=0, depth=16, chunks=4):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        hi = torch.tensor([13, 9, 2, 2], device=device)
        self.bank = (torch.rand((chunks, depth, n_arenas, n_units, 4), device=device, generator=g) * hi).to(torch.int8).contiguous()
        self.static = self.bank[0].clone()
        self.depth, self.calls = depth, 0

    def load(self, k):
        self.static.copy_(self.bank[k % self.bank.shape[0]])
        self.calls = 0

    def __call__(self, pilot_obs, pilot_mode):
        act = self.static[(self.calls // 2) % self.depth]
        self.calls += 1
        return act


class MLPPilot(torch.nn.Module):
    """randomly initialised fight / escape networks with the reference's I/O shapes (obs 30 padded ->
    logits 13+9+2+2, models/ac_models_hetero.py), greedy arg-max decode; batched over all units"""

    def __init__(self, device, hidden=200, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)

        def net():
            m = torch.nn.Sequential(torch.nn.Linear(30, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, hidden), torch.nn.Tanh(),
                                    torch.nn.Linear(hidden, 26))
            for p in m.parameters():
                with torch.no_grad():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3)
            return m
        self.fight, self.esc = net(), net()
        self.to(device)

    @torch.no_grad()
    def forward(self, pilot_obs, pilot_mode):
        n, a, _ = pilot_obs.shape
        x = pilot_obs.reshape(n * a, 30)
        lf, le = self.fight(x), self.esc(x)
        logits = torch.where((pilot_mode.reshape(-1, 1) == 2), le, lf)
        parts = logits.split((13, 9, 2, 2), dim=1)
        act = torch.stack([p.argmax(dim=1) for p in parts], dim=1)
        return act.to(torch.int8).reshape(n, a, 4)
