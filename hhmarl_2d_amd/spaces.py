"""observation/action space objects.  gymnasium's are used when it is installed (so RLlib sees the
real types); otherwise minimal attribute holders with the same fields the reference's callers read
(envs/env_hetero.py:29-43, envs/env_hier.py:36-37)."""
import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Dict, Discrete, MultiDiscrete  # noqa: F401
except Exception:  # noqa: BLE001
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = tuple(shape) if shape is not None else self.low.shape
            self.dtype = np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape

        def sample(self, rng=np.random):
            return np.array([rng.randint(n) for n in self.nvec])

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

        def contains(self, x):
            return 0 <= int(x) < self.n

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()
