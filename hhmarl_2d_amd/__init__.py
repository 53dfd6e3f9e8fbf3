"""hhmarl_2d_amd — MI355X-native batched 2-D air-combat environment (drop-in for the
step()/reset()/observation path of IDSIA/hhmarl_2D's LowLevelEnv / HighLevelEnv)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
