"""Fixture provenance: every file under tests/golden/ must regenerate, array for array, from the committed generator run on
the real reference.  Only possible in the build container (the reference does not travel), so the test is skipped elsewhere."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists only in the build container")
def test_golden_fixtures_regenerate_from_the_committed_generator():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_env_golden.py"), "--check"], capture_output=True, text=True,
                       timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists only in the build container")
@pytest.mark.parametrize("script", ["gen_policy_golden.py", "gen_critic_golden.py", "gen_fullsize_golden.py"])
def test_network_and_critic_fixtures_regenerate(script):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", script), "--check"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
