#!/bin/bash
# round 6: soak of the 2-vs-2 kernel forms touched this round (the OWT two-wave instances: target entries handed over by the output wave) against the CPU oracle
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_soak
{
python tools/soak.py 4096 3000 3 0
python tools/soak.py 8192 1500 3 0
python tools/soak.py 2048 1500 1 0
python tools/soak.py 2048 1500 2 0
python tools/soak.py 4096 1500 3 1 0
SOAK_KW='{"friendly_punish": true, "glob_frac": 0.3}' python tools/soak.py 4096 1500 3 0
SOAK_KW='{"friendly_kill": false, "rew_scale": 2.0}' python tools/soak.py 1000 1500 3 0
python tools/soak.py 4096 1000 3 1 1
python tools/soak.py 32768 600 3 0
python tools/soak.py 262144 120 3 0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_soak/soak.log
