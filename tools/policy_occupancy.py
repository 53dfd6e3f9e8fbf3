import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hhmarl_2d_amd import _lib as L
torch.zeros(1).cuda()
lib = L.lib()
for w, n in ((0, "hh_k_policy_h<1>"), (1, "hh_k_policy_h<2>"), (3, "hh_k_policy_w16<4>"), (4, "hh_k_policy_ppo")):   # 2 was hh_k_policy_w (retired in round 6)
    v = C.c_int32(0)
    rc = lib.hh_policy_occupancy(w, C.byref(v))
    print(n, "rc", rc, "workgroups per CU:", v.value)
