import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, oracle_lib as O
from hhmarl_2d_amd.world import World, make_config
from helpers import random_actions
kw=dict(n_arenas=300, seed=99, arena_offset=1000, auto_reset=True, level=1)
g=World(make_config(**kw)); o=O.OracleWorld(O.make_config(**kw))
g.reset(); o.reset(); rng=np.random.default_rng(5)
for t in range(20):
    act=random_actions(rng,(300,),g.n_ctrl)
    g.step(torch.from_numpy(act).cuda()); o.step(act)
    a=g.event_masks(); b=o.event_masks()
    if not np.array_equal(a,b):
        idx=np.argwhere(a!=b)[:,0]
        print('t',t,'idx',idx[:10],[hex(x) for x in a[idx][:10]],[hex(x) for x in b[idx][:10]])
