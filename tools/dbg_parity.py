import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, oracle_lib as O
from hhmarl_2d_amd.world import World, make_config
from helpers import random_actions, pursuit_actions
level=int(sys.argv[1]) if len(sys.argv)>1 else 1
kw=dict(n_arenas=300, seed=99, arena_offset=1000, auto_reset=True, level=level)
g=World(make_config(**kw)); o=O.OracleWorld(O.make_config(**kw))
g.reset(); o.reset(); rng=np.random.default_rng(5)
for t in range(160):
    act=random_actions(rng,(300,),g.n_ctrl)
    obs,rew,val,done=[x.cpu().numpy() for x in g.step(torch.from_numpy(act).cuda())]
    obs_o,rew_o,val_o,done_o=o.step(act)
    bad=False
    for name,a,b in (("rew",rew,rew_o),("val",val,val_o),("done",done,done_o),("obs",obs,obs_o),("mask",g.event_masks(),o.event_masks())):
        if not np.array_equal(a,b):
            idx=np.argwhere(a!=b); print('t',t,name,'n diff',len(idx),'first',idx[:4].tolist(), a[tuple(idx[0])], b[tuple(idx[0])]); bad=True
    sg,so=g.get_state(),o.get_state()
    for k in sg:
        if not np.array_equal(sg[k],so[k]):
            idx=np.argwhere(sg[k]!=so[k]); print('t',t,'state',k,len(idx),idx[:4].tolist(), sg[k][tuple(idx[0])], so[k][tuple(idx[0])]); bad=True
    if bad:
        n=idx[0][0]; print('arena',n,'masks',hex(g.event_masks()[n]),hex(o.event_masks()[n])); print(sg['ac_i'][n]); print(so['ac_i'][n]); print(sg['rk_i'][n], so['rk_i'][n]); break
print("done")
