import numpy as np, torch, sys
sys.path.insert(0, '.')
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
name='bob'; n=2048; T=40
kw=dict(dt=0.01,max_steps=500)
a=vs.VecSimEnv(name,n,**kw); b=vs.VecSimEnv(name,n,**kw)
for e in (a,b):
    e.set_auto_reset(True, seed=17); e.reset(seed=1)
a.step_random(T, seed=4, record=True)
tr=a.traj(T)
for t in range(T):
    ob=b.get(L.VS_OBS)
    bad=(ob!=tr['obs'][t]).any(axis=1)
    if bad.any():
        i=np.where(bad)[0][0]
        print(t,'nbad',bad.sum(),'lane',i,'b obs',ob[i],'a obs',tr['obs'][t][i],'prev done a',tr['done'][t-1][i] if t else None)
    b.step(torch.from_numpy(tr['act'][t]).cuda())
    print(t,'done a',tr['done'][t].sum(),'done b',b.get(L.VS_DONE).sum())
