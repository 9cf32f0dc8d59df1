import numpy as np, torch, sys
sys.path.insert(0, '.')
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
name='qcp-su'; n=2048; T=16
kw=dict(dt=0.002,max_steps=8000)
a=vs.VecSimEnv(name,n,**kw); b=vs.VecSimEnv(name,n,**kw)
a.reset(seed=1); b.reset(seed=1)
a.step_random(T, seed=4, record=True)
tr=a.traj(T)
for t in range(T):
    ob=b.get(L.VS_OBS)
    d_obs=np.abs(ob-tr['obs'][t]).max(axis=0)
    b.step(torch.from_numpy(tr['act'][t]).cuda())
    d_rew=np.abs(b.get(L.VS_REW)-tr['rew'][t]).max()
    nz=(b.get(L.VS_REW)!=tr['rew'][t]).sum()
    print(t, 'obs diff per dim', d_obs, 'rew diff', d_rew, 'n', nz)
