import sys
sys.path.insert(0, '.')
import simurlacra_amd as vs
import torch
def cycles(n):
    for k in range(n):
        e = vs.VecSimEnv("qq-su", 65536, 0.004, 4000)
        e.set_auto_reset(True, seed=1); e.reset(seed=k)
        e.step_random(20, seed=3, record=(k % 2 == 0))
        e.set_act_pipeline(delay=2); e.step_random(5, seed=3)
        e.close()
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]
base = cycles(5)
for n in (100, 200, 400):
    free = cycles(n)
    print("after +%d cycles: device memory in use grew by %.1f MB" % (n, (base - free) / 1e6))
