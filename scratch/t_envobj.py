import sys, time; sys.path.insert(0, ".")
import numpy as np, simurlacra_amd as vs
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
env.reset()
a = np.array([0.5])
for _ in range(50): env.step(a)
t0 = time.perf_counter()
for _ in range(1000): env.step(a)
print("env object: %.1f us per step" % ((time.perf_counter() - t0) * 1e3))
