"""variant.py <name> <units: comma list of family numbers / 'vecsim' / 'mixed'> [extra hipcc flags ...]
Builds scratch/r3/lib_<name>.so: the named translation units recompiled with the extra flags, every other object taken from
the main in-tree build (simurlacra_amd/csrc/build).  Select it at run time with VS_LIB_PATH."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simurlacra_amd.csrc import build as b

name, units, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"lib_{name}.so")
odir = out + ".build"
os.makedirs(odir, exist_ok=True)
hipcc = b.hipcc_path()
objs, todo = [], []
for src, obj, fl in b.units():
    base = os.path.basename(obj)
    key = base.replace("vecsim_family_", "").replace("vecsim_", "").replace(".o", "")
    if key in units or (base == "vecsim.o" and "vecsim" in units):
        o = os.path.join(odir, base)
        todo.append([hipcc, *b.FLAGS, *extra, *fl, "-c", "-o", o, src])
        objs.append(o)
    else:
        objs.append(obj)
with ThreadPoolExecutor(8) as pool:
    for r in pool.map(lambda c: subprocess.run(c, capture_output=True, text=True), todo):
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
print(out)
