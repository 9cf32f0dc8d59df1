"""Build libvecsim.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is eleven translation units -- the C-ABI (vecsim.hip), one per env family (vecsim_family.hip with
-DVS_FAMILY=n) and the mixed-batch kernels (vecsim_mixed.hip) -- compiled in parallel and linked into one shared object."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
HEADERS = [os.path.join(HERE, "vecsim_kernels.h"), os.path.join(HERE, "vecsim_envs.h"), os.path.join(HERE, "vecsim_dual.h"),
           os.path.join(HERE, "..", "..", "include", "vecsim.h")]
N_FAMILIES = 9
OBJ_DIR = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libvecsim.so")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall",
         # contraction only inside one source expression: every kernel variant (step / fused rollout / broadcast or
         # per-env constants) then rounds identically, which the bit-exactness tests rely on
         "-ffp-contract=on",
         # packed fp32 VALU ops (v_pk_fma_f32 ...) issue slower than the two scalar ops they replace on gfx950
         "-fno-slp-vectorize", "-Wno-unused-function"]


def units():
    """[(source, object, extra flags)]"""
    u = [(os.path.join(HERE, "vecsim.hip"), os.path.join(OBJ_DIR, "vecsim.o"), []),
         (os.path.join(HERE, "vecsim_mixed.hip"), os.path.join(OBJ_DIR, "vecsim_mixed.o"), [])]
    for f in range(N_FAMILIES):
        u.append((os.path.join(HERE, "vecsim_family.hip"), os.path.join(OBJ_DIR, f"vecsim_family_{f}.o"), [f"-DVS_FAMILY={f}"]))
    return u


DEPS = HEADERS + [os.path.join(HERE, n) for n in ("vecsim.hip", "vecsim_family.hip", "vecsim_mixed.hip")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvecsim cannot be built (there is no CPU fallback)")


def _newer(path, than):
    return not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in than)


def is_stale():
    return _newer(OUT, DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None, jobs=None):
    out = out or OUT
    if not force and not extra_flags and out == OUT and not is_stale():
        return out
    hipcc = hipcc_path()
    # a build with extra flags (diagnostic macros) never writes into the main object directory or over libvecsim.so: a later
    # plain build() would see those objects as fresh and ship the instrumented code
    if extra_flags and out == OUT:
        import hashlib

        out = os.path.join(HERE, "libvecsim_" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] + ".so")
    obj_dir = OBJ_DIR if out == OUT else out + ".build"
    os.makedirs(obj_dir, exist_ok=True)
    todo = []
    objs = []
    for src, obj, fl in units():
        obj = os.path.join(obj_dir, os.path.basename(obj))
        objs.append(obj)
        if force or extra_flags or _newer(obj, HEADERS + [src]):
            todo.append([hipcc, *FLAGS, *extra_flags, *fl, "-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
        return res.stderr

    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 2)))
    if todo:
        with ThreadPoolExecutor(jobs) as pool:
            for warn in pool.map(run, todo):
                if verbose and warn.strip():
                    print(warn, flush=True)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


if __name__ == "__main__":
    import sys
    import time

    t0 = time.time()
    print(build(force="--force" in sys.argv or len(sys.argv) == 1, verbose=True))
    print(f"{time.time() - t0:.1f} s")
