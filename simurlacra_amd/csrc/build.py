"""Build libvecsim.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "vecsim.hip")
DEPS = [SRC, os.path.join(HERE, "vecsim_envs.h"), os.path.join(HERE, "..", "..", "include", "vecsim.h")]
OUT = os.path.join(HERE, "libvecsim.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvecsim cannot be built (there is no CPU fallback)")


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall",
           # contraction only inside one source expression: every kernel variant (step / fused rollout / broadcast or
           # per-env constants) then rounds identically, which the bit-exactness tests rely on
           "-ffp-contract=on",
           # packed fp32 VALU ops (v_pk_fma_f32 ...) issue slower than the two scalar ops they replace on gfx950
           "-fno-slp-vectorize",
           "-Wno-unused-function", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
