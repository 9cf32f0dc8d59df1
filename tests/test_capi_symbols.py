"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/vecsim.h declares, its
static tables match the reference, and compute entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from simurlacra_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from simurlacra_amd.csrc import build

    build.build()
    return L.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vecsim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    decl = declared_functions()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/vecsim.h but not exported by libvecsim.so"
    assert sorted(L.exported_symbols()) == decl  # the ctypes table binds exactly the header


def test_static_tables_match_reference(lib, golden_dir):
    tab = json.load(open(os.path.join(golden_dir, "randomizers.json")))
    from simurlacra_amd import env_dims, nominal_params, param_names

    expect = {"omo": (2, 1, 2, 3, 0, 2), "bob": (4, 1, 4, 8, 0, 4), "qq-su": (4, 1, 6, 11, 0, 4),
              "qcp-su": (4, 1, 5, 17, 1, 4), "qbb": (8, 2, 8, 20, 2, 4), "qq-st": (4, 1, 6, 11, 0, 4),
              "qcp-st": (4, 1, 5, 17, 1, 4), "pend": (2, 1, 3, 5, 0, 2), "bob-d": (4, 1, 4, 8, 0, 4)}
    for name, t in L.ENV_TYPES.items():
        assert lib.vs_env_name(t).decode() == name
        d = env_dims(name)
        assert (d["S"], d["A"], d["O"], d["P"], d["H"], d["I"]) == expect[name]
        names = param_names(name)
        assert sorted(names) == sorted(tab[name]["nominal"])
        nom = nominal_params(name)
        for k, v in zip(names, nom):
            assert v == np.float32(tab[name]["nominal"][k])
    assert lib.vs_env_name(99) is None and lib.vs_param_name(0, 99) is None
    long = nominal_params("qcp-su", long=True)
    assert long[12] == np.float32(0.23) and long[13] == np.float32(0.641 / 2)


def test_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.vs_create(2, 16, 0.004, 4000, 0, None, C.byref(h))
    assert rc == L.VS_ERR_HIP and not h.value
    assert b"no HIP device" in lib.vs_last_error(None)
    from simurlacra_amd import VecSimEnv

    with pytest.raises(RuntimeError, match="no HIP device"):
        VecSimEnv("qq-su", 16, 0.004, 4000)


def test_bad_arguments(lib):
    h = C.c_void_p()
    assert lib.vs_create(77, 16, 0.004, 4000, 0, None, C.byref(h)) == L.VS_ERR_ARG
    assert lib.vs_create(2, 0, 0.004, 4000, 0, None, C.byref(h)) == L.VS_ERR_ARG
    assert lib.vs_create(2, 16, -1.0, 4000, 0, None, C.byref(h)) == L.VS_ERR_ARG
    assert lib.vs_env_dims(-1, None, None, None, None, None, None, None) == L.VS_ERR_ARG


def test_plain_c_host_program_links_and_runs(lib, tmp_path):
    """include/vecsim.h from C: compile tests/capi/capi_demo.c with gcc against libvecsim.so and run it (static tables
    everywhere; on a GPU box also a fused rollout, here the loud no-GPU failure)"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "capi_demo")
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "capi", "capi_demo.c"), "-o", exe, "-L", libdir, "-l:libvecsim.so",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "static tables ok" in out.stdout
    assert ("no GPU: vs_create failed loudly" in out.stdout) or ("fused steps ok" in out.stdout)
