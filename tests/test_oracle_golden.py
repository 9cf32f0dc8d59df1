"""
The oracle (oracle/cpu_ref.py) against what the reference itself produced (tests/golden/*.npz, made by
oracle/gen_golden.py) and against the reference's own known-answer tests.  CPU only.
"""
import json
import os

import numpy as np
import pytest

from oracle import cpu_ref

ENVS = ["omo", "bob", "qq-su", "qcp-su", "qbb", "qq-st", "qcp-st", "pend", "bob-d"]
KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
      "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500),
      "qq-st": dict(dt=0.01, max_steps=500), "qcp-st": dict(dt=0.01, max_steps=300),
      "pend": dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2])), "bob-d": dict(dt=0.01, max_steps=500)}
# fp64 restatement vs fp64 reference: op-for-op, only BLAS/LAPACK summation order may differ
RTOL, ATOL = 1e-11, 1e-13


def load(golden_dir, kind, name):
    return np.load(os.path.join(golden_dir, f"{kind}_{name.replace('-', '_')}.npz"))


@pytest.mark.parametrize("name", ENVS)
def test_single_step_cases(golden_dir, name):
    g = load(golden_dir, "step", name)
    ref = cpu_ref.make_ref(name, **KW[name])
    assert list(g["param_names"]) == list(ref.param_names)
    out = ref.step(g["state"], g["hidden"], g["act"], g["params"], g["curr_step"])
    np.testing.assert_allclose(out["state"], g["nstate"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["obs"], g["obs"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["rew"], g["rew"], rtol=1e-10, atol=1e-300)
    if ref.H:
        np.testing.assert_allclose(out["hidden"], g["nhidden"], rtol=1e-10, atol=1e-12)
    assert np.array_equal(out["done"], g["done"])  # bit-exact
    assert g["done"].any() and not g["done"].all()
    slo, shi, alo, ahi = ref.bounds(g["params"])
    np.testing.assert_allclose(slo, g["state_lo"], rtol=1e-15)
    np.testing.assert_allclose(shi, g["state_hi"], rtol=1e-15)
    np.testing.assert_allclose(ahi, g["act_hi"], rtol=1e-15)
    if ref.rew_kind == cpu_ref.REW_SCALED_EXP:
        np.testing.assert_allclose(ref.c_max(g["params"]), g["c_max"], rtol=1e-13)


@pytest.mark.parametrize("name", ENVS)
def test_trajectories(golden_dir, name):
    g = load(golden_dir, "traj", name)
    ref = cpu_ref.make_ref(name, **KW[name])
    n = g["params"].shape[0]
    r = ref.reset(g["params"], g["init"])
    np.testing.assert_allclose(r["obs"], g["reset_obs"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r["state"], g["state"][:, 0], rtol=RTOL, atol=ATOL)
    if ref.H:
        np.testing.assert_allclose(r["hidden"], g["hidden"][:, 0], rtol=0, atol=2e-6)  # fp32 IK (Q8)
    # step every env in lock-step from the reference's own previous state (one-step parity along the trajectory) ...
    saw_done = False
    for i in range(n):
        L = int(g["length"][i])
        yielded = np.zeros(1, dtype=bool)
        for t in range(L):
            out = ref.step(g["state"][i:i + 1, t], g["hidden"][i:i + 1, t], g["act"][i:i + 1, t], g["params"][i:i + 1],
                           np.array([t]), yielded=yielded)
            yielded = out["yielded"]
            np.testing.assert_allclose(out["state"][0], g["state"][i, t + 1], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(out["obs"][0], g["obs"][i, t], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(out["rew"][0], g["rew"][i, t], rtol=1e-10, atol=1e-300)
            assert bool(out["done"][0]) == bool(g["done"][i, t])
            saw_done |= bool(g["done"][i, t])
    if name in ("omo", "bob", "bob-d", "qcp-st"):
        assert saw_done
    # ... and free-running (closed loop in the oracle) over a short horizon
    state, hidden = r["state"], r["hidden"]
    hidden = g["hidden"][:, 0] if ref.H else hidden
    T = int(min(40, g["length"].min()))
    for t in range(T):
        out = ref.step(state, hidden, g["act"][:, t], g["params"], np.full(n, t))
        state, hidden = out["state"], out["hidden"]
    np.testing.assert_allclose(state, g["state"][:, T], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("name", ENVS)
def test_reset_cases(golden_dir, name):
    g = load(golden_dir, "reset", name)
    ref = cpu_ref.make_ref(name, **KW[name])
    for i in range(g["params"].shape[0]):
        full = bool(g["full"][i])
        init = g["init"][i][None, :] if full else g["init"][i][None, :ref.I]
        r = ref.reset(g["params"][i:i + 1], init, init_is_full_state=full)
        np.testing.assert_allclose(r["state"][0], g["state"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(r["obs"][0], g["obs"][i], rtol=RTOL, atol=ATOL)
        if ref.H:
            np.testing.assert_allclose(r["hidden"][0], g["hidden"][i], rtol=0, atol=2e-6)
    if name in ("bob", "bob-d"):
        lo0, hi0 = ref.init_bounds(g["params"], box=0)
        lo1, hi1 = ref.init_bounds(g["params"], box=1)
        np.testing.assert_allclose(np.concatenate([lo0, lo1], axis=1), g["init_lo"], rtol=1e-14)
        np.testing.assert_allclose(np.concatenate([hi0, hi1], axis=1), g["init_hi"], rtol=1e-14)
    else:
        lo, hi = ref.init_bounds(g["params"])
        np.testing.assert_allclose(lo, g["init_lo"], rtol=1e-14)
        np.testing.assert_allclose(hi, g["init_hi"], rtol=1e-14)


def test_cfg1_omo_500_steps(golden_dir):
    """BASELINE.json configs[0]: OneMassOscillatorSim, 1 env, 500 Euler steps, reference NumPy path"""
    g = np.load(os.path.join(golden_dir, "cfg1_omo_500.npz"))
    ref = cpu_ref.make_ref("omo", dt=0.02, max_steps=500)
    params = ref.nominal_params(1)
    r = ref.reset(params, np.array([[-0.7, 0.0]]))
    np.testing.assert_array_equal(r["obs"][0], g["obs0"])
    state, yielded = r["state"], np.zeros(1, dtype=bool)
    for t in range(500):
        out = ref.step(state, np.zeros((1, 0)), g["act"][t][None, :], params, np.array([t]), yielded=yielded)
        state, yielded = out["state"], out["yielded"]
        np.testing.assert_allclose(state[0], g["state"][t + 1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out["rew"][0], g["rew"][t], rtol=1e-9)
        assert bool(out["done"][0]) == bool(g["done"][t])


def test_qbb_ik(golden_dir):
    g = np.load(os.path.join(golden_dir, "qbb_ik.npz"))
    ang = cpu_ref.qbb_ik_fp32(g["th"], g["r"], g["l"])
    np.testing.assert_allclose(ang, g["ang"], rtol=0, atol=2e-6)


def test_survey_anchors():
    """SURVEY.md 8(c) sanity anchors (values printed from the reference)"""
    ref = cpu_ref.make_ref("qq-su", dt=0.004, max_steps=4000)
    p = ref.nominal_params(1)
    out = ref.step(np.array([[0.5, 3.0, -4.0, 40.0]]), np.zeros((1, 0)), np.array([[-7.0]]), p, np.array([0]))
    assert abs(out["rew"][0] - 0.001921930464825996) < 1e-15
    np.testing.assert_allclose(out["state"][0], [0.4841, 3.1599, -3.9524, 39.9743], atol=5e-5)
    ref = cpu_ref.make_ref("qcp-su", dt=0.002, max_steps=8000)
    out = ref.step(np.array([[0.05, 0.3, -0.2, 1.5]]), np.zeros((1, 1)), np.array([[3.0]]), ref.nominal_params(1),
                   np.array([0]))
    assert abs(out["rew"][0] - 0.017430662613738906) < 1e-15
    assert abs(out["hidden"][0, 0] - (-50.55938812802471)) < 1e-9
    ref = cpu_ref.make_ref("omo", dt=0.02, max_steps=3)
    out = ref.step(np.array([[0.99, 9.0]]), np.zeros((1, 0)), np.array([[30.0]]), ref.nominal_params(1), np.array([0]))
    assert out["done"][0] and abs(out["rew"][0] - (-1010.6119)) < 1e-3


def test_radial_fold_kat():
    """Pyrado/tests/test_tasks.py:82-99 (test_modulated_rew_fcn), Q = I4, R = I2, QuadrErrRewFcn"""
    s = np.array([[1.0, 2.0, 3.0, 4.0]])
    a = np.zeros((1, 2))
    Qd, Rd = np.ones(4), np.ones(2)
    err = cpu_ref.radial_fold(np.zeros((1, 4)) - s, [0, 1, 3], 2)
    assert -cpu_ref.weighted_quadr_cost(err, -a, Qd, Rd)[0] == -(1 ** 2 + 3 ** 2)
    err = cpu_ref.radial_fold(np.zeros((1, 4)) - s, [1, 3], np.array([2, 3]))
    assert -cpu_ref.weighted_quadr_cost(err, -a, Qd, Rd)[0] == -(1 ** 2 + 3 ** 2 + 1 ** 2)


SEED_KAT = [  # Pyrado/tests/test_set_seed.py:35-54
    (0, None, None, 813134492), (0, None, 0, 813134492), (0, None, 1, 4276188331), (0, 0, None, 813134492),
    (0, 0, 0, 813134492), (0, 0, 1, 4276188331), (0, 1, None, 229607210), (0, 1, 0, 229607210),
    (0, 1, 1, 3918913762), (1, None, None, 532102107), (1, None, 0, 532102107), (1, None, 1, 2754337450),
    (1, 0, None, 532102107), (1, 0, 0, 532102107), (1, 0, 1, 2754337450), (1, 1, None, 713485941),
    (1, 1, 0, 713485941), (1, 1, 1, 3511146676)]


@pytest.mark.parametrize("base,sub,subsub,expected", SEED_KAT)
def test_seed_kat(base, sub, subsub, expected):
    assert cpu_ref.derive_seed(base, sub, subsub) == expected


def test_seed_golden(golden_dir):
    for b, s, ss, exp in json.load(open(os.path.join(golden_dir, "set_seed.json"))):
        assert cpu_ref.derive_seed(b, s, ss) == exp


def test_nominal_params_match_reference(golden_dir):
    tab = json.load(open(os.path.join(golden_dir, "randomizers.json")))
    for name, cls in cpu_ref.ENV_REFS.items():
        nom = tab[name]["nominal"]
        assert list(nom.keys()) == sorted(cls.param_names)  # json sort_keys
        # get_nominal_domain_param() is a classmethod with long=False as default, also for QCartPoleStabSim
        nominal = cls.nominal_params(1, long=False)[0] if name == "qcp-st" else cls.nominal_params(1)[0]
        for k, v in zip(cls.param_names, nominal):
            assert nom[k] == pytest.approx(v, rel=1e-15, abs=0)


def test_act_norm_wrapper_golden(golden_dir):
    """the oracle's act_norm flag against steps of the reference's ActNormWrapper (action_normalization.py:63-89)"""
    g = np.load(os.path.join(golden_dir, "wrappers.npz"))
    for name in ("qq-su", "qbb"):
        tag = name.replace("-", "_")
        ref = cpu_ref.make_ref(name, **KW[name], act_norm=True)
        n = g[f"{tag}_state"].shape[0]
        out = ref.step(g[f"{tag}_state"], np.zeros((n, ref.H)), g[f"{tag}_act"], ref.nominal_params(n), np.zeros(n, dtype=int))
        np.testing.assert_allclose(out["state"], g[f"{tag}_nstate"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(out["rew"], g[f"{tag}_rew"], rtol=1e-10)
        assert np.array_equal(out["done"], g[f"{tag}_done"])


VARIANTS = {
    "qcp_su_simple": ("qcp-su", dict(dt=0.002, max_steps=8000, simple_dynamics=True)),
    "qcp_su_long": ("qcp-su", dict(dt=0.002, max_steps=8000, long=True)),
    "qcp_su_tame_init": ("qcp-su", dict(dt=0.002, max_steps=8000, wild_init="False")),
    "qbb_simple": ("qbb", dict(dt=0.01, max_steps=500, simple_dynamics=True)),
    "qq_su_task_args": ("qq-su", dict(dt=0.004, max_steps=4000, task_args=dict(
        state_des=np.array([0.3, -np.pi, 0.5, -0.2]), Q=np.diag([2.0, 0.5, 1e-2, 1e-3]), R=np.diag([1e-2])))),
    "bob_task_args": ("bob", dict(dt=0.01, max_steps=500, task_args=dict(
        state_des=np.array([0.2, 0.0, 0.0, 0.0]), Q=np.diag([1e4, 1e2, 1e2, 1e1]), R=np.diag([0.5])))),
    "omo_inf_steps": ("omo", dict(dt=0.02, max_steps=float("inf"))),
    "qcp_st_short_pole": ("qcp-st", dict(dt=0.01, max_steps=300, long=False, simple_dynamics=False)),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_ctor_and_task_variants(golden_dir, tag):
    """constructor / task options of the reference classes (simple_dynamics, long, wild_init, task_args, max_steps=inf)"""
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    name, kw = VARIANTS[tag]
    ref = cpu_ref.make_ref(name, **kw)
    n = g[f"{tag}__state"].shape[0]
    P = np.tile(g[f"{tag}__params"], (n, 1))
    out = ref.step(g[f"{tag}__state"], g[f"{tag}__hidden"], g[f"{tag}__act"], P, g[f"{tag}__curr_step"])
    np.testing.assert_allclose(out["state"], g[f"{tag}__nstate"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["obs"], g[f"{tag}__obs"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["rew"], g[f"{tag}__rew"], rtol=1e-10, atol=1e-300)
    assert np.array_equal(out["done"], g[f"{tag}__done"])
    if ref.H:
        np.testing.assert_allclose(out["hidden"], g[f"{tag}__nhidden"], rtol=1e-10, atol=1e-12)
    if f"{tag}__init_lo" in g.files:
        lo, hi = ref.init_bounds(P[:1])
        np.testing.assert_allclose(lo[0], g[f"{tag}__init_lo"], rtol=1e-14)
        np.testing.assert_allclose(hi[0], g[f"{tag}__init_hi"], rtol=1e-14)


def test_oracle_derivatives_match_the_forks_autograd(golden_dir):
    """central finite differences of the oracle's step (fp64) against the Jacobians the fork computes with torch.autograd
    through its differentiable QCartPole step (tests/golden/jac_qcp_su.npz, oracle/gen_golden.py:gen_jacobians): pins the
    derivative structure of the restatement -- RK4 with the th_ddot chain, clip and dead zone (zero action gradient), the
    reward on the unclipped action -- to the reference, not only its values"""
    g = np.load(os.path.join(golden_dir, "jac_qcp_su.npz"))
    ref = cpu_ref.make_ref("qcp-su", float(g["dt"]), int(g["max_steps"]))
    n, S, A = g["state"].shape[0], 4, 1
    P, hidden, curr = g["params"], g["hidden"], np.full(n, 10)
    x0 = np.concatenate([g["state"], g["act"]], axis=1)
    out0 = ref.step(g["state"], hidden, g["act"], P, curr)
    np.testing.assert_allclose(out0["state"], g["nstate"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(out0["rew"], g["rew"], rtol=1e-10)
    np.testing.assert_allclose(out0["hidden"], g["nhidden"], rtol=1e-10, atol=1e-12)
    js = np.zeros((n, S, S + A))
    jr = np.zeros((n, S + A))
    h = 1e-6
    for k in range(S + A):
        outs = []
        for sgn in (+1, -1):
            x = x0.copy()
            x[:, k] += sgn * h
            outs.append(ref.step(x[:, :S], hidden, x[:, S:], P, curr))
        js[:, :, k] = (outs[0]["state"] - outs[1]["state"]) / (2 * h)
        jr[:, k] = (outs[0]["rew"] - outs[1]["rew"]) / (2 * h)
    # kinks within h of the evaluation point (clip edge, dead-zone edge, sign of the cart velocity) spoil a finite difference
    tn, tp = (list(g["param_names"]).index(k) for k in ("voltage_thold_neg", "voltage_thold_pos"))
    a = g["act"][:, 0]
    near = (np.abs(np.abs(a) - 6.0) < 1e-4) | (np.abs(a - P[:, tn]) < 1e-4) | (np.abs(a - P[:, tp]) < 1e-4) | \
           (np.abs(g["state"][:, 2]) < 1e-4)
    assert near.sum() <= 2
    ok = ~near
    scale = np.abs(g["jac_state"]).max(axis=2, keepdims=True)
    assert (np.abs(js - g["jac_state"]) / scale)[ok].max() < 1e-7
    # the fork keeps state_des / Q / R in float32 (pi rounded to 24 bits): its reward gradient carries that rounding
    assert (np.abs(jr - g["jac_rew"]) / np.abs(g["jac_rew"]).max(axis=1, keepdims=True))[ok].max() < 2e-6
