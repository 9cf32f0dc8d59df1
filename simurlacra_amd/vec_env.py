"""
Batched, device-resident environment: the Python face of one libvecsim handle (include/vecsim.h).

``VecSimEnv`` is the vectorised counterpart of Pyrado's ``SimPyEnv`` (P/environments/pysim/base.py:43-289): N independent
environment instances of one family, one per wavefront lane on an MI355X.  Control-path data (domain parameters, explicit
initial states, masks) are NumPy arrays; hot-path data (actions in, observations / rewards / done flags out) are
device buffers exposed zero-copy as torch tensors.  There is no CPU implementation behind this class.
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L
from .exceptions import ShapeErr, TypeErr, ValueErr

_BUF_DTYPES = {
    L.VS_STATE: ("f4", "S"), L.VS_OBS: ("f4", "O"), L.VS_REW: ("f4", 1), L.VS_DONE: ("u1", 1),
    L.VS_HIDDEN: ("f4", "H"), L.VS_STEPCOUNT: ("i4", 1), L.VS_ERRFLAG: ("u1", 1), L.VS_RETURNS: ("f4", 1),
    L.VS_PARAMS: ("f4", "P"), L.VS_CONSTS: ("f4", "K"), L.VS_FAILED: ("u1", 1),
    L.VS_EPSTAT_COUNT: ("u4", 1), L.VS_EPSTAT_RETSUM: ("f4", 1), L.VS_EPSTAT_LENSUM: ("i4", 1),
}


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch.as_tensor() can wrap a libvecsim buffer without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2,
                                             strides=None)
        self._owner = owner  # keep the handle alive


def env_dims(name):
    lib = L.load()
    vals = [C.c_int() for _ in range(7)]
    rc = lib.vs_env_dims(L.ENV_TYPES[name], *[C.byref(v) for v in vals])
    if rc != 0:
        raise ValueErr(msg=f"unknown env type {name}")
    return dict(zip("SAOPHIK", (v.value for v in vals)))


def param_names(name):
    lib = L.load()
    t = L.ENV_TYPES[name]
    return [lib.vs_param_name(t, i).decode() for i in range(env_dims(name)["P"])]


def nominal_params(name, long=False):
    lib = L.load()
    d = env_dims(name)
    out = (C.c_float * d["P"])()
    lib.vs_nominal_params(L.ENV_TYPES[name], L.VS_FLAG_LONG_POLE if long else 0, out)
    return np.array(out[:], dtype=np.float32)


class VecSimEnv:
    """N environments of one Pyrado pysim family on one GPU."""

    def __init__(self, name, n_envs, dt, max_steps=math.inf, task_args=None, device=0, simple_dynamics=None,
                 long=None, wild_init="True", init_state=None):
        if name not in L.ENV_TYPES:
            raise ValueErr(msg=f"unknown environment name {name!r}; expected one of {sorted(L.ENV_TYPES)}")
        if not isinstance(dt, (int, float)):
            raise TypeErr(given=dt, expected_type=(int, float))  # Env.__init__, P/environments/base.py:56-57
        if dt < 0:
            raise ValueErr(given=dt, ge_constraint="0")
        if max_steps < 1:
            raise ValueErr(given=max_steps, ge_constraint="1")
        if not (isinstance(task_args, dict) or task_args is None):
            raise TypeErr(given=task_args, expected_type=dict)  # P/environments/pysim/base.py:70-71
        # ctor defaults of the reference classes: QCartPoleStabSim(long=True, simple_dynamics=True)
        # (quanser_cartpole.py:452-459), everything else False
        if simple_dynamics is None:
            simple_dynamics = name == "qcp-st"
        if long is None:
            long = name == "qcp-st"
        self._lib = L.load()
        self.name = name
        self.n_envs = int(n_envs)
        self.dt = float(dt)
        self.max_steps = max_steps
        self.device = int(device)
        self.dims = env_dims(name)
        self.param_names = param_names(name)
        self._flags = (L.VS_FLAG_SIMPLE_DYNAMICS if simple_dynamics else 0) | (L.VS_FLAG_LONG_POLE if long else 0)
        cfg = L.TaskCfg()
        cfg.use_defaults = 1
        cfg.flags = self._flags
        cfg.wild_init = {"True": 0, "False": 1}.get(str(wild_init), 2)
        if init_state is not None:  # PendulumSim(init_state=...): the fixed state of its SingularStateSpace
            ist = np.asarray(init_state, dtype=np.float64).reshape(-1)
            if ist.size != self.dims["S"]:
                raise ShapeErr(given=ist, expected_match=(self.dims["S"],))
            cfg.init_state[: ist.size] = list(map(float, ist))
        if task_args:
            cfg.use_defaults = 0
            des, qd, rd = self.default_task(name)
            if "state_des" in task_args:
                des = np.asarray(task_args["state_des"], dtype=np.float64).reshape(-1)
            if "Q" in task_args:
                qd = self._diag(task_args["Q"], "Q")
            if "R" in task_args:
                rd = self._diag(task_args["R"], "R")
            if des.shape != (self.dims["S"],) or qd.shape != (self.dims["S"],) or rd.shape != (self.dims["A"],):
                raise ShapeErr(msg="task_args state_des / Q / R do not match the state / action dimensions")
            cfg.state_des[: des.size] = list(map(float, des))
            cfg.q_diag[: qd.size] = list(map(float, qd))
            cfg.r_diag[: rd.size] = list(map(float, rd))
        self._cfg = cfg
        h = C.c_void_p()
        ms = 0 if max_steps == math.inf else int(max_steps)
        rc = self._lib.vs_create(L.ENV_TYPES[name], self.n_envs, self.dt, ms, self.device, C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"vs_create failed ({rc}): {self._lib.vs_last_error(None).decode()}")
        self._h = h
        self.ld = int(self._lib.vs_ld(h))
        self._traj_cap = 0

    # ------------------------------------------------------------------------------------------------ helpers
    @staticmethod
    def default_task(name):
        """Reference defaults of _create_task (state_des, diag Q, diag R) for each env."""
        pi = np.pi
        return {
            "omo": (np.zeros(2), np.array([1e1, 1e-2]), np.array([1e-6])),
            "bob": (np.zeros(4), np.array([1e5, 1e3, 1e3, 1e2]), np.array([1.0])),
            "qq-su": (np.array([0.0, pi, 0.0, 0.0]), np.array([1.0, 1.0, 2e-2, 5e-3]), np.array([4e-3])),
            "qcp-su": (np.array([0.0, pi, 0.0, 0.0]), np.array([3e-1, 5e-1, 5e-3, 1e-3]), np.array([1e-3])),
            "qbb": (np.zeros(8), np.array([1e0, 1e0, 5e3, 5e3, 1e-2, 1e-2, 5e-1, 5e-1]), np.array([1e-2, 1e-2])),
            "qq-st": (np.array([0.0, pi, 0.0, 0.0]), np.array([3.0, 4.0, 2.0, 2.0]), np.array([5e-2])),
            "qcp-st": (np.array([0.0, pi, 0.0, 0.0]), np.array([5e-0, 1e1, 1e-2, 1e-2]), np.array([1e-3])),
            "pend": (np.array([pi, 0.0]), np.array([1e-0, 1e-3]), np.array([1e-2])),
            "bob-d": (np.zeros(4), np.array([1e5, 1e3, 1e3, 1e2]), np.array([1.0])),
        }[name]

    @staticmethod
    def _diag(M, label):
        M = np.asarray(M, dtype=np.float64)
        if M.ndim == 1:
            return M
        if np.count_nonzero(M - np.diag(np.diag(M))) != 0:
            raise ValueErr(msg=f"The weight matrix {label} must be diagonal for the device kernels")
        return np.diag(M).copy()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.vs_last_error(self._h).decode()
            if rc == L.VS_ERR_ARG:
                raise ValueErr(msg=f"{what}: {msg}")
            raise RuntimeError(f"{what} failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mask_arg(self, mask):
        if mask is None:
            return None, None
        m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if m.shape != (self.n_envs,):
            raise ShapeErr(given=m, expected_match=(self.n_envs,))
        return m, m.ctypes.data_as(C.c_void_p)

    # ------------------------------------------------------------------------------------------------ domain parameters
    def _params_matrix(self, params):
        """dict name -> scalar/array [N]  or array [N, P]  ->  f32 [P][N] (unspecified names keep the current value)"""
        P = self.dims["P"]
        if isinstance(params, dict):
            unknown = [k for k in params if k not in self.param_names]
            if unknown:
                raise ValueErr(msg=f"unsupported domain parameter(s) {unknown} for env {self.name}")
            cur = self.get(L.VS_PARAMS).T.copy()  # [P][N]
            for k, v in params.items():
                cur[self.param_names.index(k)] = np.broadcast_to(np.asarray(v, dtype=np.float32).reshape(-1), (self.n_envs,)) \
                    if np.size(v) in (1, self.n_envs) else self._bad_shape(v)
            return np.ascontiguousarray(cur, dtype=np.float32)
        arr = np.asarray(params, dtype=np.float32)
        if arr.shape != (self.n_envs, P):
            raise ShapeErr(given=arr, expected_match=(self.n_envs, P))
        return np.ascontiguousarray(arr.T)

    def _bad_shape(self, v):
        raise ShapeErr(given=np.asarray(v), expected_match=(self.n_envs,))

    def set_params(self, params, mask=None):
        """The `domain_param` setter for all (or the masked) envs: params -> _calc_constants -> spaces -> task.reset."""
        mat = self._params_matrix(params)
        m, mp = self._mask_arg(mask)
        self._check(self._lib.vs_set_params(self._h, mat.ctypes.data_as(C.c_void_p), self.n_envs, mp), "vs_set_params")

    def set_params_uniform(self, params=None):
        vec = nominal_params(self.name, long=bool(self._flags & L.VS_FLAG_LONG_POLE))
        if params:
            for k, v in params.items():
                if k not in self.param_names:
                    raise ValueErr(msg=f"unsupported domain parameter {k!r} for env {self.name}")
                vec[self.param_names.index(k)] = float(v)
        arr = (C.c_float * len(vec))(*map(float, vec))
        self._check(self._lib.vs_set_params_uniform(self._h, arr), "vs_set_params_uniform")

    def _specs(self, specs):
        """specs: iterable of (name, kind, mean, spread, clip_lo, clip_up[, aux, roundint]) with kind 'normal' (spread =
        std), 'uniform' (spread = halfspan) or 'bernoulli' (mean = val_0, spread = val_1, aux = prob_1), or DomainParam
        objects (DomainRandomizer.device_specs() does the conversion)"""
        kinds = {"normal": L.VS_DP_NORMAL, "uniform": L.VS_DP_UNIFORM, "bernoulli": L.VS_DP_BERNOULLI}
        rows = []
        for s in specs:
            if not isinstance(s, (tuple, list)):
                from .domain_randomization import DomainRandomizer

                s = DomainRandomizer(s).device_specs()[0]
            name, kind, mean, spread, lo, hi = s[:6]
            aux = float(s[6]) if len(s) > 6 else 0.0
            rnd = int(bool(s[7])) if len(s) > 7 else 0
            if name not in self.param_names:
                raise ValueErr(msg=f"unsupported domain parameter {name!r} for env {self.name}")
            if kind not in kinds:
                raise ValueErr(given=kind, eq_constraint="normal, uniform or bernoulli")
            if kind == "bernoulli" and not 0.0 <= aux <= 1.0:
                raise ValueErr(given=aux, ge_constraint="0", le_constraint="1")
            rows.append(L.DpSpec(self.param_names.index(name), kinds[kind], float(mean), float(spread), float(lo),
                                 float(hi), aux, rnd))
        arr = (L.DpSpec * max(len(rows), 1))(*rows)
        return arr, len(rows)

    def sample_params(self, specs, seed=0, mask=None):
        arr, n = self._specs(specs)
        m, mp = self._mask_arg(mask)
        self._check(self._lib.vs_sample_params(self._h, arr, n, int(seed) & (2 ** 64 - 1), mp), "vs_sample_params")

    def set_randomizer(self, specs):
        arr, n = self._specs(specs or [])
        self._check(self._lib.vs_set_randomizer(self._h, arr, n), "vs_set_randomizer")

    def set_param_buffer(self, param_sets, selection="cyclic"):
        """DomainRandWrapperBuffer on the device. param_sets: list of dicts (missing names keep the nominal value) or
        array [B, P]; None / empty removes the buffer."""
        if selection not in ("cyclic", "random"):
            raise ValueErr(given=selection, eq_constraint="cyclic or random")
        if param_sets is None or len(param_sets) == 0:
            self._check(self._lib.vs_set_param_buffer(self._h, None, 0, 0), "vs_set_param_buffer")
            return
        if isinstance(param_sets[0], dict):
            base = nominal_params(self.name, long=bool(self._flags & L.VS_FLAG_LONG_POLE))
            mat = np.tile(base, (len(param_sets), 1))
            for b, dct in enumerate(param_sets):
                for k, v in dct.items():
                    if k not in self.param_names:
                        raise ValueErr(msg=f"unsupported domain parameter {k!r} for env {self.name}")
                    mat[b, self.param_names.index(k)] = float(np.asarray(v).reshape(-1)[0])
        else:
            mat = np.asarray(param_sets, dtype=np.float32)
            if mat.ndim != 2 or mat.shape[1] != self.dims["P"]:
                raise ShapeErr(given=mat, expected_match=(len(param_sets), self.dims["P"]))
        soa = np.ascontiguousarray(mat.T, dtype=np.float32)
        self._check(self._lib.vs_set_param_buffer(self._h, soa.ctypes.data_as(C.c_void_p), soa.shape[1],
                                                  0 if selection == "cyclic" else 1), "vs_set_param_buffer")

    def set_max_steps(self, max_steps):
        """env.max_steps = ... : the running episodes and their step counters are untouched (vs_set_max_steps)"""
        if max_steps < 1:
            raise ValueErr(given=max_steps, ge_constraint="1")
        self._check(self._lib.vs_set_max_steps(self._h, 0 if max_steps == math.inf else int(max_steps)), "vs_set_max_steps")
        self.max_steps = max_steps

    def set_dt(self, dt):
        self._check(self._lib.vs_set_dt(self._h, float(dt)), "vs_set_dt")
        self.dt = float(dt)

    def set_act_norm(self, on=True):
        """Fuse ActNormWrapper into the kernels: incoming actions are in [-1, 1]."""
        self._check(self._lib.vs_set_act_norm(self._h, int(bool(on))), "vs_set_act_norm")

    def _fvec(self, x, width, what):
        if x is None:
            return None
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float32), (width,)))
        if not np.isfinite(arr).all():
            raise ValueErr(msg=f"{what} must be finite")
        return arr.ctypes.data_as(C.POINTER(C.c_float)), arr  # keep the array alive

    def set_act_pipeline(self, delay=0, noise_mean=None, noise_std=None, noise_normed=False, noise_after_delay=False,
                         seed=0):
        """Fuse GaussianActNoiseWrapper / ActDelayWrapper into the kernels (vs_set_act_pipeline).  The policy's action is
        de-normalised (ActNormWrapper, if fused), then  [+ noise] -> delay queue -> [+ noise]  -> env.step."""
        delay = int(round(delay))  # ActDelayWrapper.delay rounds (action_delay.py:57-62)
        if not 0 <= delay <= L.VS_MAX_ACT_DELAY:
            raise ValueErr(given=delay, ge_constraint="0", le_constraint=str(L.VS_MAX_ACT_DELAY))
        A = self.dims["A"]
        m, s = self._fvec(noise_mean, A, "noise_mean"), self._fvec(noise_std, A, "noise_std")
        self._check(self._lib.vs_set_act_pipeline(self._h, delay, m[0] if m else None, s[0] if s else None,
                                                  int(bool(noise_normed)), int(bool(noise_after_delay)),
                                                  C.c_uint64(seed & (2 ** 64 - 1))), "vs_set_act_pipeline")

    def set_obs_pipeline(self, scale=None, shift=None, noise_std=None, seed=0):
        """Fuse ObsNormWrapper / GaussianObsNoiseWrapper into the kernels: obs' = obs * scale + shift + noise_std * z
        (vs_set_obs_pipeline).  `compose_obs_stages` turns a wrapper stack into these three vectors."""
        O = self.dims["O"]
        a, b, s = self._fvec(scale, O, "scale"), self._fvec(shift, O, "shift"), self._fvec(noise_std, O, "noise_std")
        self._check(self._lib.vs_set_obs_pipeline(self._h, a[0] if a else None, b[0] if b else None, s[0] if s else None,
                                                  C.c_uint64(seed & (2 ** 64 - 1))), "vs_set_obs_pipeline")

    # ------------------------------------------------------------------------------------------------ reset / step
    def reset(self, init_state=None, mask=None, seed=0):
        """SimPyEnv.reset for all (masked) envs. init_state: None (sample init space) or [N, I] / [N, S]."""
        m, mp = self._mask_arg(mask)
        ptr, full = None, 0
        keep = None
        if init_state is not None:
            arr = np.asarray(init_state, dtype=np.float32)
            if arr.ndim != 2 or arr.shape[0] != self.n_envs or arr.shape[1] not in (self.dims["I"], self.dims["S"]):
                raise ShapeErr(given=arr, expected_match=(self.n_envs, self.dims["I"]))
            full = int(arr.shape[1] == self.dims["S"])
            keep = np.ascontiguousarray(arr.T)
            ptr = keep.ctypes.data_as(C.c_void_p)
        self._check(self._lib.vs_reset(self._h, ptr, self.n_envs, full, mp, int(seed) & (2 ** 64 - 1)), "vs_reset")

    def set_index_offset(self, first_global_index=0):
        """Global index of lane 0: random streams are keyed by (offset + lane), see include/vecsim.h."""
        self._check(self._lib.vs_set_index_offset(self._h, int(first_global_index) & 0xFFFFFFFF), "vs_set_index_offset")

    def set_auto_reset(self, on=True, seed=0):
        self._check(self._lib.vs_set_auto_reset(self._h, int(bool(on)), int(seed) & (2 ** 64 - 1)), "vs_set_auto_reset")

    def step(self, actions):
        """One SimPyEnv.step for every env. `actions`: torch CUDA tensor of shape [N, A] (row-major policy output),
        [A, N] / [A, ld] (struct-of-arrays) or [N] when A == 1."""
        es, ds = self._act_strides(actions)
        self._check(self._lib.vs_step(self._h, C.c_void_p(actions.data_ptr()), es, ds), "vs_step")

    def step_record(self, actions, row=None):
        """step() that also writes the step's record (the observation the policy saw, its action, reward, done bit; in
        record mode 2 also state / applied action / hidden state) into row `row` of the trajectory buffers; row=None uses
        and advances the handle's device-side row counter (set_record_row), which is what a captured graph needs"""
        es, ds = self._act_strides(actions)
        self._check(self._lib.vs_step_record(self._h, C.c_void_p(actions.data_ptr()), es, ds, -1 if row is None else int(row)),
                    "vs_step_record")

    def set_record_row(self, row=0):
        self._check(self._lib.vs_set_record_row(self._h, int(row)), "vs_set_record_row")

    def seek_random(self, step_index=0):
        """Reposition the action stream of step_random (absolute step index, see include/vecsim.h)."""
        self._check(self._lib.vs_seek_random(self._h, int(step_index)), "vs_seek_random")

    def _act_strides(self, actions):
        A = self.dims["A"]
        if not hasattr(actions, "data_ptr"):
            raise TypeErr(given=actions, expected_type="torch.Tensor (device)")
        if not actions.is_cuda or str(actions.dtype) != "torch.float32":
            raise TypeErr(msg="actions must be a float32 tensor on the GPU")
        shp = tuple(actions.shape)
        if shp == (self.n_envs, A) or (A == 1 and shp == (self.n_envs,)):
            return (actions.stride(0), actions.stride(1)) if actions.dim() == 2 else (actions.stride(0), 0)
        if len(shp) == 2 and shp[0] == A and shp[1] in (self.n_envs, self.ld):
            return actions.stride(1), actions.stride(0)
        raise ShapeErr(given=actions, expected_match=(self.n_envs, A))

    def step_jac(self, actions):
        """vs_step plus the step Jacobians; returns dict(state=[N,S,S+A], rew=[N,S+A], obs=[N,O,S+A]) (host copies)."""
        es, ds = self._act_strides(actions)
        self._check(self._lib.vs_step_jac(self._h, C.c_void_p(actions.data_ptr()), es, ds), "vs_step_jac")
        S, A, O = self.dims["S"], self.dims["A"], self.dims["O"]
        out = {}
        for key, which, rows in (("state", L.VS_JAC_STATE, S), ("rew", L.VS_JAC_REW, 1), ("obs", L.VS_JAC_OBS, O)):
            buf = np.empty((rows * (S + A), self.ld), dtype=np.float32)
            self._check(self._lib.vs_copy_to_host(self._h, which, buf.ctypes.data_as(C.c_void_p)), "vs_copy_to_host")
            arr = buf[:, : self.n_envs].reshape(rows, S + A, self.n_envs).transpose(2, 0, 1)
            out[key] = np.ascontiguousarray(arr[:, 0] if key == "rew" else arr)
        return out

    def set_traj_capacity(self, t_max):
        if t_max > self._traj_cap:
            self._check(self._lib.vs_set_traj_capacity(self._h, int(t_max)), "vs_set_traj_capacity")
            self._traj_cap = int(t_max)

    def set_record_mode(self, mode=1):
        """What a recording step_random keeps per step: 1 = [obs | act | rew], 2 = + [state | act_app | hidden] -- the
        fields rollout() returns (P/sampling/rollout.py:305-325).  A change drops the record buffers."""
        if int(mode) != self.record_mode:
            self._check(self._lib.vs_set_record_mode(self._h, int(mode)), "vs_set_record_mode")
            self._traj_cap = 0
            self._traj_t0 = 0

    @property
    def record_mode(self):
        return int(self._lib.vs_record_mode(self._h))

    def set_freeze_done(self, on=True):
        """step() leaves lanes alone whose done flag is set (rollout() stops at done); off: env.step() keeps stepping"""
        self._check(self._lib.vs_set_freeze_done(self._h, int(bool(on))), "vs_set_freeze_done")

    def set_lean_step(self, on=True):
        """step() keeps what SimPyEnv.step returns -- (obs, rew, done) -- and neither the running return VS_RETURNS nor the
        VS_FAILED byte: exactly the 117 algorithmic bytes per QQube env step of SURVEY.md 8(d) (default: both are kept)"""
        self._check(self._lib.vs_set_lean_step(self._h, int(bool(on))), "vs_set_lean_step")

    def set_traj_offset(self, t0=0):
        """first record row of the next recording step_random (consecutive launches fill one long buffer)"""
        self._check(self._lib.vs_set_traj_offset(self._h, int(t0)), "vs_set_traj_offset")
        self._traj_t0 = int(t0)

    _VARIANTS = {None: -1, "k_rollout": 0, "k_rollout_ws": 1, "k_rollout_ws64": 2, "k_rollout_ws64g": 3, "k_rollout_ws256g": 4}

    def set_rollout_variant(self, variant=None):
        """None: automatic; 'k_rollout' / 'k_rollout_ws' (256-env workgroups) / 'k_rollout_ws64' (64-env workgroups) pin the
        fused kernel (bit-identical results every way)"""
        self._check(self._lib.vs_set_rollout_variant(self._h, self._VARIANTS[variant]), "vs_set_rollout_variant")

    _POLICY_SHAPES = {None: -1, "64": 0, "256": 1, "mfma": 2}

    def set_policy_shape(self, shape=None):
        """None: automatic; '64' / '256': the in-kernel policy network on the vector ALU in 64- / 256-env workgroups; 'mfma': 256-env
        workgroups with the hidden layers on the matrix cores (one and two hidden layers).  Same network, another summation order."""
        self._check(self._lib.vs_set_policy_shape(self._h, self._POLICY_SHAPES[shape]), "vs_set_policy_shape")

    def rollout_variant(self):
        """the kernel vs_step_random launches for the current configuration"""
        return {0: "k_rollout", 1: "k_rollout_ws", 2: "k_rollout_ws64", 3: "k_rollout_ws64g", 4: "k_rollout_ws256g"}[self._lib.vs_rollout_variant(self._h)]

    def traj_layout(self, mode=None):
        """(F, nq, h2, h1): a record is F floats -- mode 1: [obs | act | rew], mode 2: + [state | act_app | hidden] --
        stored as nq planes of 4, h2 of 2 and h1 of 1 floats per env (include/vecsim.h, VS_TRAJ_REC)"""
        v = [C.c_int() for _ in range(4)]
        mode = self.record_mode if mode is None else int(mode)
        self._check(self._lib.vs_traj_layout(L.ENV_TYPES[self.name], mode, *[C.byref(x) for x in v]), "vs_traj_layout")
        return tuple(x.value for x in v)

    def record_fields(self, mode=None):
        """{field: (first column, width)} of a record in the given mode"""
        S, A, O, H = (self.dims[k] for k in "SAOH")
        f = {"obs": (0, O), "act": (O, A), "rew": (O + A, 1)}
        if (self.record_mode if mode is None else int(mode)) == 2:
            b = O + A + 1
            f.update(state=(b, S), act_app=(b + S, A), hidden=(b + S + A, H))
        return f

    def traj_planes(self):
        """zero-copy views of the record buffers: [(plane [T_cap, ld, w], first record column)] and the done-flag words
        i32 [ceil(T_cap / 32), ld] (bit t % 32 of word [t // 32, i] = done flag of recorded step t of env i)"""
        import torch

        F, nq, h2, h1 = self.traj_layout()
        ld, dev = self.ld, f"cuda:{self.device}"
        ptr = self._lib.vs_get(self._h, L.VS_TRAJ_REC)
        rows = torch.as_tensor(_DevArray(ptr, (self._traj_cap, F * ld), "<f4", self), device=dev)
        planes, off, col = [], 0, 0
        for w, count in ((4, nq), (2, h2), (1, h1)):
            for _ in range(count):
                planes.append((rows[:, off:off + w * ld].view(self._traj_cap, ld, w), col))
                off += w * ld
                col += w
        dptr = self._lib.vs_get(self._h, L.VS_TRAJ_DONE)
        words = torch.as_tensor(_DevArray(dptr, ((self._traj_cap + 31) // 32, ld), "<i4", self), device=dev)
        return planes, words

    def traj_done(self, k_steps=None, n=None):
        """done flags of the recorded steps as a bool tensor [T, n] on the device (unpacked from the bit words)"""
        import torch

        T = self._traj_cap if k_steps is None else int(k_steps)
        n = self.n_envs if n is None else int(n)
        words = self.traj_planes()[1][: (T + 31) // 32, :n]
        shifts = torch.arange(32, device=words.device, dtype=torch.int32)
        bits = (words[:, None, :] >> shifts[None, :, None]) & 1  # [W, 32, n]
        return bits.reshape(-1, n)[:T].bool()

    def gather_traj(self, t_idx, lane_idx):
        """records [len(t_idx), F] of the (step, env) pairs given by two index tensors: reads only what is asked for"""
        import torch

        planes, _ = self.traj_planes()
        return torch.cat([p[t_idx, lane_idx] for p, _ in planes], dim=1)

    def rollout_lengths(self, n, t_steps):
        """vs_rollout_lengths: (lengths [n] int64, done_last [n] bool) on the device -- a rollout ends with the first recorded
        step whose done bit is set, or with the records"""
        import torch

        dev = f"cuda:{self.device}"
        lengths = torch.empty(n, dtype=torch.int64, device=dev)
        done_last = torch.empty(n, dtype=torch.uint8, device=dev)
        self._check(self._lib.vs_rollout_lengths(self._h, int(n), int(t_steps), C.c_void_p(lengths.data_ptr()),
                                                 C.c_void_p(done_last.data_ptr())), "vs_rollout_lengths")
        return lengths, done_last.bool()

    def pack_traj(self, n, t_steps, lengths, starts, total=None):
        """vs_pack_traj: rollout j = the first lengths[j] recorded steps of lane j (j < n), the rollouts one after the other in ONE
        matrix on the device (lengths / starts: int64 device tensors, starts the exclusive cumulative sum): `rows` [total + n, F],
        rollout j in rows starts[j] + j .. starts[j] + j + lengths[j] -- its steps and, last, the entry behind them (final
        observation / state / hidden state).  Returns dict(rows=..., obs, act, rew; record mode 2: state, act_app, hidden | None):
        the fields are strided VIEWS of `rows` ([total + n, width]; rew [total + n]), all indexed by the same rows."""
        import torch

        total = int(starts[-1] + lengths[-1]) if total is None else int(total)  # (a device sync unless the caller knows it)
        dev = lengths.device
        F = self.traj_layout()[0]
        rows = torch.empty(total + n, F, device=dev)
        lengths, starts = lengths.to(torch.int64).contiguous(), starts.to(torch.int64).contiguous()
        self._check(self._lib.vs_pack_traj(self._h, int(n), int(t_steps), C.c_void_p(lengths.data_ptr()), C.c_void_p(starts.data_ptr()),
                                           C.c_void_p(rows.data_ptr())), "vs_pack_traj")
        out = dict(rows=rows)
        for k, (c0, w) in self.record_fields().items():
            out[k] = None if w == 0 else (rows[:, c0] if k == "rew" else rows[:, c0:c0 + w])
        return out

    def traj_tensors(self, k_steps=None, n=None):
        """The recorded steps as torch tensors on the device: dict(obs [T, n, O], act [T, n, A], rew [T, n], done [T, n] u8;
        in record mode 2 also state [T, n, S], act_app [T, n, A], hidden [T, n, H]).  `rec` ([T, n, F], one gather of the
        record planes) is the only copy of the records, the fields are views of it; done is unpacked from the bit words.
        The caller orders its stream with the handle's (see vs_set_stream)."""
        import torch

        T = self._traj_cap if k_steps is None else int(k_steps)
        n = self.n_envs if n is None else int(n)
        planes, _ = self.traj_planes()
        rec = torch.cat([p[:T, :n] for p, _ in planes], dim=2)  # [T, n, F]
        out = dict(rec=rec, done=self.traj_done(T, n).to(torch.uint8))
        for k, (c0, w) in self.record_fields().items():
            out[k] = rec[..., c0] if k == "rew" else rec[..., c0:c0 + w]
        return out

    def step_random(self, k_steps=1, seed=0, record=False):
        if record and getattr(self, "_traj_t0", 0) + k_steps > self._traj_cap:
            self.set_traj_capacity(getattr(self, "_traj_t0", 0) + int(k_steps))
        self._check(self._lib.vs_step_random(self._h, int(seed) & (2 ** 64 - 1), int(k_steps), int(bool(record))),
                    "vs_step_random")

    # ------------------------------------------------------------------------------------------------ policy in the kernel
    _NONLIN = {None: L.VS_NL_NONE, "none": L.VS_NL_NONE, "tanh": L.VS_NL_TANH, "relu": L.VS_NL_RELU, "sigmoid": L.VS_NL_SIGMOID}

    def set_policy_fnn(self, params, hidden_sizes, hidden_nonlin="tanh", output_nonlin=None, feat=False, obs_idx=None,
                       noise_std=None):
        """Hand a feed-forward network policy (FNN of P/policies/feed_back/fnn.py:43-160) to the fused kernel of step_policy.
        params: the flat parameter vector in torch order (parameters_to_vector(net.parameters())), a torch tensor or array;
        hidden_nonlin: one name or one per hidden layer ('tanh' | 'relu' | 'sigmoid' | None); feat: the fork's FNNPolicy
        featurisation [o_0, sin o_1, cos o_1, o_2 ..]; obs_idx: the observation rows the policy sees (ObsPartialWrapper);
        noise_std: exploration noise per action dimension.  params=None removes the network."""
        if params is None:
            self._check(self._lib.vs_set_policy_fnn(self._h, None, None, 0), "vs_set_policy_fnn")
            return
        hs = [int(x) for x in hidden_sizes]
        if not 1 <= len(hs) <= L.VS_FNN_MAX_HIDDEN or max(hs) > L.VS_FNN_MAX_WIDTH:
            raise ValueErr(msg=f"the in-kernel policy takes 1..{L.VS_FNN_MAX_HIDDEN} hidden layers of at most "
                               f"{L.VS_FNN_MAX_WIDTH} units, got {hs}")
        nl = list(hidden_nonlin) if isinstance(hidden_nonlin, (list, tuple)) else [hidden_nonlin] * len(hs)
        d = L.FnnDesc()
        d.n_hidden = len(hs)
        for k, (w, f) in enumerate(zip(hs, nl)):
            d.hidden[k] = w
            d.hidden_nonlin[k] = self._NONLIN[f]
        d.output_nonlin = self._NONLIN[output_nonlin]
        d.feat = int(bool(feat))
        if obs_idx is not None:
            idx = [int(x) for x in obs_idx]
            d.n_obs = len(idx)
            for k, x in enumerate(idx):
                d.obs_idx[k] = x
        if noise_std is not None:
            for k, x in enumerate(np.atleast_1d(np.asarray(noise_std, dtype=np.float32))):
                d.noise_std[k] = float(x)
        if hasattr(params, "detach"):
            params = params.detach().to("cpu").numpy()
        flat = np.ascontiguousarray(np.asarray(params, dtype=np.float32).reshape(-1))
        self._check(self._lib.vs_set_policy_fnn(self._h, C.byref(d), flat.ctypes.data_as(C.c_void_p), flat.size),
                    "vs_set_policy_fnn")

    def step_policy(self, k_steps=1, record=False, noise_seed=0):
        """k_steps env steps in one launch with the network of set_policy_fnn in the loop (rollout() with act = policy(obs))"""
        if record and getattr(self, "_traj_t0", 0) + k_steps > self._traj_cap:
            self.set_traj_capacity(getattr(self, "_traj_t0", 0) + int(k_steps))
        self._check(self._lib.vs_step_policy(self._h, int(k_steps), int(bool(record)), int(noise_seed) & (2 ** 64 - 1)),
                    "vs_step_policy")

    def sync(self):
        self._check(self._lib.vs_sync(self._h), "vs_sync")

    def use_stream(self, stream_ptr):
        """Launch on the caller's stream: `torch.cuda.current_stream().cuda_stream` (0 = the legacy default stream, passed
        on as hipStreamLegacy) or any other hipStream_t; None restores the handle's own stream.  On the caller's own stream
        the kernels are ordered with its torch ops for free; the handle's blocking stream is ordered with the legacy default
        stream too, but every hand-over between the two costs an implicit synchronisation (~25 us per step in a loop)."""
        HIP_STREAM_LEGACY = 1
        if stream_ptr is None:
            arg = None
        else:
            arg = C.c_void_p(int(stream_ptr) or HIP_STREAM_LEGACY)
        self._check(self._lib.vs_set_stream(self._h, arg), "vs_set_stream")

    # ------------------------------------------------------------------------------------------------ data access
    def _rows(self, which):
        dt, rows = _BUF_DTYPES[which]
        return np.dtype(dt), (self.dims[rows] if isinstance(rows, str) else rows)

    def get(self, which):
        """Host copy of a per-env buffer as [N, dim] (or [N] for scalar buffers)."""
        dt, rows = self._rows(which)
        if rows == 0:
            return np.zeros((self.n_envs, 0), dtype=dt)
        buf = np.empty((rows, self.ld), dtype=dt)
        self._check(self._lib.vs_copy_to_host(self._h, which, buf.ctypes.data_as(C.c_void_p)), "vs_copy_to_host")
        out = buf[:, : self.n_envs]
        if isinstance(_BUF_DTYPES[which][1], int):
            return out[0].copy()
        return np.ascontiguousarray(out.T)

    def put(self, which, value):
        """`state` / hidden / step-count assignment from the host ([N, dim])."""
        dt, rows = self._rows(which)
        if rows == 0:
            return
        val = np.asarray(value, dtype=dt)
        if val.ndim == 1:
            val = val[:, None]
        if val.shape != (self.n_envs, rows):
            raise ShapeErr(given=val, expected_match=(self.n_envs, rows))
        buf = np.zeros((rows, self.ld), dtype=dt)
        buf[:, : self.n_envs] = val.T
        buf[:, self.n_envs:] = val.T[:, -1:]
        self._check(self._lib.vs_copy_from_host(self._h, which, buf.ctypes.data_as(C.c_void_p)), "vs_copy_from_host")

    def tensor(self, which):
        """Zero-copy torch view [rows, ld] of a device buffer (rows = dim of the buffer)."""
        import torch

        dt, rows = self._rows(which)
        ptr = self._lib.vs_get(self._h, which)
        if not ptr or rows == 0:
            raise ValueErr(msg=f"buffer {which} is not available")
        arr = _DevArray(ptr, (rows, self.ld), {"f4": "<f4", "u1": "|u1", "i4": "<i4", "u4": "<u4"}[dt.str[1:]], self)
        return torch.as_tensor(arr, device=f"cuda:{self.device}")

    def traj(self, k_steps):
        """Host copies of the recorded trajectory buffers of the last step_random(record=True): dict of [T, N, dim]"""
        self.sync()
        tt = self.traj_tensors(k_steps)
        return {k: v.cpu().numpy() for k, v in tt.items() if k != "rec"}

    def set_episode_log(self, on=True):
        """Opt-in per-episode log (ballot-compacted ring). Off by default: see include/vecsim.h."""
        self._check(self._lib.vs_set_episode_log(self._h, int(bool(on))), "vs_set_episode_log")

    def episode_stats(self, clear=False):
        """Per-env accumulators of completed episodes since the last clear: (count [N], return sum [N], length sum [N])"""
        out = (self.get(L.VS_EPSTAT_COUNT), self.get(L.VS_EPSTAT_RETSUM), self.get(L.VS_EPSTAT_LENSUM))
        if clear:
            self._check(self._lib.vs_clear_episodes(self._h), "vs_clear_episodes")
        return out

    def episodes(self, clear=True):
        """Episode log (needs set_episode_log(True)): completed episodes since the last clear as
        (returns [M], lengths [M], env index [M])."""
        cnt = np.zeros(1, dtype=np.uint32)
        self._check(self._lib.vs_copy_to_host(self._h, L.VS_EP_COUNT, cnt.ctypes.data_as(C.c_void_p)), "vs_copy_to_host")
        cap = max(self.ld, 1 << 16)
        m = int(min(cnt[0], cap))
        ret = np.empty(cap, dtype=np.float32)
        ln = np.empty(cap, dtype=np.int32)
        ix = np.empty(cap, dtype=np.int32)
        for which, buf in ((L.VS_EP_RETURNS, ret), (L.VS_EP_LENGTHS, ln), (L.VS_EP_ENVIDX, ix)):
            self._check(self._lib.vs_copy_to_host(self._h, which, buf.ctypes.data_as(C.c_void_p)), "vs_copy_to_host")
        if clear:
            self._check(self._lib.vs_clear_episodes(self._h), "vs_clear_episodes")
        return ret[:m].copy(), ln[:m].copy(), ix[:m].copy()

    def error_count(self):
        return int(self._lib.vs_error_count(self._h))

    def raise_on_error(self):
        """The reference raises pyrado.ValueErr on a NaN action/state (P/spaces/box.py:142-146, rollout.py:193-230)."""
        n = self.error_count()
        if n > 0:
            raise ValueErr(msg=f"At least one value is NaN! ({n} environment(s) flagged)")

    def timer_start(self):
        """HIP-event stopwatch on the handle's stream (vs_timer_start / vs_timer_stop)"""
        self._check(self._lib.vs_timer_start(self._h), "vs_timer_start")

    def timer_stop(self):
        """device time [ms] of the launches issued since timer_start (waits for them)"""
        ms = C.c_float()
        self._check(self._lib.vs_timer_stop(self._h, C.byref(ms)), "vs_timer_stop")
        return float(ms.value)

    def time_step_kernel(self, iters=100, actions=None, k_steps=1, record=False):
        """Average device time [ms] per launch of the step kernel (hipEvents on the kernel's stream)."""
        if record and k_steps > self._traj_cap:
            self._check(self._lib.vs_set_traj_capacity(self._h, int(k_steps)), "vs_set_traj_capacity")
            self._traj_cap = int(k_steps)
        ms = C.c_float()
        if actions is not None:
            A = self.dims["A"]
            es, ds = (actions.stride(0), actions.stride(1)) if actions.dim() == 2 and actions.shape[0] != A else \
                (actions.stride(-1), actions.stride(0) if actions.dim() == 2 else 0)
            rc = self._lib.vs_time_step_kernel(self._h, 0, C.c_void_p(actions.data_ptr()), es, ds, 0, 0, int(iters), C.byref(ms))
        else:
            rc = self._lib.vs_time_step_kernel(self._h, 1, None, 0, 0, int(k_steps), int(bool(record)), int(iters), C.byref(ms))
        self._check(rc, "vs_time_step_kernel")
        return float(ms.value)


class MixedVecSimEnv:
    """Several env families stepped by ONE launch (BASELINE config 5): a group of VecSimEnv handles on one device, lanes
    sorted by type (each member is one contiguous segment).  Parameters, resets and data access stay with the members."""

    def __init__(self, members):
        members = list(members)
        if not 1 <= len(members) <= 5 or not all(isinstance(m, VecSimEnv) for m in members):
            raise TypeErr(given=members, expected_type="1..5 VecSimEnv")
        self.members = members
        self._lib = L.load()
        arr = (C.c_void_p * len(members))(*[m._h for m in members])
        h = C.c_void_p()
        rc = self._lib.vs_mixed_create(arr, len(members), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"vs_mixed_create failed ({rc}): {self._lib.vs_last_error(None).decode()}")
        self._h = h
        off = 0
        for m in members:  # global lane indices: independent random streams across the segments
            m.set_index_offset(off)
            off += m.n_envs

    @property
    def n_envs(self):
        return sum(m.n_envs for m in self.members)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.vs_mixed_last_error(self._h).decode()}")

    def step_random(self, k_steps=1, seed=0, record=False):
        if record:
            for m in self.members:
                if k_steps > m._traj_cap:
                    m._check(m._lib.vs_set_traj_capacity(m._h, int(k_steps)), "vs_set_traj_capacity")
                    m._traj_cap = int(k_steps)
        self._check(self._lib.vs_mixed_step_random(self._h, int(seed) & (2 ** 64 - 1), int(k_steps), int(bool(record))),
                    "vs_mixed_step_random")

    def step(self, actions):
        """actions: one torch CUDA tensor [N_q, A_q] per member"""
        n = len(self.members)
        ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in actions])
        es = (C.c_int64 * n)(*[a.stride(0) for a in actions])
        ds = (C.c_int64 * n)(*[a.stride(1) if a.dim() == 2 else 0 for a in actions])
        self._keep = actions
        self._check(self._lib.vs_mixed_step(self._h, ptrs, es, ds), "vs_mixed_step")

    def time_random(self, k_steps, record=False, iters=20, seed=0):
        if record:
            for m in self.members:
                if k_steps > m._traj_cap:
                    m._check(m._lib.vs_set_traj_capacity(m._h, int(k_steps)), "vs_set_traj_capacity")
                    m._traj_cap = int(k_steps)
        ms = C.c_float()
        self._check(self._lib.vs_mixed_time_random(self._h, int(seed), int(k_steps), int(bool(record)), int(iters), C.byref(ms)),
                    "vs_mixed_time_random")
        return float(ms.value)

    def sync(self):
        self.members[0].sync()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vs_mixed_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
