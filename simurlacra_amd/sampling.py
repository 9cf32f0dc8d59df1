"""
Host-side mirror of the sampling boundary: StepSequence (the result container algorithms consume,
P/sampling/step_sequence.py:223-362), rollout() (P/sampling/rollout.py:63-342) and ParallelRolloutSampler
(P/sampling/parallel_rollout_sampler.py:182-323).

The reference distributes rollouts over a pool of worker processes, each stepping one pickled env copy
(P/sampling/sampler_pool.py).  Here every rollout of a `sample()` call is one wavefront lane of ONE batched libvecsim
handle on the GPU; `num_workers` is accepted for interface compatibility and ignored.  What is kept:
  * the work list: `min_rollouts` rollouts, or every init_state / domain_param / their Cartesian product repeated
    `ceil(min_rollouts / len)` times (parallel_rollout_sampler.py:280-304), or rollouts until `min_steps` steps are
    collected, surplus dropped in index order (sampler_pool.py:392-469);
  * the seeding contract: rollout number `i` of the `k`-th `sample()` call depends on `(seed, k, i)` only
    (rollout.py:139, set_seed of P/__init__.py:135-183), never on how rollouts are grouped -- so results are identical for
    every `num_workers` (Pyrado/tests/test_sampling.py:589-700);
  * the result: a list of StepSequence in rollout order.
"""
import math
from itertools import product
from math import ceil
from typing import List, Optional

import numpy as np

from . import _lib as L
from .exceptions import TypeErr, ValueErr
from .policies import DummyPolicy
from .seeding import derive_seed, get_base_seed, set_seed
from .wrappers import DomainRandWrapperBuffer, DomainRandWrapperLive, fuse_wrappers, inner_env, typed_env

NO_SEED = object()


class StepSequence:
    """Rollout container: observations [T+1, O], actions [T, A], rewards [T], optionally states [T+1, S].
    The subset of P/sampling/step_sequence.py that samplers and on-policy algorithms touch."""

    def __init__(self, *, observations, actions, rewards, states=None, time=None, rollout_info=None, env_infos=None,
                 complete=True, done_last=True, **extra):
        self.observations = np.asarray(observations)
        self.actions = np.asarray(actions)
        self.rewards = np.asarray(rewards, dtype=np.float64).reshape(-1)
        if not (len(self.observations) == len(self.actions) + 1 == len(self.rewards) + 1):
            raise ValueErr(msg="observations need one entry more than actions and rewards")
        self.states = None if states is None else np.asarray(states)
        self.time = np.asarray(time) if time is not None else None
        self.rollout_info = rollout_info or {}
        self.env_infos = env_infos
        self.complete = complete
        self.done = np.zeros(len(self.rewards), dtype=bool)
        if len(self.done) and done_last:
            self.done[-1] = True
        for k, v in extra.items():
            setattr(self, k, v)

    @classmethod
    def _packed(cls, observations, actions, rewards, rollout_info, done_last, dt, init_state):
        """views into the sampler's packed arrays: no copies, no validation"""
        ro = cls.__new__(cls)
        ro.observations, ro.actions = observations, actions
        ro.rewards = rewards.astype(np.float64)
        ro.states, ro.env_infos, ro.complete = None, None, True
        ro.rollout_info = rollout_info
        ro.time = np.arange(len(rewards) + 1) * dt
        ro.done = np.zeros(len(rewards), dtype=bool)
        if len(rewards) and done_last:
            ro.done[-1] = True
        ro.init_state = init_state
        return ro

    @property
    def length(self) -> int:
        return len(self.rewards)

    def __len__(self):
        return self.length

    def undiscounted_return(self) -> float:
        return float(np.sum(self.rewards))

    def discounted_return(self, gamma: float) -> float:
        return float(np.sum(self.rewards * gamma ** np.arange(self.length)))

    def __repr__(self):
        return f"StepSequence(len={self.length}, return={self.undiscounted_return():.4g})"


def rollout(env, policy, eval: bool = False, max_steps: Optional[int] = None, reset_kwargs: Optional[dict] = None,
            render_mode=None, render_step: int = 1, no_reset: bool = False, no_close: bool = False,
            record_dts: bool = False, stop_on_done: bool = True, seed: Optional[int] = None,
            sub_seed: Optional[int] = None, sub_sub_seed: Optional[int] = None) -> StepSequence:
    """One rollout of ONE environment object through its reset()/step() surface (rollout.py:63-342): the reference's
    loop and keyword arguments (rendering is a no-op for these simulations), one kernel launch per step.  For throughput
    use ParallelRolloutSampler, which batches rollouts as lanes."""
    import time

    import torch

    if not (isinstance(reset_kwargs, dict) or reset_kwargs is None):
        raise TypeErr(given=reset_kwargs, expected_type=dict)
    if max_steps is not None:
        env.max_steps = max_steps
    if seed is not None:
        set_seed(seed, sub_seed, sub_sub_seed)
    obs = np.zeros(env.obs_space.shape) if no_reset else env.reset(**(reset_kwargs or {}))
    if hasattr(policy, "reset"):
        policy.reset()
    if hasattr(policy, "eval") and hasattr(policy, "train"):
        policy.eval() if eval else policy.train()
    env.render(render_mode, render_step=1)
    obs_hist, act_hist, act_app_hist, rew_hist, state_hist, t_hist = [], [], [], [], [], [0.0]
    dts = dict(dts_policy=[], dts_step=[], dts_remainder=[])
    done, t = False, 0.0
    t_post_step = time.time()  # the first remainder sample is meaningless, as in the reference
    while not (done and stop_on_done) and env.curr_step < env.max_steps:
        t_start = time.time()
        dts["dts_remainder"].append(t_start - t_post_step)
        if np.isnan(obs).any():
            raise ValueErr(msg="At least one observation value is NaN!")
        with torch.no_grad():
            act = policy(torch.from_numpy(np.asarray(obs)).to(torch.get_default_dtype()))
        act = act.detach().cpu().numpy()
        if np.isnan(act).any():
            raise ValueErr(msg="At least one action value is NaN!")
        t_post_policy = time.time()
        dts["dts_policy"].append(t_post_policy - t_start)
        state = env.state.copy()
        obs_next, rew, done, _ = env.step(act)
        act_app_hist.append(env.limit_act(act))
        t_post_step = time.time()
        dts["dts_step"].append(t_post_step - t_post_policy)
        obs_hist.append(np.asarray(obs))
        act_hist.append(act)
        rew_hist.append(rew)
        state_hist.append(state)
        t += env.dt
        t_hist.append(t)
        env.render(render_mode, render_step)
        obs = obs_next
    if not no_close:
        env.close()  # nothing to disconnect from for a simulation; the device handle stays
    obs_hist.append(np.asarray(obs))
    state_hist.append(env.state.copy())
    info = dict(env_name=env.name, env_spec=env.spec, domain_param=env.domain_param)
    # QCartPoleSim.reset returns the state instead of the observation (quirk Q5): keep lists when shapes differ
    try:
        observations = np.stack(obs_hist)
    except ValueError:
        observations = np.empty(len(obs_hist), dtype=object)
        observations[:] = obs_hist
    extra = {k: np.asarray(v) for k, v in dts.items()} if record_dts else {}
    return StepSequence(observations=observations, actions=np.stack(act_hist), rewards=rew_hist,
                        states=np.stack(state_hist), time=t_hist, rollout_info=info, done_last=bool(done),
                        actions_applied=np.stack(act_app_hist), **extra)


class ParallelRolloutSampler:
    """Drop-in for P/sampling/parallel_rollout_sampler.py:182-323 with the rollouts batched on the GPU."""

    def __init__(self, env, policy, num_workers: int = 1, *, min_rollouts: int = None, min_steps: int = None,
                 show_progress_bar: bool = False, seed=NO_SEED, batch_lanes: int = 4096, chunk: int = 128):
        if min_rollouts is None and min_steps is None:
            raise ValueErr(msg="At least one of min_rollouts and min_steps must be given")  # SamplerBase
        self.min_rollouts, self.min_steps = min_rollouts, min_steps
        self.env, self.policy = env, policy
        self.num_workers = num_workers  # ignored: lanes replace worker processes
        self.show_progress_bar = show_progress_bar
        if seed is NO_SEED:
            seed = get_base_seed()
        self._seed = seed
        self._sample_count = -1
        self._batch_lanes = int(batch_lanes)
        self._chunk = int(chunk)
        self._vecs = {}

    def reinit(self, env=None, policy=None):
        if env is not None:
            self.env = env
            self._vecs = {}
        if policy is not None:
            self.policy = policy

    # ------------------------------------------------------------------------------------------------ work list
    def work_list(self, init_states, domain_params):
        """[(init_state | None, domain_param | None)] in rollout order (parallel_rollout_sampler.py:280-304)"""
        n = self.min_rollouts
        if init_states is None and domain_params is None:
            return [(None, None)] * n
        if init_states is not None and domain_params is None:
            rep = ceil(n / len(init_states))
            return [(s, None) for s in rep * list(init_states)]
        if init_states is None:
            rep = ceil(n / len(domain_params))
            return [(None, d) for d in rep * list(domain_params)]
        allcombs = list(product(init_states, domain_params))
        rep = ceil(n / len(allcombs))
        return rep * allcombs

    def _key(self):
        """64-bit Philox key of this sample() call: MD5-derived like set_seed(seed, sub_seed=sample_count)"""
        if self._seed is None:
            return int(np.random.randint(0, 2 ** 31 - 1)) << 20 | (self._sample_count & 0xFFFFF)
        return (derive_seed(self._seed, self._sample_count, 0) << 32) | derive_seed(self._seed, self._sample_count, 1)

    # ------------------------------------------------------------------------------------------------ batched run
    def _vec_for(self, n):
        """a libvecsim handle with n lanes configured like self.env (ctor args, task, domain params)"""
        from .vec_env import VecSimEnv

        base = inner_env(self.env)
        key = n
        if key not in self._vecs:
            ctor = dict(base._ctor)
            ctor.pop("num_envs", None)
            ctor.pop("load_experimental_tholds", None)
            ctor.pop("mass", None)
            dev = ctor.pop("device", 0)
            self._vecs[key] = VecSimEnv(base.name, n, ctor.pop("dt"), ctor.pop("max_steps"),
                                        task_args=ctor.pop("task_args") or None, device=dev, **ctor)
        v = self._vecs[key]
        v.set_randomizer([])
        v.set_param_buffer(None)
        v.set_params_uniform(base.domain_param)
        self._fc = fuse_wrappers(self.env)  # ActNorm / act noise / act delay / obs norm / obs noise / partial obs
        return v

    def _run_batch(self, work, first_index, eval):
        """run len(work) rollouts as lanes; returns List[StepSequence] in order"""
        import torch

        n = len(work)
        v = self._vec_for(n)
        # the library's kernels and the torch ops that read its buffers must be ordered: on torch's legacy default stream
        # (pointer 0) the handle's own blocking stream is ordered with it implicitly, any other current stream is handed
        # to the handle
        v.use_stream(torch.cuda.current_stream(v.device).cuda_stream or None)
        try:
            return self._run_batch_on_stream(v, work, first_index, eval)
        finally:
            v.use_stream(None)

    def _run_batch_on_stream(self, v, work, first_index, eval):
        import torch

        n = len(work)
        base = inner_env(self.env)
        max_steps = base.max_steps
        if max_steps == math.inf:
            raise ValueErr(msg="ParallelRolloutSampler needs a finite env.max_steps")
        S, O, A = v.dims["S"], v.dims["O"], v.dims["A"]
        key = self._key()
        # lane j of this batch is rollout number first_index + j of the sample() call: its Philox streams are keyed by that
        # number, so the result does not depend on how the work list is cut into batches
        lane_key = key
        v.set_index_offset(first_index)
        v.seek_random(0)
        self._fc.apply(v, seed=lane_key)  # the wrappers of the chain, fused into the kernels; noise keyed per sample() call
        dps = [w[1] for w in work]
        live = typed_env(self.env, DomainRandWrapperLive)
        if any(d is not None for d in dps):
            mat = np.tile(np.array([base.domain_param[k] for k in v.param_names], dtype=np.float32), (n, 1))
            for j, d in enumerate(dps):
                if d is not None:
                    for k, val in d.items():
                        mat[j, v.param_names.index(k)] = float(np.asarray(val).reshape(-1)[0])
            v.set_params(mat)
        elif live is not None:
            v.sample_params(live.randomizer.device_specs(), seed=lane_key ^ 0xD1B54A32D192ED03)
        else:
            ring = typed_env(self.env, DomainRandWrapperBuffer)
            if ring is not None and ring.buffer:
                # rollout number r takes set r mod len(buffer) (cyclic) or a random one: the ring of the reference, per lane
                v.set_param_buffer([ring.buffer] if isinstance(ring.buffer, dict) else ring.buffer, ring.selection)
        inits = [w[0] for w in work]
        v.set_auto_reset(False)
        v.reset(seed=lane_key)  # init-space sample for every lane ...
        if any(s is not None for s in inits):  # ... overridden by the explicit init states
            width = {len(np.asarray(s).reshape(-1)) for s in inits if s is not None}
            if len(width) != 1:
                raise ValueErr(msg="all init states must have the same shape")
            w_ = width.pop()
            mask = np.array([s is not None for s in inits], dtype=np.uint8)
            arr = np.zeros((n, w_), dtype=np.float32)
            for j, s in enumerate(inits):
                if s is not None:
                    arr[j] = np.asarray(s, dtype=np.float32).reshape(-1)
            v.reset(init_state=arr, mask=mask, seed=lane_key)
        dev = f"cuda:{v.device}"
        obs_full = v.tensor(L.VS_OBS)[:, :n]
        keep = None if self._fc.keep.all() else torch.from_numpy(np.flatnonzero(self._fc.keep)).to(dev)

        def visible(x, dim):  # ObsPartialWrapper: the rows of the observation the outermost env reports
            return x if keep is None else x.index_select(dim, keep)
        done_t, st_t = v.tensor(L.VS_DONE)[0, :n], v.tensor(L.VS_STATE)[:, :n]
        rew_t = v.tensor(L.VS_REW)[0, :n]
        use_fused = isinstance(self.policy, DummyPolicy)
        state0 = st_t.t().clone()
        T_cap = int(max_steps)
        t = 0
        if use_fused:
            # rollout() with DummyPolicy == vs_step_random: fused steps, on-device uniform actions, lanes freeze at done.
            # Consecutive launches fill ONE device-side trajectory buffer; nothing is copied to the host inside the loop.
            v.set_traj_capacity(T_cap)
            while t < T_cap:
                k = int(min(self._chunk, T_cap - t))
                v.set_traj_offset(t)
                v.step_random(k, seed=lane_key ^ 0xA0761D6478BD642F, record=True)
                t += k
                if bool(done_t.bool().all()):  # one scalar sync per launch
                    break
            v.set_traj_offset(0)
            tt = v.traj_tensors(t, n)
            obs_T, act_T = visible(tt["obs"], 2), tt["act"]  # [T, n, dim]
            rew_T, done_T = tt["rew"], tt["done"].bool()  # [T, n]
        else:
            policy = self.policy.to(dev) if hasattr(self.policy, "to") else self.policy
            if hasattr(policy, "eval"):
                policy.eval() if eval else policy.train()
            obs_rec, act_rec, rew_rec, done_rec = [], [], [], []
            alive = torch.ones(n, dtype=torch.bool, device=dev)
            with torch.no_grad():
                while t < T_cap:
                    obs_now = visible(obs_full, 0).t().contiguous()  # [n, O']: what the policy sees and what is recorded
                    act = policy(obs_now).to(torch.float32).reshape(n, A).contiguous()
                    obs_rec.append(obs_now)
                    act_rec.append(act)
                    v.step(act)
                    rew_rec.append(rew_t.clone())
                    done_rec.append(done_t.clone())
                    t += 1
                    if t % 32 == 0 or t == T_cap:
                        alive &= ~torch.stack(done_rec[-32:]).bool().any(dim=0)
                        if not bool(alive.any()):
                            break
            obs_T, act_T = torch.stack(obs_rec), torch.stack(act_rec)  # [T, n, dim]
            rew_T, done_T = torch.stack(rew_rec), torch.stack(done_rec).bool()
        v.raise_on_error()
        # ---- split into rollouts on the device: rollout j = steps 0 .. first done of lane j, packed lane-major ----
        T = t
        ar = torch.arange(n, device=dev)
        any_done = done_T.any(dim=0)
        first = torch.where(any_done, done_T.to(torch.uint8).argmax(dim=0), torch.full_like(ar, T - 1))
        length = first + 1  # [n]
        tgrid = torch.arange(T + 1, device=dev)[None, :]
        mask = tgrid[:, :T] < length[:, None]  # [n, T]
        mask_o = tgrid <= length[:, None]  # [n, T + 1]: one observation more than steps
        obs_ext = torch.cat([obs_T, visible(obs_full, 0).t()[None]], dim=0)  # the row after the last step: the current obs
        obs_p = obs_ext.permute(1, 0, 2)[mask_o].cpu().numpy()
        act_p = act_T.permute(1, 0, 2)[mask].cpu().numpy()
        rew_p = rew_T.t()[mask].cpu().numpy()
        done_last = done_T[first, ar].cpu().numpy()
        length_h = length.cpu().numpy()
        state0_h = state0.cpu().numpy()
        params = v.get(L.VS_PARAMS)
        off = np.concatenate([[0], np.cumsum(length_h)])
        off_o = np.concatenate([[0], np.cumsum(length_h + 1)])
        dt = base.dt
        ros = []
        for j in range(n):
            Lj = int(length_h[j])
            info = dict(env_name=base.name, domain_param=dict(zip(v.param_names, params[j].tolist())),
                        rollout_number=first_index + j)
            ros.append(StepSequence._packed(obs_p[off_o[j]:off_o[j + 1]], act_p[off[j]:off[j + 1]], rew_p[off[j]:off[j + 1]],
                                            info, bool(done_last[j]), dt, state0_h[j]))
        return ros

    def sample(self, init_states: Optional[List[np.ndarray]] = None, domain_params: Optional[List[dict]] = None,
               eval: bool = False) -> List[StepSequence]:
        self._sample_count += 1
        if self.min_steps is None:
            work = self.work_list(init_states, domain_params)
            out = []
            for a in range(0, len(work), self._batch_lanes):
                out += self._run_batch(work[a:a + self._batch_lanes], a, eval)
            return out
        if init_states is not None:
            raise NotImplementedError  # as in the reference (parallel_rollout_sampler.py:315-316)
        # run_collect: rollouts in index order until min_steps (and min_rollouts) are reached, surplus dropped
        out, steps, idx = [], 0, 0
        guess = max(self.min_rollouts or 1, 1)
        while True:
            nb = int(min(self._batch_lanes, max(guess, 64)))
            batch = self._run_batch([(None, None)] * nb, idx, eval)
            for ro in batch:
                out.append(ro)
                steps += len(ro)
                if steps >= self.min_steps and len(out) >= (self.min_rollouts or 0):
                    return out
            idx += nb
            guess = max(guess, int(len(out) * (self.min_steps / max(steps, 1) - 1)) + 1)
