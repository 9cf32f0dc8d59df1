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
import gc
import math
import os
from itertools import product
from math import ceil
from typing import List, Optional

import numpy as np

from . import _lib as L
from .exceptions import TypeErr, ValueErr
from .policies import DummyPolicy, fnn_kernel_spec
from .seeding import derive_seed, get_base_seed, set_seed
from .wrappers import DomainRandWrapperBuffer, DomainRandWrapperLive, fuse_wrappers, inner_env, typed_env

NO_SEED = object()


class StepSequence:
    """Rollout container: observations [T+1, O], actions [T, A], rewards [T], optionally states [T+1, S].
    The subset of P/sampling/step_sequence.py that samplers and on-policy algorithms touch
    (`simurlacra_amd.pyrado_compat.to_pyrado_step_sequence` gives Pyrado's full class when it is installed).
    `time`, `done` and `rollout_info` of a sampler-made rollout are materialised on first access: building 4 096 of these
    per `sample()` call is host work that would otherwise cost as much as the GPU part."""

    def __init__(self, *, observations, actions, rewards, states=None, time=None, rollout_info=None, env_infos=None,
                 complete=True, done_last=True, **extra):
        self.observations = np.asarray(observations)
        self.actions = np.asarray(actions)
        self.rewards = np.asarray(rewards, dtype=np.float64).reshape(-1)
        if not (len(self.observations) == len(self.actions) + 1 == len(self.rewards) + 1):
            raise ValueErr(msg="observations need one entry more than actions and rewards")
        self.states = None if states is None else np.asarray(states)
        self._time = np.asarray(time) if time is not None else None
        self._dt = None
        self._info = rollout_info or {}
        self._info_src = None
        self.env_infos = env_infos
        self.complete = complete
        self._done = None
        self._done_last = bool(done_last)
        for k, v in extra.items():
            setattr(self, k, v)

    @classmethod
    def _packed(cls, observations, actions, rewards, info_src, done_last, dt, init_state, states=None,
                actions_applied=None, th_ddot=None):
        """views into the sampler's packed arrays (rewards already float64): no copies, no validation.
        info_src = (env_name, param_names, params_row, rollout_number)"""
        ro = cls.__new__(cls)
        ro.observations, ro.actions, ro.rewards = observations, actions, rewards
        ro.states = states  # [T + 1, S]: the state before every step and the final one (rollout.py:253, 306)
        if actions_applied is not None:
            ro.actions_applied = actions_applied  # [T, A]: env.limit_act(act) (rollout.py:244)
        if th_ddot is not None:
            ro.th_ddot = th_ddot  # [T + 1]: the fork's hidden pole acceleration before every step (rollout.py:238, 307)
        ro.env_infos = ro._time = ro._done = ro._info = None
        ro.complete = True
        ro._info_src = info_src
        ro._dt = dt
        ro._done_last = done_last
        ro.init_state = init_state
        return ro

    @property
    def time(self):
        if self._time is None and self._dt is not None:
            self._time = np.arange(len(self.rewards) + 1) * self._dt
        return self._time

    @property
    def done(self) -> np.ndarray:
        if self._done is None:
            self._done = np.zeros(len(self.rewards), dtype=bool)
            if len(self._done) and self._done_last:
                self._done[-1] = True
        return self._done

    @property
    def rollout_info(self) -> dict:
        if self._info is None:
            name, pnames, row, number = self._info_src
            self._info = dict(env_name=name, domain_param=dict(zip(pnames, row.tolist())), rollout_number=number)
        return self._info

    @property
    def length(self) -> int:
        return len(self.rewards)

    def __len__(self):
        return len(self.rewards)

    def undiscounted_return(self) -> float:
        return float(np.sum(self.rewards))

    def discounted_return(self, gamma: float) -> float:
        return float(np.sum(self.rewards * gamma ** np.arange(self.length)))

    def __repr__(self):
        return f"StepSequence(len={self.length}, return={self.undiscounted_return():.4g})"


_PLAIN_WORK = (None, None)  # a rollout without an explicit init state or domain parameters


class PackedRollouts:
    """The rollouts of one batch of lanes as packed DEVICE tensors -- what `ParallelRolloutSampler.sample()` holds right before it
    copies to the host and cuts `StepSequence`s, handed out as it is (`sample_packed()`): nothing crosses PCIe and no per-rollout
    Python object is made.  For algorithm code that consumes rollouts on the GPU (the concatenated form the reference's
    algorithms build with `StepSequence.concat`, P/sampling/step_sequence.py:777-825 -- which drops every rollout's final observation
    and state (truncate_last); here they are kept).

    ONE matrix `rows` [total + n, F] holds everything (round 3; six separate arrays before): rollout j (j = 0 .. n - 1, in rollout
    order) owns rows `offsets[j] + j : offsets[j + 1] + j + 1` -- its `lengths[j]` steps and, last, the entry behind them (final
    observation / state / hidden state; the per-step fields of that row are 0) -- and every field is a strided VIEW of it, all
    indexed by the same rows:

      observations [total + n, O]   actions [total + n, A]   rewards [total + n] (float32)
      states [total + n, S] | None  actions_applied [total + n, A] | None   th_ddot [total + n] | None  (full records)
      lengths [n] (int64)   offsets [n + 1] (int64: exclusive cumulative lengths)   total = offsets[n]
      done_last [n] (bool: the rollout ended by done, not by the step limit)
      init_states [n, S]    first_index: rollout number of rollout 0 within the sample() call

    `step_slice(j)` are the rows of rollout j's steps, `obs_slice(j)` those plus the final entry; `step_rows()` the row indices of
    all steps (for a dense, reference-style concatenation: `p.actions[p.step_rows()]`).
    """

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        return int(self.lengths.shape[0])

    @property
    def total_steps(self) -> int:
        return int(self.total)

    def step_slice(self, j: int) -> slice:
        a = int(self.offsets[j]) + j
        return slice(a, a + int(self.lengths[j]))

    def obs_slice(self, j: int) -> slice:
        return slice(int(self.offsets[j]) + j, int(self.offsets[j + 1]) + j + 1)

    def row_rollout_index(self):
        """[total + n] int64: the rollout every ROW (steps and final entries) belongs to"""
        import torch

        return torch.repeat_interleave(torch.arange(len(self), device=self.lengths.device), self.lengths + 1,
                                       output_size=self.total_steps + len(self))

    def rollout_index(self):
        """[total] int64: the rollout every packed step belongs to"""
        import torch

        return torch.repeat_interleave(torch.arange(len(self), device=self.lengths.device), self.lengths,
                                       output_size=self.total_steps)

    def step_rows(self):
        """[total] int64: the row of every step, rollout after rollout (the rows without the final entries)"""
        import torch

        return torch.arange(self.total_steps, device=self.lengths.device) + self.rollout_index()

    def undiscounted_returns(self):
        """[n] float32 on the device (the final entries carry a reward of 0)"""
        import torch

        out = torch.zeros(len(self), device=self.rewards.device, dtype=self.rewards.dtype)
        return out.index_add_(0, self.row_rollout_index(), self.rewards)


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
    """Drop-in for P/sampling/parallel_rollout_sampler.py:182-323 with the rollouts batched on the GPU.

    Every rollout of a sample() call is a lane of one handle, `batch_lanes` at a time (default 65 536: one full wave of
    envs per SIMD; the record buffer of such a batch is T x F x 65 536 floats, 8.4 GB for 4 000-step QQube rollouts -- sized
    for a 288 GB device; pass a smaller value on less).  `num_workers` is accepted and ignored."""

    def __init__(self, env, policy, num_workers: int = 1, *, min_rollouts: int = None, min_steps: int = None,
                 show_progress_bar: bool = False, seed=NO_SEED, batch_lanes: int = 65536, chunk: int = 128,
                 full_records: bool = True, fuse_policy: bool = True, graph_policy: bool = False, owned_arrays: bool = False):
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
        # full_records: rollouts carry states / actions_applied / th_ddot like the reference's (rollout.py:305-325), at 13
        # instead of 8 floats per QQube step on the device; False keeps observations / actions / rewards only
        self._full = bool(full_records)
        # fuse_policy: evaluate FNN / FNNPolicy networks (optionally inside a NormalActNoiseExplStrat) in the rollout kernel
        # itself; False keeps every policy but DummyPolicy in torch (one recording step launch per env step)
        self._fuse_policy = bool(fuse_policy)
        # graph_policy (opt-in): a policy that stays a torch module is stepped through a captured hipGraph of 32 (observation,
        # policy, recording step) iterations instead of ~10 eager launches per env step -- for policies whose forward() is
        # capturable (no host synchronisation, no data-dependent Python control flow).  A deterministic policy gives the eager
        # path's rollouts bit for bit (test); a stochastic one draws from the same distribution but not the same numbers (the two
        # warm-up steps before the capture advance torch's generator, and a captured generator offset is replayed per graph launch)
        self._graph_policy = bool(graph_policy)
        # owned_arrays: the arrays of the rollouts sample() returns are pageable copies the caller owns.  Default (False): they
        # are VIEWS of one page-locked block per field and call (~7.7 GB for 65 536 full-record QQube rollouts, 57 GB/s instead
        # of 18) -- a block stays pinned as long as ANY rollout of that call is alive, and torch's caching host allocator keeps it
        # afterwards: a caller that retains a few rollouts for long (a replay buffer, CVaRSampler's epsilon-fraction) should ask
        # for owned arrays or copy what it keeps
        self._owned = bool(owned_arrays)
        self._vecs = {}

    def _drop_handles(self):
        for v in self._vecs.values():
            v.close()
        self._vecs = {}

    def close(self):
        """release the device handles and the host-side conversion threads (idempotent)"""
        self._drop_handles()
        pool = self.__dict__.pop("_copy_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reinit(self, env=None, policy=None):
        if env is not None:
            self.env = env
            self._drop_handles()
        if policy is not None:
            self.policy = policy

    def set_min_count(self, min_rollouts=None, min_steps=None):
        """SamplerBase.set_min_count (P/sampling/sampler.py): at least one of the two"""
        if min_rollouts is None and min_steps is None:
            raise ValueErr(msg="At least one of min_rollouts and min_steps must be given")
        self.min_rollouts, self.min_steps = min_rollouts, min_steps

    # ------------------------------------------------------------------------------------------------ work list
    def work_list(self, init_states, domain_params):
        """[(init_state | None, domain_param | None)] in rollout order (parallel_rollout_sampler.py:280-304)"""
        n = self.min_rollouts
        if init_states is None and domain_params is None:
            return [_PLAIN_WORK] * n
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
        # ONE handle at a time, sized to the exact lane count (the padding lanes of a bigger one would run, and record,
        # rollouts nobody asked for); a different count closes the previous handle and its record buffers -- the
        # min_steps loop changes its batch size every round and must not grow device memory
        key = n
        if key not in self._vecs:
            self._drop_handles()
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

    def _run_batch(self, work, first_index, eval, packed_out=False, plain=False):
        """run len(work) rollouts as lanes; returns List[StepSequence] in order (packed_out: one PackedRollouts)"""
        import torch

        n = len(work)
        v = self._vec_for(n)
        # the library's kernels and the torch ops that read its buffers must be ordered: the handle launches on torch's
        # current stream for the duration of the batch (pointer 0 = the legacy default stream)
        v.use_stream(torch.cuda.current_stream(v.device).cuda_stream)
        try:
            return self._run_batch_on_stream(v, work, first_index, eval, packed_out, plain)
        finally:
            v.use_stream(None)

    def _run_batch_on_stream(self, v, work, first_index, eval, packed_out=False, plain=False):
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
        # (`plain`: the caller passed neither init states nor domain parameters -- every entry of the work list is (None, None);
        # the generators that look for one over 65 536 entries cost a millisecond each, five times per call)
        plain_work = bool(plain)
        dps = [None] * n if plain_work else [w[1] for w in work]
        live = typed_env(self.env, DomainRandWrapperLive)
        if not plain_work and any(d is not None for d in dps):
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
        inits = [None] * n if plain_work else [w[0] for w in work]
        has_inits = not plain_work and any(s is not None for s in inits)
        v.set_auto_reset(False)

        def reset_lanes():  # (a pure function of the seeds and the work list: the graph path calls it a second time)
            v.reset(seed=lane_key)  # init-space sample for every lane ...
            if has_inits:  # ... overridden by the explicit init states
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

        reset_lanes()
        dev = f"cuda:{v.device}"
        obs_full = v.tensor(L.VS_OBS)[:, :n]
        keep = None if self._fc.keep.all() else torch.from_numpy(np.flatnonzero(self._fc.keep)).to(dev)

        def visible(x, dim):  # ObsPartialWrapper: the rows of the observation the outermost env reports
            return x if keep is None else x.index_select(dim, keep)
        done_t, st_t = v.tensor(L.VS_DONE)[0, :n], v.tensor(L.VS_STATE)[:, :n]
        rew_t = v.tensor(L.VS_REW)[0, :n]
        H = v.dims["H"]
        hid_t = v.tensor(L.VS_HIDDEN)[:, :n] if H else None
        use_fused = isinstance(self.policy, DummyPolicy)
        # a feed-forward network policy the kernel can evaluate itself (vs_step_policy): rollout() with act = policy(obs)
        # fused like the DummyPolicy path -- unless a wrapper pipeline (noise / delay / observation normalisation) is on
        fc = self._fc
        plain_chain = (fc.delay == 0 and not np.any(fc.noise_std) and not np.any(fc.noise_mean) and not np.any(fc.var)
                       and np.all(fc.scale == 1) and not np.any(fc.shift))
        fnn = fnn_kernel_spec(self.policy) if (self._fuse_policy and plain_chain and not use_fused) else None
        if fnn is not None and base.name == "bob-d":
            fnn = None
        state0 = st_t.t().clone()
        T_cap = int(max_steps)
        t = 0
        full = self._full
        if use_fused:
            # rollout() with DummyPolicy == vs_step_random: fused steps, on-device uniform actions, lanes freeze at done.
            # Consecutive launches fill ONE device-side trajectory buffer; nothing is copied to the host inside the loop.
            v.set_record_mode(2 if full else 1)
            v.set_traj_capacity(T_cap)
            launches = 0
            # steps per launch: `chunk` (128) for short horizons, up to a quarter of the horizon for long ones -- at 4 000 steps a
            # batch of 65 536 lanes always has a lane that runs to the end, so 32 launches of 128 steps only cost host time
            # (2 of the 8.6 ms of a 65 536-rollout sample_packed() call); lanes that are done are frozen either way
            chunk = int(min(max(self._chunk, T_cap // 8), 1024))
            while t < T_cap:
                k = int(min(chunk, T_cap - t))
                v.set_traj_offset(t)
                v.step_random(k, seed=lane_key ^ 0xA0761D6478BD642F, record=True)
                t += k
                launches += 1
                # one scalar sync per FOUR launches: a wave whose rollouts have all ended leaves the fused kernel at once, so
                # up to three launches too many cost next to nothing, a host round trip per launch does (a third of a
                # 4 096-rollout call)
                if (launches & 3) == 0 and t < T_cap and bool(done_t.bool().all()):
                    break
            v.set_traj_offset(0)
        elif fnn is not None:
            # rollout() with a network policy == vs_step_policy: observation -> network -> (exploration noise) -> step ->
            # record inside ONE kernel, `chunk` steps per launch, lanes freeze at done; the same record planes as above
            if hasattr(self.policy, "reset"):
                self.policy.reset()
            # (eval=True keeps the exploration noise, like every other path here and like the reference: rollout() only calls
            # policy.eval(), and StochasticActionExplStrat.forward samples action_dist_at(act).rsample() whatever the mode,
            # P/exploration/stochastic_action.py:80-96)
            v.set_policy_fnn(obs_idx=None if fc.keep.all() else np.flatnonzero(fc.keep), **fnn)
            v.set_record_mode(2 if full else 1)
            v.set_traj_capacity(T_cap)
            while t < T_cap:
                k = int(min(self._chunk, T_cap - t))
                v.set_traj_offset(t)
                v.step_policy(k, record=True, noise_seed=lane_key ^ 0x8CB92BA72F3D8DD7)
                t += k
                if bool(done_t.bool().all()):  # one scalar sync per launch
                    break
            v.set_traj_offset(0)
        else:
            # policy in the loop: rollout() with the caller's policy (rollout.py:185-258).  One recording step kernel per env
            # step -- vs_step_record writes the observation the policy saw, its action, the reward, the done bit and (full
            # records) state / applied action / hidden state into the same device-side planes the fused path fills -- so a
            # step costs the policy's own kernels plus two launches (observation transpose, step) and nothing is cloned.
            policy = self.policy.to(dev) if hasattr(self.policy, "to") else self.policy
            if hasattr(policy, "eval"):
                policy.eval() if eval else policy.train()
            if hasattr(policy, "reset"):
                policy.reset()
            v.set_record_mode(2 if full else 1)
            v.set_traj_capacity(T_cap)
            v.set_traj_offset(0)
            # rollout() stops stepping an env at done (rollout.py:185): finished lanes are frozen by the step kernel, so
            # nothing the policy makes of their last observation can move them or raise their NaN flag
            v.set_freeze_done(True)
            try:
                if self._graph_policy:
                    # one hipGraph of SEG iterations, replayed until every lane is done: the step kernel takes its record row from a
                    # device-side counter (vs_set_record_row), so a replay continues where the last one stopped
                    SEG = 32
                    v.set_traj_capacity((T_cap + SEG - 1) // SEG * SEG)

                    def one_step():
                        obs_now = visible(obs_full, 0).t().contiguous()
                        act = policy(obs_now).to(torch.float32).reshape(n, A).contiguous()
                        v.step_record(act, row=None)

                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream(v.device))
                    v.use_stream(side.cuda_stream)  # before the capture starts: stream switches synchronise
                    try:
                        with torch.cuda.stream(side), torch.no_grad():
                            v.set_record_row(0)
                            for _ in range(2):  # warm-up of the policy's kernels (library handles, workspaces) ...
                                one_step()
                            side.synchronize()
                            reset_lanes()       # ... undone: the same reset, lane for lane
                            if hasattr(policy, "reset"):
                                policy.reset()
                            state0 = st_t.t().clone()
                            v.set_record_row(0)
                            side.synchronize()
                            graph = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(graph, stream=side):
                                for _ in range(SEG):
                                    one_step()
                            while t < T_cap:
                                graph.replay()
                                t += SEG
                                if bool(done_t.bool().all()):  # one scalar sync per replay
                                    break
                        side.synchronize()
                    finally:
                        v.use_stream(torch.cuda.current_stream(v.device).cuda_stream)
                with torch.no_grad():
                    while t < T_cap and not self._graph_policy:
                        obs_now = visible(obs_full, 0).t().contiguous()  # [n, O']: what the policy sees
                        act = policy(obs_now).to(torch.float32).reshape(n, A).contiguous()
                        v.step_record(act, row=t)
                        t += 1
                        if (t % 32 == 0 or t == T_cap) and bool(done_t.bool().all()):  # one scalar sync per 32 steps
                            break
            finally:
                v.set_freeze_done(False)  # (also when the policy or the capture raises: the cached handle is reused)
        v.raise_on_error()
        # ---- split into rollouts on the device: rollout j = steps 0 .. first done of lane j, packed lane-major.  Lanes freeze
        # at done (all three paths): VS_OBS / VS_STATE / VS_HIDDEN hold every lane's final observation and state.  One kernel
        # (vs_pack_traj) reads only the steps that belong to a rollout from the record planes and writes the packed arrays.
        T = t
        length, done_last_d = v.rollout_lengths(n, T)  # [n]: first done of every lane (or the records' end), from the bit words
        total = int(length.sum())  # the one size-dependent sync
        start = torch.cumsum(length, 0) - length
        pk = v.pack_traj(n, T, length, start, total=total)
        # ONE matrix rows[total + n, F]: rollout j in rows start[j] + j .. start[j] + j + length[j] (its steps, then the entry behind
        # them: final observation / state / hidden state); every field below is a strided view of it
        rows = pk["rows"]
        fields = v.record_fields()
        if packed_out:
            qcp_dev = base.name.startswith("qcp") and H and full
            return PackedRollouts(
                rows=rows, observations=visible(pk["obs"], 1), actions=pk["act"], rewards=pk["rew"],
                states=pk["state"] if full else None, actions_applied=pk["act_app"] if full else None,
                th_ddot=pk["hidden"][:, 0] if qcp_dev else None, lengths=length, offsets=torch.cat([start, start[-1:] + length[-1:]]),
                total=total, done_last=done_last_d, init_states=state0.contiguous(), first_index=first_index,
                env_name=base.name, dt=base.dt, param_names=v.param_names, domain_params=v.tensor(L.VS_PARAMS)[:, :n].t().clone())
        # device -> host: the matrix in one transfer into pinned memory; the rollouts' fields are views of that block (rewards:
        # one conversion to float64 for all rollouts)
        rows_h, done_h, length_h, state0_h = self._to_host([rows, done_last_d.to(torch.uint8), length, state0.contiguous()])
        c_rew = fields["rew"][0]
        rew_p = self._convert(rows_h[:, c_rew], np.float64)
        col = lambda k: rows_h[:, fields[k][0]:fields[k][0] + fields[k][1]]
        obs_p = col("obs") if keep is None else rows_h[:, np.flatnonzero(self._fc.keep)]
        act_p = col("act")
        st_p, app_p = (col("state"), col("act_app")) if full else (None, None)
        hid_p = col("hidden") if (full and H) else None
        done_last = done_h.astype(bool).tolist()
        params = v.get(L.VS_PARAMS)
        lens = length_h.tolist()
        off_o = np.concatenate([[0], np.cumsum(length_h + 1)]).tolist()  # first row of every rollout
        dt, name, pnames = base.dt, base.name, v.param_names
        packed = StepSequence._packed
        qcp = name.startswith("qcp") and hid_p is not None  # the fork's th_ddot field exists for its cartpole only
        # (the cyclic collector off while the list is built: tens of thousands of fresh containers trigger generation after
        # generation of it, a third of this loop at 65 536 rollouts, and nothing built here is garbage)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            ros = [packed(obs_p[off_o[j]:off_o[j + 1]], act_p[off_o[j]:off_o[j] + lens[j]], rew_p[off_o[j]:off_o[j] + lens[j]],
                          (name, pnames, params[j], first_index + j), done_last[j], dt, state0_h[j],
                          None if st_p is None else st_p[off_o[j]:off_o[j + 1]],
                          None if app_p is None else app_p[off_o[j]:off_o[j] + lens[j]],
                          hid_p[off_o[j]:off_o[j + 1], 0] if qcp else None) for j in range(n)]
        finally:
            if gc_was_on:
                gc.enable()
        return ros

    def _convert(self, src, dtype):
        """a (possibly strided) host array as a dense array of another dtype, in chunks on a few threads (NumPy releases the GIL)"""
        dst = np.empty(src.shape, dtype=dtype)
        rows = src.shape[0] if src.ndim else 0
        chunk_rows = 1 << 20
        if rows <= chunk_rows:
            np.copyto(dst, src, casting="unsafe")
            return dst
        pool = self._pool()
        jobs = [pool.submit(np.copyto, dst[a:a + chunk_rows], src[a:a + chunk_rows], "unsafe") for a in range(0, rows, chunk_rows)]
        for j in jobs:
            j.result()
        return dst

    def _pool(self):
        if not hasattr(self, "_copy_pool"):
            from concurrent.futures import ThreadPoolExecutor

            self._copy_pool = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)))))
        return self._copy_pool

    def _to_host(self, tensors, out_dtypes=None):
        """device tensors as NumPy arrays the caller owns.  Every tensor goes into a FRESH pinned host tensor (a pageable .cpu()
        of ~70 MB runs at ~3 GB/s here, a pinned copy at 57 GB/s) and the array handed out is a view of that pinned memory -- no
        second copy on the host (a staging buffer reused by the next call needed one: 7.7 GB of single- and then multi-threaded
        memcpy per 65 536-rollout call, half of a 4 096-rollout call).  The pinned block lives as long as a rollout refers to it
        and then returns to torch's caching host allocator, which serves the next call without another hipHostMalloc.
        out_dtypes[k]: dtype of the k-th result where it differs (rewards -> float64): converted in chunks on a few threads
        (NumPy releases the GIL), as soon as that tensor's own transfer has landed."""
        import torch

        self._pool()
        stream = torch.cuda.current_stream(tensors[0].device)
        staged = []
        for t in tensors:
            pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            pinned.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            staged.append((pinned, ev))
        out, jobs = [], []
        chunk_bytes = 8 << 20
        for k, (pinned, ev) in enumerate(staged):
            src = pinned.numpy()  # (keeps the pinned tensor alive)
            want = out_dtypes[k] if out_dtypes and out_dtypes[k] is not None else src.dtype
            if np.dtype(want) == src.dtype and not self._owned:
                out.append(src)
                continue
            dst = np.empty(src.shape, dtype=want)
            out.append(dst)
            ev.synchronize()
            rows = src.shape[0] if src.ndim else 0
            if rows == 0 or src.nbytes <= chunk_bytes:
                np.copyto(dst, src, casting="unsafe")
                continue
            step = max(1, int(rows * chunk_bytes // src.nbytes))
            for a in range(0, rows, step):
                jobs.append(self._copy_pool.submit(np.copyto, dst[a:a + step], src[a:a + step], "unsafe"))
        for j in jobs:
            j.result()
        stream.synchronize()  # every transfer has landed before the views are handed out
        return out

    def sample(self, init_states: Optional[List[np.ndarray]] = None, domain_params: Optional[List[dict]] = None,
               eval: bool = False) -> List[StepSequence]:
        self._sample_count += 1
        if self.min_steps is None:
            work = self.work_list(init_states, domain_params)
            out = []
            for a in range(0, len(work), self._batch_lanes):
                out += self._run_batch(work[a:a + self._batch_lanes], a, eval, plain=init_states is None and domain_params is None)
            return out
        if init_states is not None:
            raise NotImplementedError  # as in the reference (parallel_rollout_sampler.py:315-316)
        # run_collect: rollouts in index order until min_steps (and min_rollouts) are reached, surplus dropped
        out, steps, idx = [], 0, 0
        guess = max(self.min_rollouts or 1, 1)
        while True:
            nb = int(min(self._batch_lanes, max(guess, 64)))
            batch = self._run_batch([_PLAIN_WORK] * nb, idx, eval, plain=True)
            for ro in batch:
                out.append(ro)
                steps += len(ro)
                if steps >= self.min_steps and len(out) >= (self.min_rollouts or 0):
                    return out
            idx += nb
            guess = max(guess, int(len(out) * (self.min_steps / max(steps, 1) - 1)) + 1)


    def sample_packed(self, init_states: Optional[List[np.ndarray]] = None, domain_params: Optional[List[dict]] = None,
                      eval: bool = False) -> List[PackedRollouts]:
        """sample() without the host side: the same rollouts (same work list, seeds and order) as packed device tensors, one
        `PackedRollouts` per batch of lanes (one in all but very large calls).  No counterpart in the reference, whose workers
        return host `StepSequence`s; `min_rollouts` mode only."""
        if self.min_steps is not None:
            raise ValueErr(msg="sample_packed() runs a fixed number of rollouts: construct the sampler with min_rollouts")
        self._sample_count += 1
        work = self.work_list(init_states, domain_params)
        return [self._run_batch(work[a:a + self._batch_lanes], a, eval, packed_out=True,
                                plain=init_states is None and domain_params is None)
                for a in range(0, len(work), self._batch_lanes)]


def select_cvar(rollouts: list, epsilon: float, gamma: float = 1.0) -> list:
    """The epsilon-fraction of the rollouts with the lowest discounted return, lowest first: their mean return is the
    CVaR(eps) of the set (P/sampling/cvar_sampler.py:40-62).  Sorts `rollouts` in place, like the reference."""
    rollouts.sort(key=lambda ro: ro.discounted_return(gamma))
    keep = round(len(rollouts) * epsilon)
    if keep == 0:
        raise ValueErr(given=keep, g_constraint="0")
    return rollouts[:keep]


class CVaRSampler:
    """Samples 1 / epsilon times as many rollouts with the wrapped sampler and keeps the worst epsilon-quantile (EPOpt;
    P/sampling/cvar_sampler.py:65-140).  `full_stats` holds what the reference logs about the full set."""

    def __init__(self, wrapped_sampler, epsilon: float, gamma: float = 1.0, *, min_rollouts: int = None,
                 min_steps: int = None):
        if not 0 < epsilon <= 1:
            raise ValueErr(given=epsilon, g_constraint="0", le_constraint="1")
        self._wrapped_sampler = wrapped_sampler
        self.epsilon, self.gamma = epsilon, gamma
        self.full_stats = {}
        self.set_min_count(min_rollouts=min_rollouts, min_steps=min_steps)

    def set_min_count(self, min_rollouts=None, min_steps=None):
        if min_rollouts is None and min_steps is None:
            raise ValueErr(msg="At least one of min_rollouts and min_steps must be given")
        self.min_rollouts, self.min_steps = min_rollouts, min_steps
        grow = lambda v: None if v is None else int(v / self.epsilon)  # the (1 - eps) quantile is thrown away
        self._wrapped_sampler.set_min_count(min_rollouts=grow(min_rollouts), min_steps=grow(min_steps))

    def reinit(self, env=None, policy=None):
        self._wrapped_sampler.reinit(env=env, policy=policy)

    def sample(self) -> list:
        full = self._wrapped_sampler.sample()
        rets = np.array([ro.undiscounted_return() for ro in full])
        self.full_stats = {"full avg rollout len": float(np.mean([ro.length for ro in full])),
                           "full avg return": float(rets.mean()), "full median return": float(np.median(rets)),
                           "full std return": float(rets.std())}
        return select_cvar(full, self.epsilon, self.gamma)
