#!/usr/bin/env python3
"""
bench.py -- env-steps/sec of the MI355X-native stepper on BASELINE.json's metric:
"env-steps/sec whole node, 65 536 QQubeSwingUpSim envs, random policy".

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 the command starts its own N ranks (one process per GPU, `python -m torch.distributed.run`, RCCL) and
relays rank 0's JSON line -- like the reference's sampler, which owns its worker pool (P/sampling/sampler_pool.py:98-116,
parallel_rollout_sampler.py:216-230).  The parent never touches the GPU.  Under a launcher that has already set
RANK / WORLD_SIZE (the driver's torchrun line) the process is a rank itself.

A "step" is one pass of the hot path over the batch = one launch.  In the default fused mode a launch advances every one of
the 65 536 environments of a rank by `--chunk` (400) env steps -- SimPyEnv.step: reward -> clip -> dead zone -> integrate
-> done -> observe, finished lanes auto-reset in the same kernel, actions from the on-device uniform random policy
(DummyPolicy), obs/act/rew/done of every env step recorded; in `--mode step` a launch is one env step.  `value` counts ENV
steps: envs x env-steps-per-launch x K / time ("env_steps_per_step" in the JSON).  State and constants are resident in HBM
when the timed region starts.  Each rank owns 65 536 envs on its own GPU (weak scaling, the batch shards embarrassingly);
the only collective is an RCCL all-gather of completed-episode return statistics at the end.

The records of consecutive launches rotate through a buffer of more than 1 GiB (several times the 256 MiB Infinity Cache),
so the record stream is a real HBM stream.

The JSON line also carries the roofline of the dominant kernel (HIP-event timed on the kernel's stream), the single-step
kernel at 16.7 M envs (the HBM-bound point), the full-record (record mode 2) rate, and the CPU baseline (the oracle's NumPy
port timed on this box's host cores, bounded sample).

BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend (a one-GPU rehearsal of the multi-rank path).
BENCH_DRYRUN=1: no device work at all -- only the launch / rendezvous / collective / relay plumbing (CPU tests).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "env-steps/sec whole node, 65 536 QQubeSwingUpSim envs, random policy"
# ALGORITHMIC bytes per env-step, fp32 SoA:
#   single-step kernel: 4*[(S+A+P+H+1) read + (S+H+O+1+1) written] + 1 done byte
#   fused rollout kernel with records: per step only the records leave the chip -- mode 1: 4*(O + A + 1) floats + 1 done
#   bit; mode 2: 4*(O + A + 1 + S + A + H) -- state/constants are read and written once per launch (amortised over `chunk`)
DIMS = {"omo": dict(S=2, A=1, O=2, P=3, H=0), "bob": dict(S=4, A=1, O=4, P=8, H=0), "qq-su": dict(S=4, A=1, O=6, P=11, H=0),
        "qcp-su": dict(S=4, A=1, O=5, P=17, H=1), "qbb": dict(S=8, A=2, O=8, P=20, H=2),
        "qq-st": dict(S=4, A=1, O=6, P=11, H=0), "qcp-st": dict(S=4, A=1, O=5, P=17, H=1),
        "pend": dict(S=2, A=1, O=3, P=5, H=0), "bob-d": dict(S=4, A=1, O=4, P=8, H=0)}
ENV_KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
          "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500),
          "qq-st": dict(dt=0.01, max_steps=500), "qcp-st": dict(dt=0.01, max_steps=300),
          "pend": dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2])), "bob-d": dict(dt=0.01, max_steps=500)}
ACT_HI = {"omo": 30.0, "bob": 29.43, "qq-su": 4.5, "qcp-su": 6.0, "qbb": 3.0, "qq-st": 4.5, "qcp-st": 6.0, "pend": 3.5,
          "bob-d": 29.43}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6290.0  # what a float4 copy reaches there (79 % of spec)
# vector-ALU issue peak: 1 024 SIMDs x one wave64 fp32 instruction per 2 cycles (SIMD-32) x 2.4 GHz (same guide: `v_fma_f32`
# 2 cyc throughput per SIMD, 4 cyc issue for one wave alone; transcendentals 8, integer multiplies quarter rate -- a kernel that
# mixes those in cannot reach it, and a SIMD with a single wave stops at half of it)
VALU_PEAK_GINST_S = 1024 * 2.4 / 2.0 * 1e3 / 1e3  # 1 228.8 G wave-instructions / s
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix rate (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD = the fp32 vector rate), same guide
DEFAULT_CHUNK = 400  # env steps per launch: the fixed cost of a launch (dispatch, pipeline fill and drain: ~6 us) is 3 % of it
RECORD_BUFFER_BYTES = 1.25 * 2 ** 30  # rotating record buffer: > 1 GiB, several times the 256 MiB Infinity Cache


def bytes_single_step(d):
    return 4 * ((d["S"] + d["A"] + d["P"] + d["H"] + 1) + (d["S"] + d["H"] + d["O"] + 1 + 1)) + 1


def bytes_fused_step(d, chunk, record):
    per_launch = 4 * ((d["S"] + d["P"] + d["H"] + 1) + (d["S"] + d["H"] + d["O"] + 1 + 1)) + 1  # state/consts in+out once
    if record == 2:
        per_step = 4 * (d["O"] + d["A"] + 1 + d["S"] + d["A"] + d["H"]) + 0.125
    elif record:
        per_step = 4 * (d["O"] + d["A"] + 1) + 0.125  # done flags are one bit per env and step
    else:
        per_step = 0
    return per_step + per_launch / chunk


# ---------------------------------------------------------------------------------------------------------- CPU baseline
def usable_cores():
    """cores this job may use: its CPU affinity, capped by the cgroup CPU quota (a GPU box gives one job a share of its
    host cores without necessarily narrowing the affinity mask)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(np.ceil(float(quota) / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _cpu_vector_worker(args):
    """one core: the oracle (NumPy fp64, vectorised over its share of the envs) stepping with auto-reset for budget_s"""
    env_name, n_envs, budget_s, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import cpu_ref

    kw = ENV_KW[env_name]
    ref = cpu_ref.make_ref(env_name, **kw)
    rng = np.random.default_rng(seed)
    params = ref.nominal_params(n_envs)
    lo, hi = ref.init_bounds(params) if not env_name.startswith("bob") else ref.init_bounds(params, 0)

    def fresh_states():
        st = rng.uniform(lo, hi)
        return ref.state_from_init(ref.polar_to_init(st)) if env_name == "qbb" else st

    state = fresh_states()
    hidden = np.zeros((n_envs, ref.H))
    _, _, alo, ahi = ref.bounds(params)
    steps = np.zeros(n_envs, dtype=np.int64)
    t0 = time.perf_counter()
    n_steps = 0
    while time.perf_counter() - t0 < budget_s:
        out = ref.step(state, hidden, rng.uniform(alo, ahi), params, steps)
        state, hidden, steps = out["state"], out["hidden"], out["curr_step"]
        d = out["done"]
        if d.any():  # auto-reset, as on the device
            state[d] = fresh_states()[d]
            hidden[d] = 0
            steps[d] = 0
        n_steps += 1
    el = time.perf_counter() - t0
    return n_envs * n_steps / el, n_steps, el


def _cpu_scalar_worker(args):
    """one core, the shape of one worker of the reference's ParallelRolloutSampler (P/sampling/sampler_pool.py:392-469,
    rollout.py:137-311): ONE env object stepped one step at a time by a uniform random policy, rollout after rollout
    (reset -> step until done or max_steps) until min_steps steps are collected"""
    env_name, min_steps, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import cpu_ref

    kw = ENV_KW[env_name]
    ref = cpu_ref.make_ref(env_name, **kw)
    rng = np.random.default_rng(1000 + seed)
    params = ref.nominal_params(1)
    lo, hi = ref.init_bounds(params) if not env_name.startswith("bob") else ref.init_bounds(params, 0)
    _, _, alo, ahi = ref.bounds(params)
    collected, rollouts = 0, 0
    t0 = time.perf_counter()
    while collected < min_steps:
        st = rng.uniform(lo, hi)
        state = ref.state_from_init(ref.polar_to_init(st)) if env_name == "qbb" else st
        hidden = np.zeros((1, ref.H))
        steps = np.zeros(1, dtype=np.int64)
        done = False
        while not done and steps[0] < ref.max_steps:
            out = ref.step(state, hidden, rng.uniform(alo, ahi), params, steps)
            state, hidden, steps = out["state"], out["hidden"], out["curr_step"]
            done = bool(out["done"][0])
            collected += 1
        rollouts += 1
    return collected, rollouts, time.perf_counter() - t0


def cpu_baseline(env_name, n_envs, budget_s=None, scalar_min_steps=None):
    """CPU baseline beside the GPU number, on every core this job may use, BEFORE the GPU is touched
    (worker processes are forked from a process that has not initialised HIP):
      value             the oracle's NumPy port vectorised over the same 65 536-env workload, one process per core
      scalar_all_cores  the oracle stepped ONE env at a time per worker process -- how the reference itself is driven --
                        until every worker has collected `scalar_min_steps` steps of complete rollouts"""
    import multiprocessing as mp

    # (BENCH_CPU_BUDGET_S / BENCH_CPU_SCALAR_STEPS: the CPU tests shrink the sample; the defaults are ~10 s of CPU work)
    budget_s = float(os.environ.get("BENCH_CPU_BUDGET_S", "6.0")) if budget_s is None else budget_s
    scalar_min_steps = int(os.environ.get("BENCH_CPU_SCALAR_STEPS", "100000")) if scalar_min_steps is None else scalar_min_steps
    floor_steps = int(os.environ.get("BENCH_CPU_SCALAR_FLOOR", "20000"))
    cores = usable_cores()
    share = max(1, n_envs // cores)
    # min_steps = 1e5 for the sampler as a whole; at least 20 000 per worker so that every worker times a few
    # dozen complete rollouts
    scalar_min_steps = max(int(np.ceil(scalar_min_steps / cores)), floor_steps)
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_vector_worker, [(env_name, share, budget_s, r) for r in range(cores)])
    total = float(sum(r[0] for r in res))
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        sc = pool.map(_cpu_scalar_worker, [(env_name, scalar_min_steps, r) for r in range(cores)])
        wall = time.perf_counter() - t0
    sc_steps = int(sum(r[0] for r in sc))
    scalar_rate = sc_steps / wall
    return dict(value=total, unit="env-steps/s", cores=cores, kind="port",
                sample=f"oracle/cpu_ref.py (NumPy fp64) in {cores} processes x {share} envs, vectorised, {res[0][1]} batch steps "
                       f"per process in {res[0][2]:.1f} s incl. auto-reset; one process alone: {res[0][0]:.3g} env-steps/s; "
                       f"scalar_all_cores: {cores} processes x one env stepped one step at a time, complete rollouts until "
                       f"{scalar_min_steps} steps per process ({sc_steps} steps, {sum(r[1] for r in sc)} rollouts in {wall:.1f} s); "
                       f"reference Pyrado itself: 2.1-2.6e3 env-steps/s/core (BASELINE.md, measured in the build container)",
                single_process_value=float(res[0][0]),
                scalar_all_cores=dict(value=scalar_rate, unit="env-steps/s", cores=cores, min_steps_per_worker=scalar_min_steps,
                                      steps=sc_steps, wall_s=wall, per_core=scalar_rate / cores),
                host_cores=os.cpu_count(), affinity_cores=len(os.sched_getaffinity(0)))


# ---------------------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, args):
    """start n ranks of this very command as fresh child processes (this parent has made no GPU call and makes none),
    relay their output, exit with their code.  The CPU baseline of the line is timed HERE, on the box's host cores before any
    rank exists (north_star: "next to the reference ParallelRolloutSampler timed on the box's host cores ... in the same run"),
    and handed to rank 0 through a file."""
    import tempfile

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    tmp = None
    if not args.no_cpu_baseline:
        base = cpu_baseline(args.env, min(args.envs, 65536))
        base["timed_by"] = "launcher parent, before the ranks were started"
        fd, tmp = tempfile.mkstemp(prefix="bench_cpu_baseline_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(base, f)
        env["BENCH_CPU_BASELINE_JSON"] = tmp
    try:
        return subprocess.run(cmd, env=env).returncode
    finally:
        if tmp and os.path.exists(tmp):
            os.remove(tmp)


# ---------------------------------------------------------------------------------------------------------- one rank
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed launches (default 1000 fused / 2000 step mode)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed launches (default 50 fused / 200 step mode)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--env", default="qq-su", choices=sorted(DIMS))
    ap.add_argument("--mode", default="fused", choices=["fused", "step"])
    ap.add_argument("--chunk", type=int, default=DEFAULT_CHUNK, help="env steps per launch in fused mode")
    ap.add_argument("--record", type=int, default=1, help="0 none, 1 obs|act|rew, 2 + state|act_app|hidden")
    ap.add_argument("--per-env-params", type=int, default=1, help="1: per-env constants [K][N] (DR-capable), 0: broadcast")
    ap.add_argument("--live-dr", type=int, default=0, help="DomainRandWrapperLive on the device: redraw the first K parameters "
                    "of the family's default randomizer at every reset (BASELINE config 3: qcp-su with K = 7)")
    ap.add_argument("--graph", type=int, default=0, help="step mode: capture `chunk` policy+step iterations in one hipGraph")
    ap.add_argument("--lean-step", type=int, default=0, help="step mode: vs_set_lean_step (no running return, no failed byte: the "
                    "117-B model of SURVEY 8(d) exactly; episode returns then read 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the large-N / full-record / probe legs (quick sweeps)")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def measure(args, rank, world, local_rank, dist, rehearsal):
    """the timed region of one rank; returns what rank 0 needs for the JSON line"""
    import torch

    import simurlacra_amd as vs
    from simurlacra_amd import _lib as L
    from simurlacra_amd.dist import shard

    kw = ENV_KW[args.env]
    d = DIMS[args.env]
    n = args.envs
    env = vs.VecSimEnv(args.env, n, device=local_rank, **kw)
    if args.per_env_params:
        env.set_params(np.tile(vs.nominal_params(args.env), (n, 1)))
    if args.live_dr:
        rz = vs.create_default_randomizer(vs.ENV_CLASSES[args.env](**kw))
        env.set_randomizer(rz.device_specs()[: args.live_dr])
    first, _ = shard(n * world, rank, world)
    env.set_index_offset(first)  # global env index: lane streams do not depend on the number of GPUs
    env.set_auto_reset(True, seed=args.seed * 1000 + 1)
    if args.lean_step:
        env.set_lean_step(True)
    env.reset(seed=args.seed * 7919 + 2)
    chunk = max(1, args.chunk)
    steps = args.steps
    slots = 1
    if args.mode == "fused" and args.record:
        env.set_record_mode(args.record)
        slot_bytes = chunk * env.traj_layout()[0] * env.ld * 4
        slots = max(1, int(np.ceil(RECORD_BUFFER_BYTES / slot_bytes)))
        env.set_traj_capacity(chunk * slots)
    graph = None

    def policy_and_step():
        act = (torch.rand(n, d["A"], device=f"cuda:{local_rank}") * 2 - 1) * ACT_HI[args.env]  # DummyPolicy on the GPU
        env.step(act)

    launches = [0]

    def run(k_launches):
        if args.mode == "fused":
            rec = bool(args.record)
            for _ in range(k_launches):
                if rec:
                    env.set_traj_offset((launches[0] % slots) * chunk)
                    launches[0] += 1
                env.step_random(chunk, seed=args.seed + 3, record=rec)
        elif graph is not None:
            for _ in range(k_launches):
                graph.replay()
        else:
            for _ in range(k_launches):
                policy_and_step()

    if args.mode == "step":
        env.use_stream(torch.cuda.current_stream().cuda_stream)
        if args.graph:
            # launch-bound inner loop -> one hipGraph of `chunk` (policy, vs_step) pairs; vs_step takes no host-side
            # counter, so replaying the captured launches is exact
            side = torch.cuda.Stream()
            env.use_stream(side.cuda_stream)  # before the capture starts: stream switches synchronise
            with torch.cuda.stream(side):
                for _ in range(3):
                    policy_and_step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(chunk):
                    policy_and_step()

    ev_ms = []

    def timed(k):
        env.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.mode == "fused":
            env.timer_start()  # HIP events on the stream the kernels are launched on, around the very same launches
        run(k)
        if args.mode == "fused":
            ev_ms.append(env.timer_stop())
        env.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # pre-roll (untimed, before the W warm-up launches): the GPU comes out of the host-side set-up idle and its clock takes
    # some 10 ms of work to ramp -- a 20-launch timed region (1 ms) right behind a 5-launch warm-up would measure the ramp
    preroll = int(os.environ.get("BENCH_PREROLL", "400")) if args.mode == "fused" else 0
    if preroll:
        run(preroll)
    run(max(args.warmup, 1))
    el = timed(steps)  # THE timed region: exactly `steps` launches between two barriers + synchronisations
    repeats = [timed(steps) for _ in range(3)]  # untimed by the contract: run-to-run spread of the same region
    cnt_t, rs_t, ls_t = (env.tensor(w)[0, :n] for w in (L.VS_EPSTAT_COUNT, L.VS_EPSTAT_RETSUM, L.VS_EPSTAT_LENSUM))
    return dict(env=env, el=el, repeats=repeats, stats=(cnt_t, rs_t, ls_t), slots=slots, graph=graph is not None, preroll=preroll,
                kernel_ms_timed_region=(ev_ms[0] / steps) if ev_ms else None,
                kernel_ms_repeats=[e / steps for e in ev_ms[1:]])


def main():
    args = parse_args()
    dryrun = os.environ.get("BENCH_DRYRUN") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args))  # the ranks are children; this process never initialises the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        args.gpus = world
    if args.steps is None:
        args.steps = 1000 if args.mode == "fused" else 2000
    if args.warmup is None:
        args.warmup = 50 if args.mode == "fused" else 200
    # the CPU baseline beside the GPU number, at every N: from the launcher parent's file when this command started its own
    # ranks, else timed by rank 0 right here -- before its first HIP call (forked workers) and before the rendezvous, the
    # other ranks wait for it in init_process_group
    cpu_base = None
    if not args.no_cpu_baseline and rank == 0:
        handed = os.environ.get("BENCH_CPU_BASELINE_JSON")
        if handed and os.path.exists(handed):
            cpu_base = json.load(open(handed))
        else:
            cpu_base = cpu_baseline(args.env, min(args.envs, 65536))
            cpu_base["timed_by"] = "rank 0, before its first GPU call"

    import torch

    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- a 1-GPU rehearsal of the multi-rank code path
    # (the real runs use one GPU per rank and RCCL)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1" or dryrun
    if not dryrun:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)

    from simurlacra_amd.dist import gather_episode_stats

    n, chunk = args.envs, max(1, args.chunk)
    per_step = chunk if (args.mode == "fused" or args.graph) else 1  # env steps per launch
    if dryrun:
        # plumbing only: no device, no stepping -- the launch, the rendezvous, the collectives and the relay of rank 0's line
        if dist:
            dist.barrier()
        m = dict(env=None, el=float("nan"), repeats=[], slots=0, graph=False, preroll=0,
                 stats=tuple(torch.full((4,), float(rank + 1)) for _ in range(3)))
    else:
        m = measure(args, rank, world, local_rank, dist, rehearsal)
    env, el_own = m["env"], m["el"]
    # gather completed-episode return statistics over RCCL (the only collective of this path): per-env accumulators
    # reduced on the device, three doubles per rank on the wire
    if not dryrun:
        torch.cuda.synchronize()
    t_c = time.perf_counter()
    ep = gather_episode_stats(*m["stats"])
    if not dryrun:
        torch.cuda.synchronize()
    collective_ms = (time.perf_counter() - t_c) * 1e3
    dev_t = "cpu" if rehearsal else f"cuda:{local_rank}"
    el_all = torch.tensor([el_own], device=dev_t, dtype=torch.float64)
    if dist:
        parts = [torch.zeros_like(el_all) for _ in range(world)]
        dist.all_gather(parts, el_all)
        el_ranks = [float(p.item()) for p in parts]
    else:
        el_ranks = [el_own]
    el = max(el_ranks)  # MAX over ranks
    errs = env.error_count() if env is not None else 0

    if rank == 0:
        steps = args.steps
        total_env_steps = float(n) * steps * per_step * world
        kw = ENV_KW[args.env]
        d = DIMS[args.env]
        out = {
            "metric": METRIC,
            "value": None if dryrun else total_env_steps / el, "unit": "env-steps/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup,
            # untimed launches BEFORE the W warm-up launches (clock ramp after the host-side set-up; BENCH_PREROLL): no work of
            # the timed region is skipped or cached by them, they are simply more warm-up
            "preroll": m["preroll"],
            "ms_per_step": el / steps * 1e3, "env_steps_per_step": per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_region_s": el,
            "repeat_ms_per_step": [r / steps * 1e3 for r in m["repeats"]],
            # evidence of the multi-rank path: how many ranks answered the all-gathers, what the return gather cost
            "ranks_seen": int(ep["per_rank"].shape[0]), "collective_ms": collective_ms,
            "collective": ("none (single rank)" if world == 1 else
                           ("gloo all_gather (rehearsal)" if rehearsal else "RCCL all_gather of 3 doubles per rank")),
            "per_rank_env_steps_per_s": [float(n) * steps * per_step / e for e in el_ranks],
            "config": {"workload": f"{args.env} x {n} envs per GPU, dt {kw['dt']}, max_steps {kw['max_steps']}, uniform random "
                                   f"policy on device, auto-reset, mode={args.mode}"
                                   + (f", {chunk} steps/launch, record={args.record}, records rotating through "
                                      f"{m['slots']} x {chunk} rows" if args.mode == "fused" else "")
                                   + (f", hipGraph of {chunk} (policy, step) pairs" if m["graph"] else "")
                                   + (", per-env constants" if args.per_env_params else ", broadcast constants")
                                   + (f", live domain randomisation of {args.live_dr} parameters at every reset" if args.live_dr else ""),
                       "envs_per_gpu": n, "env": args.env, "mode": args.mode, "chunk": chunk, "record": args.record,
                       "parallelism": f"env-shard x{world}"},
            "episodes": {"completed": ep["episodes"], "mean_return": ep["mean_return"], "mean_length": ep["mean_length"]},
            "nan_flags": errs,
        }
        if dryrun:
            out["dry_run"] = True
        else:
            out["roofline"] = roofline(args, env, local_rank, d, n, chunk, m.get("kernel_ms_timed_region"))
            out["roofline"]["kernel_ms_repeats"] = m.get("kernel_ms_repeats")
        out["cpu_baseline"] = cpu_base
        if cpu_base and out["value"]:
            out["vs_cpu_baseline"] = out["value"] / cpu_base["value"]
        print(json.dumps(out), flush=True)
    if env is not None:
        env.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


# BASELINE.json's other GPU configurations, one rank's shard each, as legs of the default line:
# the same fused launch as the headline (400 env steps per launch, every step recorded, per-env constants, auto-reset, records
# rotating through > 1 GiB), timed with HIP events on the kernel's stream.
BASELINE_CONFIGS = [
    dict(key="config2", what="QQubeSwingUpSim RK4, 4 096 envs, random policy", members=[("qq-su", 4096)], live_dr=0),
    dict(key="config3", what="QCartPoleSwingUpSim + DomainRandWrapperLive (7 randomised params), 65 536 envs",
         members=[("qcp-su", 65536)], live_dr=7),
    dict(key="config4", what="QBallBalancerSim, 262 144 envs over 8 GPUs: one rank's 32 768", members=[("qbb", 32768)], live_dr=0),
    dict(key="config5", what="mixed QQube + QCartPole + BallOnBeam, 1 M envs over 8 GPUs: one rank's 130 560 (43 520 each), one launch",
         members=[("qq-su", 43520), ("qcp-su", 43520), ("bob", 43520)], live_dr=0),
]


def load_counters():
    """profiles/rNN_counters.json: per-kernel PMC counters of this very command under rocprofv3 (separate --pmc passes),
    committed once per round with the library version they are of; None when there is none for the loaded library"""
    import glob

    from simurlacra_amd import _lib as L

    try:
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))[-1]
        tab = json.load(open(tf))
        if tab.get("lib_version") == int(L.load().vs_version()):
            tab["_source"] = os.path.basename(tf)
            return tab
    except Exception:
        pass
    return None


def valu_leg(cnt, kernel_ms):
    """the vector-issue side of a leg's roofline from the PMC counters of its kernel (SQ_INSTS_VALU per launch, collected under
    rocprofv3 in a separate pass of this command) and the launch duration measured live: for the kernels that are not HBM-bound"""
    try:
        per64 = cnt["derived"]["valu_insts_per_64_env_steps"]
        insts = per64 * cnt["env_steps_per_launch"] / 64.0
        ach = insts / (kernel_ms * 1e-3) / 1e9
        out = {"bound": "valu-issue", "valu_insts_per_64_env_steps": per64, "achieved": ach, "peak": VALU_PEAK_GINST_S,
               "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_GINST_S}
        for k in ("salu_insts_per_64_env_steps", "lds_insts_per_64_env_steps", "valu_active_share_of_wave_cycles",
                  "any_inst_active_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "wait_inst_any_share_of_wave_cycles"):
            if k in cnt["derived"]:
                out[k] = cnt["derived"][k]
        return out
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def counters_of(counters, key, kernel_ms):
    if not counters or key not in counters:
        return None
    c = counters[key]
    out = {"source": counters["_source"], "kernel": c.get("kernel")}
    if "traffic_bytes_per_launch" in c:
        out["traffic"] = c["traffic_bytes_per_launch"]
        out["traffic_over_algorithmic"] = c["traffic_over_algorithmic"]
    v = valu_leg(c, kernel_ms)
    if v:
        out["valu"] = v
    return out


def baseline_config_legs(local_rank, chunk, iters=100):
    import simurlacra_amd as vs

    counters = load_counters()
    preroll = int(os.environ.get("BENCH_PREROLL", "400"))
    legs = {}
    for cfg in BASELINE_CONFIGS:
        try:
            envs = []
            for name, n in cfg["members"]:
                kw = ENV_KW[name]
                e = vs.VecSimEnv(name, n, device=local_rank, **kw)
                e.set_params(np.tile(vs.nominal_params(name), (n, 1)))
                if cfg["live_dr"]:
                    rz = vs.create_default_randomizer(vs.ENV_CLASSES[name](**kw))
                    e.set_randomizer(rz.device_specs()[: cfg["live_dr"]])
                e.set_auto_reset(True, seed=11)
                e.reset(seed=12)
                e.set_record_mode(1)
                slot_bytes = chunk * e.traj_layout()[0] * e.ld * 4
                slots = max(1, min(8, int(np.ceil(RECORD_BUFFER_BYTES / len(cfg["members"]) / slot_bytes))))
                e.set_traj_capacity(chunk * slots)
                envs.append((e, slots))
            n_tot = sum(n for _, n in cfg["members"])
            b_per = sum(bytes_fused_step(DIMS[name], chunk, 1) * n for name, n in cfg["members"]) / n_tot
            if len(envs) == 1:
                e, slots = envs[0]
                launches = [0]

                def go(k):
                    for _ in range(k):
                        e.set_traj_offset((launches[0] % slots) * chunk)
                        launches[0] += 1
                        e.step_random(chunk, seed=13, record=True)

                go(preroll + 30)  # (untimed: the headline's pre-roll -- the episodes of the batch out of phase with the common reset -- and warm-up)
                e.sync()
                e.timer_start()
                go(iters)
                ms = e.timer_stop() / iters
                kname = e.rollout_variant()
            else:
                mixed = vs.MixedVecSimEnv([e for e, _ in envs])
                ms = mixed.time_random(chunk, record=True, iters=iters)
                kname = "k_rollout_mixed"
            units = n_tot * chunk
            ach = b_per * units / (ms * 1e-3) / 1e9
            leg = {"config": cfg["what"], "kernel": kname, "envs": n_tot, "env_steps_per_launch": units, "kernel_ms": ms,
                   "env_steps_per_s": units / (ms * 1e-3), "alg_bytes_per_env_step": b_per, "achieved": ach, "unit": "GB/s",
                   "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS, "preroll": preroll if len(envs) == 1 else 0, "warmup": 30, "launches": iters}
            cn = counters_of(counters, cfg["key"], ms)
            if cn:
                leg["counters"] = cn
            legs[cfg["key"]] = leg
            if len(envs) > 1:
                mixed.close()
            for e, _ in envs:
                e.close()
        except Exception as exc:  # the headline must not depend on these legs
            legs[cfg["key"]] = {"config": cfg["what"], "error": repr(exc)}
    return legs


def roofline(args, env, local_rank, d, n, chunk, ms_region=None):
    """roofline of the dominant kernel (HIP events on the kernel's own stream) + the reference points beside it"""
    import ctypes

    import torch

    import simurlacra_amd as vs
    from simurlacra_amd import _lib as L

    if args.mode == "fused":
        # average launch duration of the dominant kernel: HIP events on its stream over the timed region itself (the K
        # launches `value` is computed from); a separate back-to-back sample after the idle gap of the host-side epilogue
        # reads up to 10 % high (clock ramp) and is kept only as a fallback
        ms = ms_region if ms_region else env.time_step_kernel(iters=500, k_steps=chunk, record=bool(args.record))
        b_per = bytes_fused_step(d, chunk, args.record)
        units = n * chunk
        kname = env.rollout_variant()
    else:
        act = (torch.rand(n, d["A"], device=f"cuda:{local_rank}") * 2 - 1) * ACT_HI[args.env]
        torch.cuda.synchronize()
        ms = env.time_step_kernel(iters=200, actions=act)
        b_per = bytes_single_step(d)
        units = n
        kname = "k_step"
    achieved = b_per * units / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "achievable_GBs": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
            "alg_bytes_per_launch": b_per * units, "kernel": kname, "kernel_ms": ms, "alg_bytes_per_env_step": b_per,
            "env_steps_per_launch": units, "traffic": None, "traffic_unit": "bytes/launch (PMC)", "traffic_source": None}
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes of this very command;
    # gfx950 correction applied) -- measured once per round, committed under profiles/ with the library version it is of
    try:
        import glob

        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        tab = json.load(open(tf))
        key = None
        if (args.mode == "fused" and args.env == "qq-su" and n == 65536 and args.record == 1 and args.per_env_params == 1
                and chunk == tab.get("fused_default", {}).get("chunk")):
            key = "fused_default"
        elif args.mode == "step" and args.env == "qq-su" and n == 16777216:
            key = "step_16m"
        if key and tab.get("lib_version") == int(L.load().vs_version()):
            roof["traffic"], roof["traffic_source"] = tab[key]["traffic_bytes_per_launch"], os.path.basename(tf)
    except Exception:
        pass
    if args.no_extras or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return roof  # the reference points and the extra legs are measured at N = 1 only: the other ranks would wait
    # the practical HBM ceiling next to the nominal one -- in-repo float4 copy / fill kernels (1 GiB per pass,
    # far beyond the caches, 4 independent 16-B accesses per thread, non-temporal), timed with HIP events
    g = ctypes.c_float()
    lib = L.load()
    if lib.vs_membw_probe(local_rank, 1 << 30, 10, ctypes.byref(g)) == 0:
        roof["copy_kernel_GBs"] = float(g.value)
        roof["frac_of_copy_kernel"] = achieved / float(g.value)
    if lib.vs_memwrite_probe(local_rank, 1 << 30, 10, ctypes.byref(g)) == 0:
        roof["write_kernel_GBs"] = float(g.value)  # a pure write stream: what the record stores compete with
        roof["frac_of_write_kernel"] = achieved / float(g.value)
    if args.mode == "fused" and args.record == 1:
        # the same launch with the full records of rollout() (state, applied action, hidden state): its own byte model
        env.set_record_mode(2)
        slot_bytes = chunk * env.traj_layout()[0] * env.ld * 4
        env.set_traj_capacity(chunk * max(1, int(np.ceil(RECORD_BUFFER_BYTES / slot_bytes))))
        ms2 = env.time_step_kernel(iters=300, k_steps=chunk, record=True)
        b2 = bytes_fused_step(d, chunk, 2)
        roof["record2"] = {"kernel": env.rollout_variant(), "kernel_ms": ms2, "env_steps_per_s": units / (ms2 * 1e-3),
                           "alg_bytes_per_env_step": b2, "achieved": b2 * units / (ms2 * 1e-3) / 1e9,
                           "frac": b2 * units / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # ... and the split of those records into rollouts (vs_pack_traj: time-major planes -> rollout-major packed arrays, what
        # rollout() / StepSequence.concat hand to an algorithm): every lane one rollout of `chunk` steps
        try:
            env.set_traj_offset(0)
            lengths = torch.full((n,), chunk, device=f"cuda:{local_rank}", dtype=torch.int64)
            starts = torch.cumsum(lengths, 0) - lengths
            for _ in range(2):
                pk = env.pack_traj(n, chunk, lengths, starts, total=n * chunk)
            env.sync()
            env.timer_start()
            for _ in range(5):
                pk = env.pack_traj(n, chunk, lengths, starts, total=n * chunk)
            msk = env.timer_stop() / 5
            bk = 2 * 4 * env.traj_layout()[0]  # every recorded float read once (time-major planes) and written once (row matrix)
            roof["pack_traj"] = {"kernel": "k_pack_traj", "recorded_steps": units, "kernel_ms": msk, "alg_bytes_per_recorded_step": bk,
                                 "achieved": bk * units / (msk * 1e-3) / 1e9, "frac": bk * units / (msk * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del pk
        except Exception as exc:  # the headline must not depend on this leg
            roof["pack_traj"] = {"error": repr(exc)}
        env.set_record_mode(1)
    if args.mode == "fused" and args.env == "qq-su":
        # the HBM-bound point of this path: one vs_step launch over 16 777 216 envs, 117 algorithmic bytes
        # per env step, far beyond every cache
        try:
            big_n = 16777216
            big = vs.VecSimEnv(args.env, big_n, device=local_rank, **ENV_KW[args.env])
            big.set_params(np.tile(vs.nominal_params(args.env), (big_n, 1)))
            big.set_auto_reset(True, seed=5)
            big.set_lean_step(True)  # what SimPyEnv.step returns and nothing else: the 117-B model of SURVEY 8(d) exactly
            big.reset(seed=6)
            act = (torch.rand(d["A"], big.ld, device=f"cuda:{local_rank}") * 2 - 1) * ACT_HI[args.env]
            torch.cuda.synchronize()
            msb = big.time_step_kernel(iters=8, actions=act)
            bb = bytes_single_step(d)
            roof["large_n"] = {"kernel": "k_step", "envs": big_n, "kernel_ms": msb, "alg_bytes_per_env_step": bb, "lean_step": True,
                               "env_steps_per_s": big_n / (msb * 1e-3), "achieved": bb * big_n / (msb * 1e-3) / 1e9,
                               "frac": bb * big_n / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "frac_of_achievable": bb * big_n / (msb * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS}
            big.close()
            del act
        except Exception as exc:  # the headline must not depend on this leg
            roof["large_n"] = {"error": repr(exc)}
    if args.mode == "fused" and args.env == "qq-su":
        # the policy-in-the-loop point of the path (rollout.py:203-219 with an FNNPolicy): vs_step_policy, the network evaluated
        # inside the fused kernel, hidden layers as fp32 MFMA -- its bound is the fp32 matrix / vector rate, not HBM
        try:
            hidden, ksteps = [64, 64], 200
            pe = vs.VecSimEnv(args.env, n, device=local_rank, **ENV_KW[args.env])
            pe.set_params(np.tile(vs.nominal_params(args.env), (n, 1)))
            net = vs.FNN(d["O"], d["A"], hidden, torch.tanh)
            pe.set_policy_fnn(net.param_values, hidden, "tanh", noise_std=np.full(d["A"], 0.1, dtype=np.float32))
            pe.set_auto_reset(True, seed=5)
            pe.reset(seed=6)
            pe.set_traj_capacity(ksteps)
            for _ in range(3):
                pe.step_policy(ksteps, record=True, noise_seed=3)
            pe.sync()
            pe.timer_start()
            for _ in range(10):
                pe.step_policy(ksteps, record=True, noise_seed=3)
            msp = pe.timer_stop() / 10
            dims = [d["O"]] + hidden + [d["A"]]
            flop = 2 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))  # per env step: the network's FMAs only
            tf = flop * n * ksteps / (msp * 1e-3) / 1e12
            roof["policy_fnn"] = {"kernel": "k_rollout_fnn (256-env workgroups, v_mfma_f32_32x32x2_f32)", "net": "6-64-64-1 tanh + exploration noise",
                                  "bound": "mfma", "dtype": "f32", "us_per_env_step_of_the_batch": msp * 1e3 / ksteps,
                                  "env_steps_per_s": n * ksteps / (msp * 1e-3), "flop_per_env_step": flop, "achieved": tf,
                                  "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS}
            pe.close()
        except Exception as exc:  # the headline must not depend on this leg
            roof["policy_fnn"] = {"error": repr(exc)}
    if args.mode == "fused" and args.env == "qq-su" and n == 65536:
        roof["configs"] = baseline_config_legs(local_rank, chunk)
        cn = counters_of(load_counters(), "headline", ms)
        if cn:
            roof["counters"] = cn
            if roof["traffic"] is None and "traffic" in cn:
                roof["traffic"], roof["traffic_source"] = cn["traffic"], cn["source"]
    roof["note"] = ("at 65 536 envs there is one wave of envs per SIMD: the fused kernel runs three cooperating waves per 64 envs "
                    "(k_rollout_ws: physics | reward + records | action generator + first record plane); its record stream is a "
                    "pure write stream -- `write_kernel_GBs` is what an in-repo pure-write probe reaches on this GPU, a reference "
                    "point beside the 8 TB/s of `peak`, not a bound (the kernel passed it once its vector-instruction count came "
                    "down: DESIGN.md section 7.0); `large_n` is the read + write HBM-bound point of the path") if args.mode == "fused" else \
                   ("one launch per env step; HBM-bound from ~1 M envs, launch-latency-bound at 65 536")
    return roof


if __name__ == "__main__":
    main()
