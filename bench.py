#!/usr/bin/env python3
"""
bench.py -- env-steps/sec of the MI355X-native stepper on BASELINE.json's metric:
"env-steps/sec whole node, 65 536 QQubeSwingUpSim envs, random policy".

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over the batch = one launch.  In the default fused mode a launch advances every one of
the 65 536 environments of a rank by `--chunk` (100) env steps -- SimPyEnv.step: reward -> clip -> dead zone -> integrate
-> done -> observe, finished lanes auto-reset in the same kernel, actions from the on-device uniform random policy
(DummyPolicy), obs/act/rew/done of every env step recorded; in `--mode step` a launch is one env step.  `value` counts ENV
steps: envs x env-steps-per-launch x K / time ("env_steps_per_step" in the JSON).  State and constants are resident in HBM
when the timed region starts.  Each rank owns 65 536 envs on its own GPU (weak scaling, the batch
shards embarrassingly); the only collective is an RCCL all-gather of completed-episode return statistics at the end.

Modes (--mode):
  fused  (default) vs_step_random: `--chunk` env steps per launch, state in registers, obs/act/rew/done of EVERY step
                   streamed to the trajectory buffers (record=1) -- what a rollout sampler needs.
  step             one vs_step launch per env step, actions drawn by torch.rand on the GPU each step (policy in the loop).

The JSON line also carries the roofline of the dominant kernel (HIP-event timed on the kernel's stream) and the CPU
baseline (the oracle's NumPy port timed on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ALGORITHMIC bytes per env-step, fp32 SoA:
#   single-step kernel: 4*[(S+A+P+H+1) read + (S+H+O+1+1) written] + 1 done byte
#   fused rollout kernel with record: per step only the records leave the chip: 4*(O + A + 1) + 1; state/constants
#   are read and written once per launch (amortised over `chunk` steps and added below)
DIMS = {"omo": dict(S=2, A=1, O=2, P=3, H=0), "bob": dict(S=4, A=1, O=4, P=8, H=0), "qq-su": dict(S=4, A=1, O=6, P=11, H=0),
        "qcp-su": dict(S=4, A=1, O=5, P=17, H=1), "qbb": dict(S=8, A=2, O=8, P=20, H=2),
        "qq-st": dict(S=4, A=1, O=6, P=11, H=0), "qcp-st": dict(S=4, A=1, O=5, P=17, H=1),
        "pend": dict(S=2, A=1, O=3, P=5, H=0), "bob-d": dict(S=4, A=1, O=4, P=8, H=0)}
ENV_KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
          "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500),
          "qq-st": dict(dt=0.01, max_steps=500), "qcp-st": dict(dt=0.01, max_steps=300),
          "pend": dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2])), "bob-d": dict(dt=0.01, max_steps=500)}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def bytes_single_step(d):
    return 4 * ((d["S"] + d["A"] + d["P"] + d["H"] + 1) + (d["S"] + d["H"] + d["O"] + 1 + 1)) + 1


def bytes_fused_step(d, chunk, record):
    per_launch = 4 * ((d["S"] + d["P"] + d["H"] + 1) + (d["S"] + d["H"] + d["O"] + 1 + 1)) + 1  # state/consts in+out once
    per_step = (4 * (d["O"] + d["A"] + 1) + 1) if record else 0
    return per_step + per_launch / chunk


def _cpu_worker(args):
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
    # scalar mode: one env stepped in a Python loop (how the reference itself is driven, minus its deepcopy)
    k, t1 = 0, time.perf_counter()
    if seed == 0:
        p1, s1, h1, st1 = params[:1], state[:1].copy(), hidden[:1].copy(), steps[:1].copy()
        while time.perf_counter() - t1 < 2.0:
            o = ref.step(s1, h1, rng.uniform(alo[:1], ahi[:1]), p1, st1)
            s1, h1, st1 = o["state"], o["hidden"], o["curr_step"]
            if o["done"][0]:
                s1, st1 = fresh_states()[:1], st1 * 0
            k += 1
    return n_envs * n_steps / el, n_steps, el, k / max(time.perf_counter() - t1, 1e-9)


def cpu_baseline(env_name, n_envs, budget_s=10.0):
    """CPU baseline beside the GPU number: the oracle (the NumPy port of the reference algorithm) on the host cores of
    this box, one process per core, each vectorised over its share of the same 65 536-env workload.  Runs BEFORE the GPU
    is touched (worker processes are forked from a process that has not initialised HIP)."""
    import multiprocessing as mp

    cores = max(1, min(16, os.cpu_count() or 1))  # the GPU box gives one job 16 CPU cores
    share = n_envs // cores
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(env_name, share, budget_s, r) for r in range(cores)])
    total = float(sum(r[0] for r in res))
    return dict(value=total, unit="env-steps/s", cores=cores, kind="port",
                sample=f"oracle/cpu_ref.py (NumPy fp64) in {cores} processes x {share} envs, vectorised, {res[0][1]} batch steps "
                       f"per process in {res[0][2]:.1f} s incl. auto-reset; one process alone: {res[0][0]:.3g} env-steps/s; scalar N=1 "
                       f"loop: {res[0][3]:.0f} env-steps/s; reference Pyrado itself: 2.1-2.6e3 env-steps/s/core (BASELINE.md, "
                       f"measured in the build container)",
                single_process_value=float(res[0][0]), scalar_value=float(res[0][3]), host_cores=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed launches (default 200 fused / 2000 step mode)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed launches (default 20 fused / 200 step mode)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--env", default="qq-su", choices=sorted(DIMS))
    ap.add_argument("--mode", default="fused", choices=["fused", "step"])
    ap.add_argument("--chunk", type=int, default=100, help="env steps per launch in fused mode")
    ap.add_argument("--record", type=int, default=1)
    ap.add_argument("--per-env-params", type=int, default=1, help="1: per-env constants [K][N] (DR-capable), 0: broadcast")
    ap.add_argument("--live-dr", type=int, default=0, help="DomainRandWrapperLive on the device: redraw the first K parameters "
                    "of the family's default randomizer at every reset (BASELINE config 3: qcp-su with K = 7)")
    ap.add_argument("--graph", type=int, default=0, help="step mode: capture `chunk` policy+step iterations in one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    cpu_base = None
    if not args.no_cpu_baseline and world == 1:
        cpu_base = cpu_baseline(args.env, min(args.envs, 65536))  # before any HIP call (forked workers)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- a 1-GPU rehearsal of the multi-rank code path
    # (the real runs use one GPU per rank and RCCL)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)

    import simurlacra_amd as vs
    from simurlacra_amd import _lib as L

    kw = ENV_KW[args.env]
    d = DIMS[args.env]
    n = args.envs
    env = vs.VecSimEnv(args.env, n, device=local_rank, **kw)
    if args.per_env_params:
        env.set_params(np.tile(vs.nominal_params(args.env), (n, 1)))
    if args.live_dr:
        rz = vs.create_default_randomizer(vs.ENV_CLASSES[args.env](**kw))
        env.set_randomizer(rz.device_specs()[: args.live_dr])
    from simurlacra_amd.dist import gather_episode_stats, shard

    first, _ = shard(n * world, rank, world)
    env.set_index_offset(first)  # global env index: lane streams do not depend on the number of GPUs
    env.set_auto_reset(True, seed=args.seed * 1000 + 1)
    env.reset(seed=args.seed * 7919 + 2)
    if args.steps is None:
        args.steps = 200 if args.mode == "fused" else 2000
    if args.warmup is None:
        args.warmup = 20 if args.mode == "fused" else 200
    chunk = max(1, args.chunk)
    steps = args.steps  # launches
    per_step = chunk if (args.mode == "fused" or args.graph) else 1  # env steps per launch
    act_hi = {"omo": 30.0, "bob": 29.43, "qq-su": 4.5, "qcp-su": 6.0, "qbb": 3.0, "qq-st": 4.5, "qcp-st": 6.0, "pend": 3.5,
              "bob-d": 29.43}[args.env]

    graph = None

    def policy_and_step():
        act = (torch.rand(n, d["A"], device=f"cuda:{local_rank}") * 2 - 1) * act_hi  # DummyPolicy on the GPU
        env.step(act)

    def run(k_launches):
        if args.mode == "fused":
            for _ in range(k_launches):
                env.step_random(chunk, seed=args.seed + 3, record=bool(args.record))
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
    run(max(args.warmup, 1))
    env.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    env.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # gather completed-episode return statistics over RCCL (the only collective of this path): per-env accumulators
    # reduced on the device, three doubles per rank on the wire
    cnt_t, rs_t, ls_t = (env.tensor(w)[0, :n] for w in (L.VS_EPSTAT_COUNT, L.VS_EPSTAT_RETSUM, L.VS_EPSTAT_LENSUM))
    ep = gather_episode_stats(cnt_t, rs_t, ls_t)
    el_t = torch.tensor([el], device="cpu" if rehearsal else f"cuda:{local_rank}", dtype=torch.float64)
    if dist:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    el = float(el_t.item())
    errs = env.error_count()

    if rank == 0:
        total_env_steps = float(n) * steps * per_step * world
        value = total_env_steps / el
        # roofline of the dominant kernel: HIP events on the kernel's own stream
        if args.mode == "fused":
            ms = env.time_step_kernel(iters=20, k_steps=chunk, record=bool(args.record))
            b_per = bytes_fused_step(d, chunk, bool(args.record))
            units = n * chunk
            kname = env.rollout_variant()
        else:
            act = (torch.rand(n, d["A"], device=f"cuda:{local_rank}") * 2 - 1) * act_hi
            torch.cuda.synchronize()
            ms = env.time_step_kernel(iters=200, actions=act)
            b_per = bytes_single_step(d)
            units = n
            kname = "k_step"
        achieved = b_per * units / (ms * 1e-3) / 1e9
        # the practical HBM ceiling next to the nominal one -- an in-repo float4 copy kernel (read + write of
        # 1 GiB per pass, far beyond the caches), timed with HIP events
        copy_gbs, write_gbs = None, None
        try:
            import ctypes

            g = ctypes.c_float()
            if L.load().vs_membw_probe(local_rank, 1 << 30, 10, ctypes.byref(g)) == 0:
                copy_gbs = float(g.value)
            if L.load().vs_memwrite_probe(local_rank, 1 << 30, 10, ctypes.byref(g)) == 0:
                write_gbs = float(g.value)  # a pure write stream: what the record stores of the fused kernel compete with
        except Exception:
            pass
        # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes of this very
        # command; gfx950 correction applied) -- measured once per round and committed under profiles/
        traffic, traffic_src = None, None
        try:
            import glob

            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
            tab = json.load(open(tf))
            if (args.mode == "fused" and args.env == "qq-su" and n == 65536 and chunk == 100 and args.record == 1
                    and args.per_env_params == 1):
                traffic, traffic_src = tab["fused_default"]["traffic_bytes_per_launch"], os.path.basename(tf)
            elif args.mode == "step" and args.env == "qq-su" and n == 16777216:
                traffic, traffic_src = tab["step_16m"]["traffic_bytes_per_launch"], os.path.basename(tf)
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec whole node, 65 536 QQubeSwingUpSim envs, random policy",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": el / steps * 1e3, "env_steps_per_step": per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env} x {n} envs per GPU, dt {kw['dt']}, max_steps {kw['max_steps']}, uniform random "
                                   f"policy on device, auto-reset, mode={args.mode}"
                                   + (f", {chunk} steps/launch, record={args.record}" if args.mode == "fused" else "")
                                   + (f", hipGraph of {chunk} (policy, step) pairs" if graph is not None else "")
                                   + (", per-env constants" if args.per_env_params else ", broadcast constants")
                                   + (f", live domain randomisation of {args.live_dr} parameters at every reset" if args.live_dr else ""),
                       "envs_per_gpu": n, "env": args.env, "mode": args.mode, "chunk": chunk, "record": args.record,
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "copy_kernel_GBs": copy_gbs,
                         "frac_of_copy_kernel": (achieved / copy_gbs) if copy_gbs else None,
                         "write_kernel_GBs": write_gbs,
                         "frac_of_write_kernel": (achieved / write_gbs) if write_gbs else None, "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                         "traffic_source": traffic_src, "alg_bytes_per_launch": b_per * units, "kernel": kname,
                         "kernel_ms": ms, "alg_bytes_per_env_step": b_per, "env_steps_per_launch": units,
                         "note": ("at 65 536 envs there is one wave of envs per SIMD: the fused kernel is bound by VALU issue "
                                  "(~250 instructions per 64 envs and step; k_rollout_ws splits them over two waves per "
                                  "SIMD, DESIGN.md sections 4 and 7), not by HBM; with records on the write stream "
                                  "saturates at ~4 TB/s (1.2e11 env-steps/s) from 131 072 envs") if args.mode == "fused" else
                                 ("one launch per env step; HBM-bound from ~1 M envs (at the rate of the in-repo copy kernel "
                                  "at 16.7 M envs), launch-latency-bound at 65 536")},
            "episodes": {"completed": ep["episodes"], "mean_return": ep["mean_return"], "mean_length": ep["mean_length"]},
            "nan_flags": errs,
        }
        out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    env.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
