"""PMC counters per kernel of the default bench command -> profiles/rNN_counters.json (scratch/r3/profile_r3.sh).

HBM traffic: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (the gfx950 correction of MI355X_MICROARCH.md's HBM section).
SQ counters are summed over the chip; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (same guide)."""
import collections
import csv
import glob
import json
import os
import sys

prof, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# which kernel each leg of the line launches (substring of the demangled name, spaces removed)
KERNELS = {
    "headline": "k_rollout_ws<vs::QQT<0>,false,true,1,4,256,false,3,0>",
    "record2": "k_rollout_ws<vs::QQT<0>,false,true,2,4,256,false,3,0>",
    "config2": "k_rollout_ws<vs::QQT<0>,false,true,1,4,64,false,3,0>",
    "config3": "k_rollout_ws<vs::QcpT<0>,false,true,1,4,64,false,3,1>",
    "config4": "k_rollout_ws<vs::Qbb,false,true,1,4,64,false,3,0>",
    "config5": "k_rollout_mixed<true,1,false>",
    "large_n": "k_step<vs::QQT<0>,false,true,false,0,false,true>",
    "pack_traj": "k_pack_traj<vs::QQT<0>,2>",
}


def collect(dirname):
    """{leg: {counter: (mean value per dispatch, dispatches)}}"""
    res = {k: collections.defaultdict(list) for k in KERNELS}
    files = sorted(glob.glob(f"{prof}/{dirname}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace(" ", "")
            for leg, pat in KERNELS.items():
                if pat in name:
                    res[leg][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {leg: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for leg, d in res.items()}


def last_json(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])


bd = last_json(f"{prof}/bench_trace.json")
roof = bd["roofline"]
legs = {"headline": dict(env_steps=roof["env_steps_per_launch"], alg_bytes=roof["alg_bytes_per_launch"])}
if "record2" in roof and "alg_bytes_per_env_step" in roof["record2"]:
    legs["record2"] = dict(env_steps=roof["env_steps_per_launch"], alg_bytes=roof["record2"]["alg_bytes_per_env_step"] * roof["env_steps_per_launch"])
for k, v in roof.get("configs", {}).items():
    if "env_steps_per_launch" in v:
        legs[k] = dict(env_steps=v["env_steps_per_launch"], alg_bytes=v["alg_bytes_per_env_step"] * v["env_steps_per_launch"])
if "large_n" in roof and "envs" in roof["large_n"]:
    legs["large_n"] = dict(env_steps=roof["large_n"]["envs"], alg_bytes=roof["large_n"]["alg_bytes_per_env_step"] * roof["large_n"]["envs"])
if "pack_traj" in roof and "recorded_steps" in roof["pack_traj"]:
    legs["pack_traj"] = dict(env_steps=roof["pack_traj"]["recorded_steps"],
                             alg_bytes=roof["pack_traj"]["alg_bytes_per_recorded_step"] * roof["pack_traj"]["recorded_steps"])

fetch, write, sq1, sq2 = collect("fetch"), collect("write"), collect("sq1"), collect("sq2")
from simurlacra_amd import _lib as L

res = {"lib_version": int(L.load().vs_version()),
       "command": "rocprofv3 --pmc <counters> -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 (BENCH_PREROLL=50), one pass per "
                  "counter group: FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2; averages per dispatch of the leg's kernel",
       "units": "FETCH/WRITE_SIZE in KB (FETCH doubled in fetch_bytes_corrected); SQ *_CYCLES / ACTIVE_INST_* / WAIT_* in quad-cycles "
                "summed over all waves (or SIMDs for BUSY_CYCLES) of the chip"}
for leg, meta in legs.items():
    e = {"kernel": KERNELS[leg], "env_steps_per_launch": meta["env_steps"], "algorithmic_bytes_per_launch": meta["alg_bytes"]}
    f, w = fetch[leg].get("FETCH_SIZE"), write[leg].get("WRITE_SIZE")
    if f and w:
        fb, wb = f[0] * 1024 * 2, w[0] * 1024
        e.update(FETCH_SIZE_KB=f[0], WRITE_SIZE_KB=w[0], fetch_bytes_corrected=fb, write_bytes=wb, traffic_bytes_per_launch=fb + wb,
                 traffic_over_algorithmic=(fb + wb) / meta["alg_bytes"], launches_sampled=[f[1], w[1]])
    sq = {}
    for src in (sq1[leg], sq2[leg]):
        for c, (v, n) in src.items():
            sq[c] = v
    if sq:
        e["sq"] = sq
        waves, steps64 = sq.get("SQ_WAVES"), meta["env_steps"] / 64.0
        d = {}
        if "SQ_INSTS_VALU" in sq:
            d["valu_insts_per_64_env_steps"] = sq["SQ_INSTS_VALU"] / steps64
            d["salu_insts_per_64_env_steps"] = sq.get("SQ_INSTS_SALU", 0.0) / steps64
            d["lds_insts_per_64_env_steps"] = sq.get("SQ_INSTS_LDS", 0.0) / steps64
        if "SQ_ACTIVE_INST_VALU" in sq and "SQ_WAVE_CYCLES" in sq:
            # share of the waves' lifetime in which they had a VALU instruction executing / any instruction executing
            d["valu_active_share_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
            d["any_inst_active_share_of_wave_cycles"] = sq.get("SQ_ACTIVE_INST_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
            d["valu_quad_cycles_per_valu_inst"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
        if "SQ_WAIT_ANY" in sq and "SQ_WAVE_CYCLES" in sq:
            d["wait_any_share_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            d["wait_inst_any_share_of_wave_cycles"] = sq.get("SQ_WAIT_INST_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
        if waves:
            d["waves_per_launch"] = waves
        e["derived"] = d
    res[leg] = e
json.dump(res, open(out, "w"), indent=1)
for leg in legs:
    e = res[leg]
    print(leg, "traffic/alg", round(e.get("traffic_over_algorithmic", float("nan")), 3), {k: round(v, 3) for k, v in e.get("derived", {}).items()})
