"""HBM traffic per launch from the rocprofv3 PMC passes of scratch/profile_all.sh -> profiles/rNN_traffic.json
(FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled: the gfx950 correction of MI355X_MICROARCH.md's HBM section)."""
import csv
import glob
import json
import sys

prof, out = sys.argv[1], sys.argv[2]


def avg(dirname, counter, kernel):
    import os
    vals = []
    kernel = {"k_rollout_ws64": "k_rollout_ws<", "k_rollout_ws64g": "k_rollout_ws<", "k_rollout_ws256g": "k_rollout_ws<"}.get(kernel, kernel)  # (the shape is a template argument of k_rollout_ws)
    files = sorted(glob.glob(f"{prof}/{dirname}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # the newest pass only (a merged output directory may hold an older run's files too)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def block(fetch_dir, write_dir, kernel, alg_bytes, cmd):
    f, nf = avg(fetch_dir, "FETCH_SIZE", kernel)
    w, nw = avg(write_dir, "WRITE_SIZE", kernel)
    fb, wb = f * 1024 * 2, w * 1024
    return {"kernel": kernel, "launches_sampled": [nf, nw], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
            "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
            "algorithmic_bytes_per_launch": alg_bytes, "ratio": (fb + wb) / alg_bytes,
            "note": "FETCH_SIZE doubled (gfx950 reports half of a coalesced stream, MI355X_MICROARCH.md HBM section); "
                    f"separate --pmc passes of `{cmd}`; averages over the sampled launches"}


def last_json(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])


bd = last_json(f"{prof}/bench_trace.json")
import ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _lib as L
res = {"lib_version": int(L.load().vs_version()),
       "fused_default": block("fetch", "write", bd["roofline"]["kernel"], bd["roofline"]["alg_bytes_per_launch"],
                              "bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 50")}
bs = last_json(f"{prof}/bench_step16m.json")
res["fused_default"]["chunk"] = bd["config"]["chunk"]  # env steps per launch of the profiled command
res["step_16m"] = block("fetch_step16m", "write_step16m", "k_step", bs["roofline"]["alg_bytes_per_launch"],
                        "bench.py --no-cpu-baseline --no-extras --mode step --envs 16777216 --steps 30 --warmup 5")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["ratio"], 3) for k, v in res.items() if isinstance(v, dict)}))
