"""Rebuild profiles/traffic.json from a round's profile directories.

usage: python profiles/update_traffic.py profiles/r03

For every workload directory that holds bench_profiled.json + pmc_summary.json (written by profile_workload.sh) the
record gives bench.py what it cannot measure inside its own process: HBM bytes per iteration of the timed launch
(FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes) and what limits the kernel.
useful_lane_frac = wave-instructions a ray's own work would need at 64 of 64 lanes / vector instructions executed per
ray, with the per-step instruction counts of the kernel's listing: 55 per descend step, 65 per triangle test, 330 per
shading step (the oracle's per-ray step counts are in the bench line of the same directory)."""
import json, os, sys

root = sys.argv[1]
out = {}
for wl in sorted(os.listdir(root)):
    d = os.path.join(root, wl)
    if not (os.path.exists(os.path.join(d, "pmc_summary.json")) and os.path.exists(os.path.join(d, "bench_profiled.json"))):
        continue
    p = json.load(open(os.path.join(d, "pmc_summary.json")))
    b = json.load(open(os.path.join(d, "bench_profiled.json")))
    r = b["roofline"]
    iters, rays = r["iterations_per_launch"], r["rays_per_launch"]
    valu = p["SQ_INSTS_VALU"] / rays
    pr = r["per_ray"]
    own = (55 * pr["nodes_visited"] + 65 * pr["tri_tests"] + 330) / 64
    wait = p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"]
    memory_bound = wait > 0.65
    out[f"{wl}:spath_run"] = {
        "hbm_bytes_per_iteration": (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024 / iters,
        "limiter": "scattered line fetches (L2 misses per ray against the GPU's gather ceiling)" if memory_bound
                   else "vector issue under divergence",
        "valu_insts_per_ray": round(valu, 1),
        "useful_lane_frac": round(own / valu, 3),
        "exec_lane_frac": round(p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"]), 3),
        "waves_waiting_frac": round(wait, 3),
        "l2_hit_rate": round(p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"]), 3),
        "l2_read_requests_to_memory_per_ray": round(p["TCC_EA0_RDREQ_sum"] / rays, 2),
        "waves_per_cu": round(p["MeanOccupancyPerCU"], 1),
        "profile": f"{d}/pmc_summary.json",
        "source": f"{d}/pmc_summary.json: (FETCH_SIZE + WRITE_SIZE) KiB x 1024 of the timed {iters:g}-iteration launch / {iters:g}; "
                  "FETCH_SIZE unscaled: every TCC->EA read request of these gathers is 64 bytes (FETCH_SIZE = 64 x "
                  "TCC_EA0_RDREQ here), calibrated on random 16-B and 48-B per-lane gathers in "
                  "profiles/r01_final/fetch_size_calibration.txt",
    }
json.dump(out, open(os.path.join(os.path.dirname(root.rstrip('/')), "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
