"""Config 5 (GPS L1 + E1B + B1I + GLONASS, one epoch) with work kept side by side on hardware queues of their own, against one
context / one stream: (a) consecutive steps alternating between contexts (bench.py --lanes), (b) the four searches of ONE step queued
on several contexts (ShardedSearch.enable_job_lanes), (c) both.  Alternating with the serial baseline in the same process; records are
compared with the serial ones.  usage: python tools/exp_cfg5_overlap.py"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import gnss_dsp_tools_amd as g
from gnss_dsp_tools_amd import acquire, sharded

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
jobs = bench.build_jobs(bench.CONFIGS[5], 1, dev)
cells = sum(j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
print(json.dumps({"jobs": [j["label"] for j in jobs], "cost_order": sorted(range(len(jobs)), key=lambda i: -sharded.ShardedSearch._job_cost(jobs[i]))}))
keep = []


def lane(job_lanes=1, order="cost", plain=False, cus=None):
    own = None if plain else acquire.MaskedStream(0)
    st = torch.cuda.Stream(dev) if plain else own.torch_stream
    keep.append(own)
    e = acquire.Engine(0)
    with torch.cuda.stream(st):
        sh = sharded.ShardedSearch(engine=e)
        if job_lanes > 1:
            sh.enable_job_lanes(job_lanes, order=order, cus=cus)
    return (lambda st=st: torch.cuda.stream(st)), sh


def timed(lanes, k):
    run = bench.make_run_steps(lanes, jobs)
    run(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, out


base = [lane()]
ref = [t.cpu().numpy().copy() for t in timed(base, 2)[1]]
timed(base, 30)
# jobs: 0 GPS L1, 1 E1B, 2 B1I, 3 GLONASS
variants = [("steps", 2, 1, "cost"), ("steps", 3, 1, "cost"), ("steps_plain_streams", 2, 1, "cost"), ("jobs", 1, 2, "cost"), ("jobs", 1, 3, "cost"), ("jobs", 1, 4, "cost"),
            ("jobs", 1, 2, [2, 3, 1, 0]), ("jobs", 1, 2, [1, 2, 3, 0]), ("jobs", 1, 2, [2, 1, 0, 3]), ("jobs", 1, 3, [2, 1, 3, 0]), ("both", 2, 2, "cost"),
            # E1B (and GPS L1) on a lane limited to 48 / 96 CUs, B1I and GLONASS on the whole device
            ("jobs_second_lane_on_48_cus", 1, 2, [2, 1, 3, 0]), ("jobs_second_lane_on_96_cus", 1, 2, [2, 1, 3, 0])]
for kind, nsteps, njobs, order in variants:
    cus = int(kind.split("_")[-2]) if kind.endswith("_cus") else None
    lanes = [lane(njobs, order, plain=kind.endswith("plain_streams"), cus=cus) for _ in range(nsteps)]
    timed(lanes, 4)
    res = []
    for rep in range(3):
        a, _ = timed(base, 30)
        b, out = timed(lanes, 30)
        res.append((a, b))
    same = all(np.array_equal(r, t.cpu().numpy()) for r, t in zip(ref, out))
    a = float(np.median([r[0] for r in res]))
    b = float(np.median([r[1] for r in res]))
    print(json.dumps({"kind": kind, "steps_in_flight": nsteps, "job_lanes": njobs, "order": order, "serial_ms": round(a, 3), "ms": round(b, 3), "speedup": round(a / b, 3),
                      "cells_per_s": float("%.4g" % (cells / (b * 1e-3))), "records_identical_to_serial": bool(same)}))
    torch.cuda.synchronize()
    for _, sh in lanes:
        sh.close_job_lanes()
        sh.engine.close()
