"""Which of config 5's searches gains from a second step in flight: every signal alone and all pairs, one context / stream against two
alternating ones (the bench's --lanes), same process, alternating measurements.  usage: python tools/exp_lanes_per_signal.py"""
import itertools
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
jobs = bench.build_jobs(bench.CONFIGS[5], 1, dev)


def lane():
    st = torch.cuda.Stream(dev)
    e = acquire.Engine(0)
    with torch.cuda.stream(st):
        return (lambda st=st: torch.cuda.stream(st)), sharded.ShardedSearch(engine=e)


L = [lane(), lane()]


def timed(lanes, sel, k):
    run = bench.make_run_steps(lanes, sel)
    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


subsets = [[i] for i in range(4)] + [list(c) for c in itertools.combinations(range(4), 2)] + [[0, 1, 2, 3]]
for sub in subsets:
    sel = [jobs[i] for i in sub]
    timed(L[:1], sel, 10)
    a, b = [], []
    for rep in range(3):
        a.append(timed(L[:1], sel, 40))
        b.append(timed(L, sel, 40))
    a, b = float(np.median(a)), float(np.median(b))
    print(json.dumps({"signals": [jobs[i]["label"] for i in sub], "one_lane_ms": round(a, 3), "two_lanes_ms": round(b, 3), "saved_ms": round(a - b, 3), "speedup": round(a / b, 3)}))
