"""Why config 5's step (7.3-7.9 ms) is longer than the sum of its four searches run alone (6.6 ms): shader clock and socket power
sampled (amdgpu sysfs, bench.ClockSampler) while each search loops alone for ~1.5 s, while the whole step loops on one context, and
while it loops with two steps in flight.  usage: python tools/exp_cfg5_clocks.py"""
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


def loop(lanes, sel, seconds=1.5):
    run = bench.make_run_steps(lanes, sel)
    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)
    torch.cuda.synchronize()
    k = max(10, int(seconds / ((time.perf_counter() - t0) / 10)))
    s = bench.ClockSampler(0).start()
    t0 = time.perf_counter()
    run(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    c = s.stop()
    return ms, c


rows = [([i], 1) for i in range(4)] + [([0, 1, 2, 3], 1), ([0, 1, 2, 3], 2), ([1, 2], 1), ([1, 2], 2), ([2, 3], 1)]
alone = 0.0
for sub, nl in rows * 2:
    ms, c = loop(L[:nl], [jobs[i] for i in sub])
    print(json.dumps({"signals": [jobs[i]["label"] for i in sub], "steps_in_flight": nl, "ms_per_step": round(ms, 3),
                      "sclk_mhz_mean": round(c["sclk_mhz_mean"]), "sclk_mhz_min": c["sclk_mhz_min"], "sclk_mhz_max": c["sclk_mhz_max"], "power_w_mean": round(c["power_w_mean"])}))
