"""Config 5's four searches in every queueing order on ONE context / stream (no overlap): the memory-bound E1B search pulls the shader
clock down (the power budget moves to HBM), and the arithmetic-bound search queued behind it pays for that until the clock is back.
ms per step for all 24 orders, alternating with the listing order, same process.  usage: python tools/exp_cfg5_order.py"""
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
short = {"gps-l1": "L1", "galileo-e1b": "E1B", "beidou-b1i": "B1I", "glonass-l1": "GLO"}
e = acquire.Engine(0)
e.use_torch_stream(dev)
sh = sharded.ShardedSearch(engine=e)


def timed(sel, k):
    for _ in range(4):
        sh.search_jobs_async(sel).wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        sh.search_jobs_async(sel).wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


timed(jobs, 40)
out = []
for perm in itertools.permutations(range(4)):
    sel = [jobs[i] for i in perm]
    a = [timed(jobs, 25) for _ in range(2)]
    b = [timed(sel, 25) for _ in range(2)]
    out.append((float(np.mean(b)), float(np.mean(a)), "-".join(short[jobs[i]["label"]] for i in perm)))
for b, a, name in sorted(out):
    print(json.dumps({"order": name, "ms": round(b, 3), "listing_order_ms": round(a, 3), "speedup": round(a / b, 3)}))
