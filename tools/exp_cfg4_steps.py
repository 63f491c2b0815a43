#!/usr/bin/env python3
"""Per-step wall times of the config-4 step (two signals on one engine), 160 steps with a device synchronise after each: looks for one-off
stalls (allocations, plan builds, collector pauses) that a short timed region would catch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from gnss_dsp_tools_amd import acquire, sharded

dev = torch.device("cuda", 0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
jobs = bench.build_jobs(bench.CONFIGS[cfg], bench.CONFIGS[cfg]["epochs"], dev)
eng = acquire.Engine(0)
eng.use_torch_stream(dev)
sh = sharded.ShardedSearch(engine=eng)
ts = []
if len(sys.argv) > 2:
    import gc
    gc.collect()
    gc.disable()
for i in range(160):
    t0 = time.perf_counter()
    sh.search_jobs_async(jobs).wait()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("median %.3f ms; steps above 1.5 x median:" % np.median(ts), [(int(i), round(float(t), 2)) for i, t in enumerate(ts) if t > 1.5 * np.median(ts)])
eng.close()
