#!/usr/bin/env python3
"""Each signal of BASELINE config 5 searched alone, tie-safe locations on and then off on the same engine (third repetition timed, search +
synchronise): shows which signal of the synthetic workload has ambiguous pairs and what their complex128 re-evaluation costs."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, numpy as np
import bench
from gnss_dsp_tools_amd import acquire
dev = torch.device("cuda", 0)
jobs = bench.build_jobs(bench.CONFIGS[5], 1, dev)
for job in jobs:
    eng = acquire.Engine(0)
    eng.use_torch_stream(dev)
    for rep in range(3):
        t0 = time.perf_counter()
        out = eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(job["label"], "ms", round(dt * 1e3, 3), eng.tie_stats(), flush=True)
    eng.set_option("tie_safe", 0)
    for rep in range(3):
        t0 = time.perf_counter()
        out = eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(job["label"], "ms tie off", round(dt * 1e3, 3), flush=True)
    eng.close()
