"""Config 5 as a two-stage pipeline: the memory-bound search(es) of step i+1 on one context / hardware queue, the arithmetic-bound ones of
step i on another, both free-running (no join per step) -- against whole steps alternating between two contexts (bench.py --lanes 2) and
against one context.  usage: python tools/exp_cfg5_split_lanes.py"""
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
jobs = bench.build_jobs(bench.CONFIGS[5], 1, dev)          # 0 GPS L1, 1 E1B, 2 B1I, 3 GLONASS
cells = sum(j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
keep = []


def lane():
    own = acquire.MaskedStream(0)
    keep.append(own)
    e = acquire.Engine(0)
    with torch.cuda.stream(own.torch_stream):
        return own.torch_stream, sharded.ShardedSearch(engine=e)


L = [lane() for _ in range(3)]


def run_split(split, k):
    """split: list of job-index lists, one per lane; every lane runs its subset k times, free-running"""
    for _ in range(k):
        for (st, sh), sub in zip(L, split):
            with torch.cuda.stream(st):
                sh.search_jobs_async([jobs[i] for i in sub]).wait()


def run_steps(nl, k):
    for i in range(k):
        st, sh = L[i % nl]
        with torch.cuda.stream(st):
            sh.search_jobs_async(jobs).wait()


def timed(fn, k=40):
    fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


variants = [("one context", lambda k: run_steps(1, k)), ("whole steps on 2 contexts", lambda k: run_steps(2, k)), ("whole steps on 3 contexts", lambda k: run_steps(3, k)),
            ("E1B | B1I GLO L1", lambda k: run_split([[1], [2, 3, 0]], k)), ("E1B L1 | B1I GLO", lambda k: run_split([[1, 0], [2, 3]], k)),
            ("E1B GLO | B1I L1", lambda k: run_split([[1, 3], [2, 0]], k)), ("E1B | B1I L1 | GLO", lambda k: run_split([[1], [2, 0], [3]], k)),
            ("E1B | B1I | GLO L1", lambda k: run_split([[1], [2], [3, 0]], k))]
for rep in range(2):
    for name, fn in variants:
        ms = float(np.median([timed(fn) for _ in range(3)]))
        print(json.dumps({"variant": name, "ms_per_step": round(ms, 3), "cells_per_s": float("%.4g" % (cells / (ms * 1e-3)))}))
