#!/usr/bin/env python3
"""One config-5 signal (default glonass-l1) searched 30 times with tie-safe locations on: for rocprofv3 --kernel-trace --stats runs that
show what the re-evaluation kernels cost next to the search kernels.  usage: tools/exp_tie_one.py [label] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from gnss_dsp_tools_amd import acquire

label = sys.argv[1] if len(sys.argv) > 1 else "glonass-l1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
job = [j for j in bench.build_jobs(bench.CONFIGS[5], 1, dev) if j["label"] == label][0]
eng = acquire.Engine(0)
eng.use_torch_stream(dev)
for _ in range(3):
    eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
torch.cuda.synchronize()
print(label, "ms per search", (time.perf_counter() - t0) / reps * 1e3, eng.tie_stats())
eng.close()
