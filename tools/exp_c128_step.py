#!/usr/bin/env python3
"""A few 1024-epoch config-2 steps on engine 5 (fused4k_c128_kernel) for counter / trace runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from gnss_dsp_tools_amd import acquire

dev = torch.device("cuda", 0)
job = bench.build_jobs(bench.CONFIGS[2], 1024, dev)[0]
eng = acquire.Engine(0, engine=5)
eng.use_torch_stream(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
b.record()
torch.cuda.synchronize()
print("engine 5, 1024 epochs of config 2: %.3f ms per step over %d steps" % (a.elapsed_time(b) / n, n))
eng.close()
