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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng.search_batch_dev(job["sig"], job["x"], job["items"], job["dop"], job["B"])
torch.cuda.synchronize()
eng.close()
