"""Upper bound on what the code-spectrum reads cost the headline kernel: the same 1024-epoch step with 32 DIFFERENT items (32 x 32 KB of code
spectra re-read from L2 by every workgroup: 44 GB per step through the L2 -> CU path) against 32 times the SAME item (one 32 KB row, hot in
every CU's vector cache).  Same arithmetic, same number of rows.  usage: python tools/exp_code_spectrum_traffic.py"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import gnss_dsp_tools_amd as g
from gnss_dsp_tools_amd import acquire

dev = torch.device("cuda", 0)
job = bench.build_jobs(bench.CONFIGS[2], 1024, dev)[0]
e = acquire.Engine(0)
e.use_torch_stream(dev)
dop = job["dopplers"]


def timed(items, k=40):
    for _ in range(8):
        e.search_batch_dev(job["name"], job["x"], items, dop, job["blocks"])
    torch.cuda.synchronize()
    s = bench.ClockSampler(0).start()
    t0 = time.perf_counter()
    for _ in range(k):
        e.search_batch_dev(job["name"], job["x"], items, dop, job["blocks"])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    c = s.stop()
    return ms, c


for rep in range(3):
    for label, items in (("32 different items", list(range(1, 33))), ("one item 32 times", [7] * 32), ("two items 16 times each", [7, 19] * 16), ("four items 8 times each", [3, 7, 19, 28] * 8)):
        ms, c = timed(items)
        print(json.dumps({"items": label, "ms_per_step": round(ms, 4), "sclk_mhz_mean": round(c["sclk_mhz_mean"]), "power_w_mean": round(c["power_w_mean"])}))
