#!/usr/bin/env python3
"""Run single legs of bench.py's default line without the rest: tools/bench_legs.py [reference_precision] [next_rows] [stream]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

env = {"world": 1, "rank": 0, "local_rank": 0, "use_dist": False, "dev": torch.device("cuda", 0)}
torch.cuda.set_device(0)
for leg in sys.argv[1:] or ["reference_precision", "next_rows", "stream"]:
    if leg == "reference_precision":
        print(json.dumps({leg: bench.reference_precision(None, env, 0.0)}))
    elif leg == "next_rows":
        print(json.dumps({leg: bench.next_rows(None, env)}))
    elif leg == "stream":
        from gnss_dsp_tools_amd import acquire
        e = acquire.Engine(0)
        print(json.dumps({leg: bench.stream_ceilings(e)}))
        e.close()
