#!/usr/bin/env python3
"""Wall time of the long-code C call alone (start phases prepared beforehand, samples resident on the device): GLONASS P and L2CL shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, longcode

eng = acquire.Engine(0)
rng = np.random.default_rng(5)
for name, code, K, blocks, n, fs in (("glonass p", "glonass.p", 1000, 5, 65536, 16384000.0), ("l2cl", "gps.l2cl", 75, 5, 81920, 4096000.0)):
    x = torch.view_as_complex(torch.randn((blocks * n, 2), device="cuda", dtype=torch.float32).contiguous())
    phase0 = np.ascontiguousarray(rng.uniform(0, 1e5, (K, blocks)))
    for _ in range(5):
        longcode._run(eng, x, fs, code, 1 if code != "glonass.p" else 0, 1234.0, phase0, blocks, n)
    t0 = time.perf_counter()
    N = 200
    for _ in range(N):
        q = longcode._run(eng, x, fs, code, 1 if code != "glonass.p" else 0, 1234.0, phase0, blocks, n)
    dt = (time.perf_counter() - t0) / N
    print("%-10s K=%d blocks=%d n=%d: %.1f us per call, sum(q)=%.6e" % (name, K, blocks, n, dt * 1e6, float(np.sum(q))))
eng.close()
