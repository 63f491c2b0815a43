#!/usr/bin/env python3
"""Front-end throughput on the GPU: int8 IQ at 69.984 MS/s x 85 ms (the reference's example rate and default --time 80 + 5 ms,
acquire-gps-l1.py:50-52,67,80) -> mix -> filtfilt(161 taps) -> resample, device-resident input."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, signals

fs, coff, ms_pad = 69984000.0, -9334875.0, 85
n = int(fs * 0.001 * ms_pad)
rng = np.random.default_rng(1)
iq = torch.from_numpy(rng.integers(-90, 90, size=2 * n, dtype=np.int8)).cuda()
eng = acquire.Engine(0)
eng.use_torch_stream()
for name in ("gps-l1", "gps-l5i", "galileo-e1b"):
    sig = signals.get(name)
    for _ in range(3):
        out = eng.frontend_dev(sig, iq, fs, coff, ms_pad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = eng.frontend_dev(sig, iq, fs, coff, ms_pad)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    L = n + 2 * 483
    alg = n * (2 + 8) + (n * 8 + L * 8) + (L * 8 + n * 8) + out.numel() * (16 + 8)       # mix, FIR fwd, FIR bwd, resample
    print(json.dumps({"signal": name, "in_samples": n, "out_samples": out.numel(), "ms": dt * 1e3, "Msamples_in_per_s": n / dt / 1e6,
                      "alg_GBps": alg / dt / 1e9, "frac_8TBps": alg / dt / 8e12, "x_realtime": (ms_pad * 1e-3) / dt}))
eng.close()
