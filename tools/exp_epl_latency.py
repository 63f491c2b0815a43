#!/usr/bin/env python3
"""Where a resident-block early/prompt/late call (36 correlators, 4096 samples) spends its time: the Python wrapper, the bare C call
(gacq_correlate_batch_dev with pre-built arrays), and the same through the host-buffer entry point."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import _native as nat
from gnss_dsp_tools_amd import acquire, tracking

eng = acquire.Engine(0)
rng = np.random.default_rng(5)
xb = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
xd = torch.from_numpy(xb).cuda()
prns = np.arange(1, 13)
code_p = rng.uniform(0, 1023, 12)
cf = 1.023e6 / 4096000.0 * (1 + rng.uniform(-2e-6, 2e-6, 12))


def med(fn, n=300):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e6


print("python early_prompt_late, resident block: %.1f us" % med(lambda: tracking.early_prompt_late("gps.ca", xd, prns, code_p, cf, 0.05, engine=eng)))
print("python early_prompt_late, host block:     %.1f us" % med(lambda: tracking.early_prompt_late("gps.ca", xb, prns, code_p, cf, 0.05, engine=eng)))
plan = tracking.EplPlan("gps.ca", prns, 0.05, engine=eng)
print("python EplPlan call, resident block:      %.1f us" % med(lambda: plan(xd, code_p, cf)))
print("python EplPlan call, host block:          %.1f us" % med(lambda: plan(xb, code_p, cf)))
K = 36
p = np.ascontiguousarray(np.repeat(prns, 3), dtype=np.int32)
c = np.zeros(K)
f = (code_p[:, None] + np.array([-0.05, 0.0, 0.05])[None, :]).ravel().copy()
r = np.repeat(cf, 3).copy()
out = np.empty(K, dtype=np.complex128)
args = (eng._ctx, ctypes.c_void_p(xd.data_ptr()), 4096, b"gps.ca", 0, p.ctypes.data_as(nat.c_int_p), c.ctypes.data_as(nat.c_double_p),
        f.ctypes.data_as(nat.c_double_p), r.ctypes.data_as(nat.c_double_p), K, out.ctypes.data_as(nat.c_double_p))
print("bare C call gacq_correlate_batch_dev:     %.1f us" % med(lambda: nat.lib.gacq_correlate_batch_dev(*args)))
for opt in ("watch_results", "bar_upload"):
    eng.set_option(opt, 0)
    print("  with %s = 0:                 %.1f us" % (opt, med(lambda: nat.lib.gacq_correlate_batch_dev(*args))))
    eng.set_option(opt, 1)
eng.close()
