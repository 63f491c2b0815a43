#!/usr/bin/env python3
"""Peak records of a few fixed searches as a SHA-256: the A/B check of a variant build --
`python tools/dump_peaks.py` and `python tools/variant.py <name> tools/dump_peaks.py` must print the same digest."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, signals, synth

eng = acquire.Engine(0)
eng.use_torch_stream()
h = hashlib.sha256()
for name, items, ds, ms, E in (("beidou-b1i", list(range(1, 64)), [-10000.0, 10000.0, 500.0], 10, 1), ("glonass-l1", list(range(-7, 8)), [-3000.0, 3000.0, 250.0], 4, 2),
                               ("beidou-b2i", [1, 5, 9], [-1000.0, 1000.0, 100.0], 2, 3), ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 5)):
    sig = signals.get(name)
    B = sig.blocks(ms)
    xs = synth.make_epochs(sig, B, 77, synth.default_sats(items), E, nsamp=sig.samples_needed(B))
    pk = eng.search_batch_dev(sig, torch.from_numpy(xs).cuda(), items, acquire.doppler_grid(ds), B)
    torch.cuda.synchronize()
    h.update(pk.cpu().numpy().tobytes())
print("peaks sha256", h.hexdigest())
eng.close()
