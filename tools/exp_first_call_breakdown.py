#!/usr/bin/env python3
"""Where the first search of a fresh process goes (what every command-line invocation pays): library load, context, signal build (chips,
replicas, code spectra), first search (workspaces, rocFFT plans of the tie-safe complex128 spectra, ...), second search.
usage (GPU box): tools/exp_first_call_breakdown.py <signal> [blocks]"""
import os
import sys
import time

t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

t_torch = time.perf_counter()
from gnss_dsp_tools_amd import acquire, signals, synth, codes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "beidou-b2ad"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
sig = signals.get(name)
items = codes.prns(sig.code) if not sig.item_sep == ":" else list(range(-7, 8))
dop = acquire.doppler_grid([-7000.0, 7000.0, 200.0])
need = (B + (1 if sig.pad else 0)) * sig.n
xs = synth.make_epochs(sig, 1, 5, [], 1, nsamp=need)
torch.cuda.init()
xd = torch.from_numpy(xs).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
eng = acquire.Engine(0)
eng.use_torch_stream()
t1 = time.perf_counter()
eng._plan(sig, items)
torch.cuda.synchronize()
t2 = time.perf_counter()
for tie in (1,):
    pk = eng.search_batch_dev(sig, xd, items, dop, B)
    torch.cuda.synchronize()
t3 = time.perf_counter()
pk = eng.search_batch_dev(sig, xd, items, dop, B)
torch.cuda.synchronize()
t4 = time.perf_counter()
print("%s B=%d: import torch %.2f s | context %.1f ms | signal build (%d items) %.1f ms | first search %.1f ms | second search %.1f ms"
      % (name, B, t_torch - t_start, 1e3 * (t1 - t0), len(items), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)))
