#!/usr/bin/env python3
"""Leak / lifetime soak on the GPU box: contexts, signals, staging rings and device groups created and destroyed in a loop, searches
of every engine in between; free device memory (hipMemGetInfo through torch) must come back to where it started and the
host RSS (pinned staging, plans) must stay flat.
usage: tools/soak_leak_check.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import psutil
import torch

from gnss_dsp_tools_amd import acquire, signals, synth


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def one_round(i):
    for name, items, ds, ms in (("gps-l1", [3, 11, 19], [-1000.0, 1000.0, 250.0], 2), ("beidou-b1i", [6, 33], [0.0, 1000.0, 250.0], 2),
                                ("gps-l5i", [7, 8], [1000.0, 2000.0, 250.0], 1), ("galileo-e1b", [5], [1000.0, 1500.0, 125.0], 8),
                                ("glonass-l1", [-2, 5], [0.0, 750.0, 250.0], 1)):
        sig = signals.get(name)
        B = sig.blocks(ms)
        xs = synth.make_epochs(sig, B, 100 + i, synth.default_sats(items), 3, nsamp=sig.samples_needed(B))
        eng = acquire.Engine(0, workspace_bytes=(1 << 20) if i % 3 == 0 else None)
        try:
            if i % 2 == 0:
                # tie-safe re-evaluation buffers (lists, complex128 spectra, row scratch, per-block rows, counters) come and go with the
                # context too: eps = 2e-2 makes many pairs ambiguous, every fifth round re-evaluates every row
                eng.set_option("tie_eps_ppb", 1000000000 if i % 10 == 0 else 20000000)
                eng.set_option("tie_cap", 4096)
            eng.search_all(sig, xs[0], items, ds, ms)
            eng.search_batch_host(sig, xs, items, acquire.doppler_grid(ds), B)
            if i % 4 == 0:
                eng.set_engine(5)
                eng.search_all(sig, xs[0], items[:1], ds, ms)
        finally:
            eng.close()
    grp = acquire.DeviceGroup([0, 0])
    try:
        sig = signals.get("gps-l1")
        xs = synth.make_epochs(sig, 1, 7 + i, [(3, 0.4, 537.0, 1201)], 4, nsamp=4096)
        grp.search_batch_host(sig, xs, [3, 4, 11], acquire.doppler_grid([-2000.0, 2000.0, 250.0]), 1)
    finally:
        grp.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    one_round(0)                                  # first use: rocFFT / HIP runtime pools settle
    one_round(1)
    base = free_mb()
    rss0 = psutil.Process().memory_info().rss / 2**20
    low = base
    for i in range(2, n + 2):
        one_round(i)
        f = free_mb()
        low = min(low, f)
        if (i - 1) % 10 == 0:
            print("after %3d rounds: free %.0f MiB (start %.0f), host RSS %.0f MiB (start %.0f)" % (i - 1, f, base, psutil.Process().memory_info().rss / 2**20, rss0),
                  flush=True)
    end = free_mb()
    print("free device memory: start %.0f MiB, end %.0f MiB, lowest %.0f MiB" % (base, end, low))
    rss1 = psutil.Process().memory_info().rss / 2**20
    print("host RSS: start %.0f MiB, end %.0f MiB" % (rss0, rss1))
    if base - end > 64 or rss1 - rss0 > 256:
        print("LEAK: device %.0f MiB / host %.0f MiB did not come back" % (base - end, rss1 - rss0))
        sys.exit(1)


if __name__ == "__main__":
    main()
