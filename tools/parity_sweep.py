#!/usr/bin/env python3
"""Statistical parity sweep on the GPU box: many seeded epochs (mostly noise-only PRNs, i.e. near-tied peaks) through the HIP
path and through the fp64 oracle; reports peak-location mismatches and the worst relative metric error per signal.
This is evidence, not a test: results go to profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gnss_dsp_tools_amd import acquire, signals, synth
from oracle import acq_oracle

PLAN = [
    # signal, items, doppler_search, ms, epochs
    ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 1000),
    ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 500.0], 4, 100),
    ("beidou-b1i", [1, 6, 20, 33, 63], [-2000.0, 2000.0, 250.0], 3, 40),
    ("glonass-l1", [-7, -1, 0, 4], [-2000.0, 2000.0, 250.0], 2, 40),
    ("galileo-e1b", [3, 11, 24], [-1000.0, 1000.0, 125.0], 8, 20),
    ("gps-l5i", [1, 7, 30], [-1000.0, 1000.0, 200.0], 2, 20),
    ("galileo-e6b", [2, 9], [-1000.0, 1000.0, 200.0], 2, 20),
    ("gps-l1cd", [9], [1400.0, 1700.0, 50.0], 10, 3),
    ("gps-l2cm", [3, 17], [-400.0, 400.0, 100.0], 40, 3),
    ("xona-x5p", [0], [-1000.0, 1000.0, 200.0], 3, 10),
]


PLAN_16K = [   # round 3: the rebuilt N = 16384 engine on its own (usage: tools/parity_sweep.py 16k)
    ("beidou-b1i", list(range(1, 64)), [-2000.0, 2000.0, 250.0], 3, 30),
    ("beidou-b2i", list(range(1, 11)), [-1000.0, 1000.0, 125.0], 2, 30),
    ("glonass-l1", list(range(-7, 8)), [-2000.0, 2000.0, 250.0], 2, 40),
    ("glonass-l2", list(range(-7, 8)), [-1500.0, 1500.0, 250.0], 1, 20),
]


def main():
    eng = acquire.Engine(0)
    for name, items, ds, ms, epochs in (PLAN_16K if sys.argv[1:2] == ["16k"] else PLAN):
        sig = signals.get(name)
        B = sig.blocks(ms)
        rows = mism = 0
        worst = 0.0
        t0 = time.time()
        # epochs go through the host-batch entry point in chunks of 32: for GPS L1 that is 32 x 40 = 1280 (epoch, Doppler) units,
        # i.e. the fused forward + correlate kernel of the bench path; the other signals run their auto engines batched
        for e0 in range(0, epochs, 32):
            xs, ne = [], min(32, epochs - e0)
            for e in range(e0, e0 + ne):
                sats = [(items[e % len(items)], 0.25, 1537.0 if ds[0] <= 1537.0 < ds[1] else 0.5 * (ds[0] + ds[1]) + 13.0, 1201 + 17 * e)] if e % 3 == 0 else []
                xs.append(synth.make_iq(sig, B, 770000 + 31 * e, sats, nsamp=sig.samples_needed(B)))
            got_all = eng.search_batch_host(sig, np.stack(xs), items, acquire.doppler_grid(ds), B)
            for x, got in zip(xs, got_all):
                xw = x.astype(np.complex128)
                for it, g in zip(items, got):
                    w = acq_oracle.search_script(name, xw, it, ds, ms)
                    rows += 1
                    if float(g[1]) != float(w[1]) or float(g[2]) != float(w[2]):
                        mism += 1
                    else:
                        worst = max(worst, abs(float(g[0]) - float(w[0])) / abs(float(w[0])))
        print(json.dumps({"signal": name, "N": sig.nfft, "blocks": B, "doppler_bins": len(np.arange(*ds)), "searches": rows,
                          "location_mismatches": mism, "worst_rel_metric_err": worst, "seconds": round(time.time() - t0, 1)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
