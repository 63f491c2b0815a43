#!/usr/bin/env python3
"""Where the time of one host-buffer gacq_search call goes (GPS L1, 32 PRNs x 40 bins, one epoch): host wall time per call
vs the GPU-side span of the same call (HIP events around H2D + kernels on the engine's stream)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, signals, synth


def main():
    sig = signals.get("gps-l1")
    items = list(range(1, 33))
    ds = [-5000.0, 5000.0, 250.0]
    dop = acquire.doppler_grid(ds)
    x = synth.make_iq(sig, 1, 5, synth.default_sats(items), nsamp=4096)
    eng = acquire.Engine(0)
    for kv in sys.argv[1:]:                       # tuning switches: name=value (gacq_set_option)
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    print("options:", sys.argv[1:])
    for _ in range(20):
        eng.search_all(sig, x, items, ds, 1)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        eng.search_all(sig, x, items, ds, 1)
        ts.append(time.perf_counter() - t0)
    print("host call gacq_search: median %.1f us, p10 %.1f us" % (np.median(ts) * 1e6, np.percentile(ts, 10) * 1e6))
    # device-resident single epoch on torch's stream with events: kernels only
    eng.use_torch_stream()
    xd = torch.from_numpy(x[None]).cuda()
    out = torch.empty((1, 32, 2), dtype=torch.float64, device="cuda")
    for _ in range(20):
        eng.search_batch_dev(sig, xd, items, dop, 1, out=out)
    torch.cuda.synchronize()
    spans = []
    for _ in range(100):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.search_batch_dev(sig, xd, items, dop, 1, out=out)
        b.record()
        torch.cuda.synchronize()
        spans.append(a.elapsed_time(b) * 1e3)
    print("GPU span of the three kernels (events, idle stream): median %.1f us" % np.median(spans))
    eng.set_profiling(True)
    eng.reset_stage_times()
    for _ in range(50):
        eng.search_batch_dev(sig, xd, items, dop, 1, out=out)
    torch.cuda.synchronize()
    print("per-kernel (events):", {k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.stage_times().items() if v[1]}, "us")
    eng.set_profiling(False)
    # H2D of 32 KB from pinned memory alone
    pin = torch.from_numpy(x[None].copy()).pin_memory()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        xd.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("32 KB pinned H2D + sync round trip: median %.1f us" % (np.median(ts) * 1e6))
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("empty synchronize: median %.1f us" % (np.median(ts) * 1e6))
    # batched host entry point: per-epoch cost
    xs = np.stack([x] * 4096)
    for E in (1, 8, 64, 512, 2048, 4096):
        eng2 = eng
        eng2.set_stream(None)
        for _ in range(3):
            eng2.search_batch_host(sig, xs[:E], items, dop, 1, raw=True)
        t0 = time.perf_counter()
        n = max(3, 256 // E)
        for _ in range(n):
            eng2.search_batch_host(sig, xs[:E], items, dop, 1, raw=True)
        dt = (time.perf_counter() - t0) / n
        print("gacq_search_batch E=%3d: %.1f us per call, %.2f us per epoch, %.3e cells/s (PCIe-inclusive, results on the host)" % (E, dt * 1e6, dt * 1e6 / E, E * 32 * 40 * 4096 / dt))
    eng.close()


if __name__ == "__main__":
    main()
