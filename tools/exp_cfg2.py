#!/usr/bin/env python3
"""Experiments on the headline shape (GPS L1, 32 PRNs x 40 bins, B=1): batch size x items-per-workgroup sweep and a
per-step timeline after an idle period.  usage: exp_cfg2.py [sweep] [timeline]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, signals, synth


def main():
    what = sys.argv[1:] or ["sweep", "timeline"]
    sig = signals.get("gps-l1")
    items = list(range(1, 33))
    dop = acquire.doppler_grid([-5000.0, 5000.0, 250.0])
    eng = acquire.Engine(0)
    eng.use_torch_stream()
    base = synth.make_epochs(sig, 1, 77, synth.default_sats(items), 8, nsamp=4096)
    if "sweep" in what:
        for E, fused, pch in [(E, fused, pch) for E in (8, 64, 256) for fused in (0, 1) for pch in (0, 8, 16, 32)]:
            xd = torch.from_numpy(np.concatenate([base] * max(1, E // 8))[:E]).cuda()
            if True:
                eng.set_option("lds_pch", pch)
                eng.set_option("fused_4k", 2 * fused)
                for _ in range(10):
                    eng.search_batch_dev(sig, xd, items, dop, 1)
                torch.cuda.synchronize()
                n = max(20, 4096 // E)
                t0 = time.perf_counter()
                for _ in range(n):
                    eng.search_batch_dev(sig, xd, items, dop, 1)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print("E=%4d fused=%d pch=%2d  %.4f ms/step  %.4f ms per 64 epochs  %.3e cells/s" % (E, fused, pch, dt * 1e3, dt * 1e3 * 64 / E, E * 32 * 40 * 4096 / dt), flush=True)
        eng.set_option("lds_pch", 0)
        eng.set_option("fused_4k", 1)
    if "timeline" in what:
        xd = torch.from_numpy(np.concatenate([base] * 8)).cuda()
        for idle in (0.0, 0.5):
            for _ in range(5):
                eng.search_batch_dev(sig, xd, items, dop, 1)
            torch.cuda.synchronize()
            time.sleep(idle)
            n = 300
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for i in range(n):
                eng.search_batch_dev(sig, xd, items, dop, 1)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
            print("timeline after %.1f s idle, ms per step: first 10 %s | 20-30 %s | 100-110 %s | last 10 %s | mean(20) %.4f mean(all) %.4f"
                  % (idle, np.round(ts[:10], 3), np.round(ts[20:30], 3), np.round(ts[100:110], 3), np.round(ts[-10:], 3), np.mean(ts[:20]), np.mean(ts)), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
