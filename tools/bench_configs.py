#!/usr/bin/env python3
"""Throughput of the other BASELINE configs' shapes (parity-test cases, not the bench line): cells/s and the
stage-boundary A_pipe GB/s per signal, device-resident inputs.  usage: bench_configs.py [--engine N] [cfg ...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, signals, synth

S = 8


def a_pipe(N, P, D, B, F=1):
    return S * N * (4 * D * B * F + P + 5 * P * D * B) + 8 * N * P * D * (B - 1)


CASES = {
    # name: (signal, items, doppler_search, ms, epochs)
    "cfg2_gps_l1": ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 64),
    "cfg2_gps_l1_e1": ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 1),
    "gps_l1_ms10": ("gps-l1", list(range(1, 33)), [-10000.0, 10000.0, 100.0], 10, 4),
    "gps_l1_cli_ms80": ("gps-l1", list(range(1, 33)), [-7000.0, 7000.0, 200.0], 80, 1),      # acquire-gps-l1.py's own defaults (--time 80)
    # the other scripts at their own command-line defaults (--time 80: B = 80, E1B B = 80 // 4 - 1 = 19 with its 50 Hz grid)
    "gps_l5i_cli": ("gps-l5i", list(range(1, 33)), [-7000.0, 7000.0, 200.0], 80, 1),
    "b1i_cli": ("beidou-b1i", list(range(1, 64)), [-7000.0, 7000.0, 200.0], 80, 1),
    "glonass_cli": ("glonass-l1", list(range(-7, 8)), [-7000.0, 7000.0, 200.0], 80, 1),
    "e1b_cli": ("galileo-e1b", list(range(1, 51)), [-9000.0, 9000.0, 50.0], 80, 1),
    "b2ad_cli": ("beidou-b2ad", list(range(1, 64)), [-7000.0, 7000.0, 200.0], 80, 1),
    "cfg3_e1b": ("galileo-e1b", list(range(1, 37)), [-4000.0, 4000.0, 125.0], 8, 1),
    "cfg4_l5i": ("gps-l5i", list(range(1, 33)), [-7000.0, 7000.0, 200.0], 1, 1),
    "cfg4_b2ad_b1": ("beidou-b2ad", list(range(1, 64)), [-7000.0, 7000.0, 200.0], None, 1),   # engine-level B=1
    "cfg5_b1i": ("beidou-b1i", list(range(1, 64)), [-10000.0, 10000.0, 100.0], 10, 1),
    "cfg5_glonass": ("glonass-l1", list(range(-7, 8)), [-10000.0, 10000.0, 100.0], 10, 1),
    "gps_l1cd": ("gps-l1cd", list(range(1, 33)), [-7000.0, 7000.0, 100.0], 20, 1),
    "gps_l2cm": ("gps-l2cm", list(range(1, 33)), [-7000.0, 7000.0, 100.0], 60, 1),
    "gal_e6b": ("galileo-e6b", list(range(1, 51)), [-9000.0, 9000.0, 200.0], 4, 1),
    "cfg5_e1b": ("galileo-e1b", list(range(1, 51)), [-10000.0, 10000.0, 100.0], 10, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stages", action="store_true", help="also print per-stage HIP-event times")
    ap.add_argument("--ws", type=int, default=0, help="workspace limit in MiB (0 = library default)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="gacq_set_option tuning switch")
    ap.add_argument("cases", nargs="*")
    args = ap.parse_args()
    eng = acquire.Engine(0, engine=args.engine, workspace_bytes=(args.ws << 20) if args.ws else None)
    eng.use_torch_stream()
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    for cid in (args.cases or list(CASES)):
        name, items, ds, ms, E = CASES[cid]
        sig = signals.get(name)
        B = 1 if ms is None else sig.blocks(ms)
        dop = acquire.doppler_grid(ds)
        nsamp = sig.samples_needed(B)
        x = synth.make_iq(sig, B, 99, synth.default_sats(items), nsamp=nsamp)
        xd = torch.from_numpy(np.stack([x] * E)).cuda()
        F = len(items) if sig.bias_hz else 1
        for _ in range(2):
            eng.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            eng.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        cells = E * len(items) * len(dop) * sig.nfft
        ap_bytes = a_pipe(sig.nfft, len(items), len(dop), B, F) * E
        if args.stages:
            eng.set_profiling(True)
            eng.reset_stage_times()
            eng.search_batch_dev(sig, xd, items, dop, B)
            torch.cuda.synchronize()
            print("   stages:", {k: round(v[0], 4) for k, v in eng.stage_times().items() if v[1]})
            eng.set_profiling(False)
        print(json.dumps({"case": cid, "signal": name, "P": len(items), "D": len(dop), "B": B, "N": sig.nfft, "epochs": E,
                          "ms": dt * 1e3, "cells_per_s": cells / dt, "cell_blocks_per_s": cells * B / dt,
                          "a_pipe_GBps": ap_bytes / dt / 1e9, "frac_8TBps": ap_bytes / dt / 8e12}))
    eng.close()


if __name__ == "__main__":
    main()
