#!/usr/bin/env python3
"""Randomised differential test on the GPU box (evidence, not a pytest): random (signal, item list, Doppler grid, blocks, epochs,
workspace limit, tuning switches) draws through
  (a) the default engine with default switches and a roomy workspace,
  (b) the same engine with random switches (items / Doppler bins per workgroup, fused kernels on/off, teams) and a random small
      workspace (group chunks, epoch chunks, Doppler slices)            -> must equal (a) byte for byte,
  (c) the complex128 verification pipeline (engine 5)                   -> the SAME peak location, always (tie-safe locations:
      near-tied candidates are re-evaluated in complex128 on the device, gacq_tiesafe.hip), metric within 2e-6.
usage: tools/fuzz_engines.py [seconds] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, codes, signals, synth

WEIGHT = {4096: 6, 16384: 4, 30690: 3, 61380: 3, 65536: 2, 81920: 1, 163840: 1}


def draw(rng):
    names = sorted(signals.SIGNALS)
    w = np.array([WEIGHT[signals.get(n).nfft] for n in names], dtype=float)
    sig = signals.get(names[rng.choice(len(names), p=w / w.sum())])
    pool = acquire.parse_list_ranges(sig.default_items, sig.item_sep) if sig.default_items else codes.prns(sig.code)      # '' = every PRN in the table
    big = sig.nfft > 20000
    P = int(rng.integers(1, 4 if big else 9))
    items = [int(pool[i]) for i in rng.integers(0, len(pool), P)]            # duplicates allowed
    D = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 13, 16, 24] if not big else [1, 2, 3, 5, 8, 9, 13]))
    lo = float(rng.integers(-40, 40)) * 125.0
    inc = float(rng.choice([25.0, 100.0, 200.0, 250.0]))
    dop = lo + inc * np.arange(D)
    B = int(rng.choice([1, 1, 2, 3])) if sig.nfft < 100000 else 1
    E = int(rng.choice([1, 2, 3, 5] if not big else [1, 2]))
    opts = {"lds_pch": int(rng.choice([0, 1, 3, 8])), "split_pch": int(rng.choice([0, 1, 3, 5])), "fused_4k": int(rng.choice([0, 1, 2])),
            "fused_16k": int(rng.choice([0, 1])), "fused_inner": int(rng.choice([0, 1, 1, 1])), "split_mfma": int(rng.choice([0, 0, 1])),
            "split_dt": int(rng.choice([0, 1, 2, 3])), "lds_ugroup": int(rng.choice([0, 1, 2, 3]))}
    row = 8 * sig.nfft * B
    ws = int(rng.choice([row // 2, 3 * row, 3 * row * D + 7 * row, 64 * row * D]))
    seed = int(rng.integers(1, 1 << 30))
    sats = [(items[0], 0.4, float(dop[min(D - 1, D // 2)]) + 37.0, int(rng.integers(0, sig.n)))] if rng.random() < 0.7 else []
    return sig, items, dop, B, E, opts, max(ws, 1 << 20), seed, sats


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 12345))
    ref = acquire.Engine(0)
    ref.use_torch_stream()
    ver = acquire.Engine(0, engine=5)
    ver.use_torch_stream()
    t0 = time.time()
    n = bugs = flips = searches = 0
    worst = 0.0
    by_n = {}
    while time.time() - t0 < budget:
        sig, items, dop, B, E, opts, ws, seed, sats = draw(rng)
        desc = {"signal": sig.name, "items": items, "dopplers": [float(dop[0]), float(dop[-1]), len(dop)], "B": B, "E": E, "opts": opts, "ws": ws, "seed": seed}
        xs = synth.make_epochs(sig, B, seed, sats, E, nsamp=sig.samples_needed(B))
        xd = torch.from_numpy(xs).cuda()
        torch.cuda.synchronize()
        a = ref.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        a = a.cpu().numpy().copy()
        alt = acquire.Engine(0, workspace_bytes=ws)
        alt.use_torch_stream()
        try:
            for k, v in opts.items():
                alt.set_option(k, v)
            b = alt.search_batch_dev(sig, xd, items, dop, B)
            torch.cuda.synchronize()
            b = b.cpu().numpy().copy()
        finally:
            alt.close()
        c = ver.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        c = c.cpu().numpy().copy()
        n += 1
        searches += E * len(items)
        by_n[sig.nfft] = by_n.get(sig.nfft, 0) + 1
        if sig.nfft % 31 == 0 and (not opts["fused_inner"] or opts["split_mfma"]):
            # Cooley-Tukey form with rocFFT inner transforms instead of the prime-factor form, or its DFT-31 on the matrix pipe: other
            # arithmetic, same answers within rounding
            qa, qb = a.view(acquire.PEAK_DTYPE).reshape(-1), b.view(acquire.PEAK_DTYPE).reshape(-1)
            relb = np.abs(qa["metric"] - qb["metric"]) / np.maximum(np.abs(qa["metric"]), 1e-30)
            moved = (qa["idx"] != qb["idx"]) | (qa["d_index"] != qb["d_index"])
            if (relb > 2e-6).any() or (moved & (relb > 1e-6)).any():
                bugs += 1
                print("MISMATCH prime-factor form vs its other forms:", float(relb.max()), json.dumps(desc), flush=True)
        elif a.tobytes() != b.tobytes():
            bugs += 1
            print("MISMATCH default vs switches/workspace:", json.dumps(desc), flush=True)
        pa, pc = a.view(acquire.PEAK_DTYPE).reshape(-1), c.view(acquire.PEAK_DTYPE).reshape(-1)
        rel = np.abs(pa["metric"] - pc["metric"]) / np.maximum(np.abs(pc["metric"]), 1e-30)
        same = (pa["idx"] == pc["idx"]) & (pa["d_index"] == pc["d_index"])
        worst = max(worst, float(rel[same].max()) if same.any() else 0.0)
        if (rel[same] > 2e-6).any():
            bugs += 1
            print("METRIC off vs complex128:", float(rel[same].max()), json.dumps(desc), flush=True)
        for k in np.nonzero(~same)[0]:
            flips += 1
            bugs += 1
            print("LOCATION differs vs complex128 (rel metric difference %.3g):" % rel[k], json.dumps(desc), flush=True)
    print(json.dumps({"draws": n, "searches": searches, "seconds": round(time.time() - t0, 1), "draws_by_fft_length": by_n, "failures": bugs,
                      "near_tie_location_flips_vs_complex128": flips, "worst_rel_metric_err_vs_complex128": worst,
                      "tie_safe": ref.tie_stats()}))
    ref.close()
    ver.close()
    sys.exit(1 if bugs else 0)


if __name__ == "__main__":
    main()
