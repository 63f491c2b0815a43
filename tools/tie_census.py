#!/usr/bin/env python3
"""Near-tie census on the GPU box (evidence, not a pytest): large batches of NOISE-ONLY epochs -- every search is decided between
near-equal candidates -- through the default fp32 engines and through engine 5 (complex128 on the device).  Counts, per signal
shape, the searches, the pairs the tie-safe machinery found ambiguous / re-evaluated, the peak locations that differ from the
complex128 engine with tie-safe locations ON (must be 0) and, on the same samples, with it OFF (what fp32 alone gets wrong).
Noise is generated on the device (torch, seeded) so that millions of searches fit into a minute.
usage: tools/tie_census.py [seconds per shape] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import acquire, codes, signals

# (signal, items, Doppler search, blocks, epochs per batch)
SHAPES = [("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 1024),       # BASELINE config 2: fused 4096 kernel
          ("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 500.0], 4, 128),        # two-kernel LDS path, B = 4
          ("xona-x1", None, [-3000.0, 3000.0, 250.0], 2, 128),                     # None: the first eight PRNs of the code table
          ("beidou-b1i", list(range(1, 64)), [-5000.0, 5000.0, 500.0], 2, 16),     # N = 16384 correlate kernel, raw metric
          ("glonass-l1", list(range(-7, 8)), [-5000.0, 5000.0, 500.0], 2, 32),     # N = 16384 fused kernel, FDMA
          ("galileo-e1b", list(range(1, 37)), [-2000.0, 2000.0, 250.0], 1, 8),     # engine 4
          ("gps-l5i", list(range(1, 33)), [-2000.0, 2000.0, 200.0], 1, 8),         # engine 3, M = 1980
          ("galileo-e6b", list(range(1, 19)), [-2000.0, 2000.0, 400.0], 2, 8),     # engine 3, M = 990, B = 2
          ("gps-l1cd", list(range(1, 17)), [-1000.0, 1000.0, 250.0], 1, 4)]        # engine 4, R = 20


def peaks(t):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(acquire.PEAK_DTYPE).reshape(-1).copy()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    eng = acquire.Engine(0)
    eng.use_torch_stream()
    ver = acquire.Engine(0, engine=5)
    ver.use_torch_stream()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    total = {"searches": 0, "flips_tie_safe_on": 0, "flips_tie_safe_off": 0}
    for name, items, ds, B, E in SHAPES:
        sig = signals.get(name)
        if items is None:
            items = [int(p) for p in codes.prns(sig.code)[:8]]
        dop = acquire.doppler_grid(ds)
        nsamp = sig.samples_needed(B)
        t0 = time.time()
        n = on = off = 0
        worst = 0.0
        before = eng.tie_stats()
        while time.time() - t0 < budget:
            xd = torch.view_as_complex(torch.randn((E, nsamp, 2), generator=gen, device="cuda", dtype=torch.float32).contiguous())
            ref = peaks(ver.search_batch_dev(sig, xd, items, dop, B))
            eng.set_option("tie_safe", 1)
            a = peaks(eng.search_batch_dev(sig, xd, items, dop, B))
            eng.set_option("tie_safe", 0)
            b = peaks(eng.search_batch_dev(sig, xd, items, dop, B))
            eng.set_option("tie_safe", 1)
            n += ref.size
            on += int(((a["idx"] != ref["idx"]) | (a["d_index"] != ref["d_index"])).sum())
            off += int(((b["idx"] != ref["idx"]) | (b["d_index"] != ref["d_index"])).sum())
            worst = max(worst, float((np.abs(a["metric"] - ref["metric"]) / ref["metric"]).max()))
        st = eng.tie_stats()
        row = {"signal": name, "nfft": sig.nfft, "items": len(items), "bins": len(dop), "blocks": B, "searches": n,
               "ambiguous_pairs": st["ambiguous_pairs"] - before["ambiguous_pairs"], "rows_reevaluated": st["rows_reevaluated"] - before["rows_reevaluated"],
               "kept_fp32": st["kept_fp32"] - before["kept_fp32"], "locations_changed_by_reevaluation": st["locations_changed"] - before["locations_changed"],
               "location_mismatches_vs_complex128_tie_safe_on": on, "location_mismatches_vs_complex128_tie_safe_off": off,
               "worst_rel_metric_err": worst, "seconds": round(time.time() - t0, 1)}
        print(json.dumps(row), flush=True)
        total["searches"] += n
        total["flips_tie_safe_on"] += on
        total["flips_tie_safe_off"] += off
    print(json.dumps(total))
    eng.close()
    ver.close()
    sys.exit(1 if total["flips_tie_safe_on"] else 0)


if __name__ == "__main__":
    main()
