#!/usr/bin/env python3
"""What would the LDS-resident complex128 row kernels (tie_recheck_split_kernel<4|16>) do as a full engine 5 for N = 16384 / 65536?
Every row of a small search is forced through the re-evaluation (tie_eps_ppb = 1e9, a list that holds them all) and the extra time over the plain
fp32 search is compared with the rocFFT double-precision pipeline (engine 5) on the same search.  usage (GPU box): python tools/exp_c128_split_rate.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnss_dsp_tools_amd import acquire, signals, synth  # noqa: E402


def timed(eng, sig, xd, items, dop, B, reps=5):
    for _ in range(2):
        eng.search_batch_dev(sig, xd, items, dop, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.search_batch_dev(sig, xd, items, dop, B)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, items, nd, B in (("glonass-l1", list(range(-3, 5)), 24, 10), ("beidou-b1i", list(range(1, 9)), 24, 10), ("galileo-e1b", list(range(1, 9)), 24, 1)):
    sig = signals.get(name)
    dop = np.linspace(-3000.0, 3000.0, nd)
    need = (B + (1 if sig.pad else 0)) * sig.n
    xs = synth.make_epochs(sig, 1, 77, [], 1, nsamp=need)
    xd = torch.from_numpy(xs).cuda()
    rows = len(items) * nd
    out = {"signal": name, "N": sig.nfft, "rows": rows, "blocks": B}
    e = acquire.Engine(0)
    e.use_torch_stream()
    out["fp32_ms"] = timed(e, sig, xd, items, dop, B)
    e.set_option("tie_eps_ppb", 1000000000)
    e.set_option("tie_cap", rows)
    out["fp32_plus_all_rows_reevaluated_ms"] = timed(e, sig, xd, items, dop, B)
    st = e.tie_stats()
    out["rows_reevaluated_per_call"] = st["rows_reevaluated"] / 7.0
    e.close()
    e5 = acquire.Engine(0, engine=5)
    e5.use_torch_stream()
    out["engine5_rocfft_double_ms"] = timed(e5, sig, xd, items, dop, B)
    e5.close()
    out["reevaluation_ms"] = out["fp32_plus_all_rows_reevaluated_ms"] - out["fp32_ms"]
    out["ratio_pipeline_over_row_kernels"] = out["engine5_rocfft_double_ms"] / out["reevaluation_ms"]
    print(json.dumps(out))
