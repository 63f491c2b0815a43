#!/usr/bin/env python3
"""End-to-end time of the acquire-<name>.py replacements at the scripts' own DEFAULT arguments on a recording-sized input: int8 IQ
file at 69.984 MS/s (the reference's example rate), default --time 80 (85 ms read), default item list, default Doppler grid --
file read, H2D, GPU front-end, search, formatted lines.  The first call of a process pays library / context / code-spectrum
start-up (reported separately).  Shapes at the defaults:
  gps-l1      32 PRNs x  70 bins x 80 blocks x N = 4096           (acquire-gps-l1.py:52-56)
  beidou-b2ad 63 PRNs x  70 bins x 80 blocks x N = 61380 padded   (acquire-beidou-b2ad.py:29: B = 80 whatever --time says)
  gps-l1cd    32 PRNs x 700 bins x  8 blocks x N = 81920, BOC     (acquire-gps-l1cd.py:19: one block per 10 ms; 20 Hz grid)
  gps-l2cm    32 PRNs x 700 bins x  3 blocks x N = 163840 padded  (acquire-gps-l2cm.py:19: one block per 20 ms, two spare)
usage (GPU box): tools/bench_cli.py [name ...]"""
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gnss_dsp_tools_amd import acquire, cli, codes, signals

FS, MS = 69984000.0, 80
CASES = {"gps-l1": -9334875.0, "beidou-b2ad": 0.0, "gps-l1cd": -9334875.0, "gps-l2cm": 0.0}      # name -> carrier offset argument


def recording(name, coff):
    """noise + one strong satellite of the signal's own code (plain BPSK; enough for the unpadded/padded plain searches to lock,
    BOC signals just see it as a slightly mismatched replica)"""
    sig = signals.get(name)
    n = int(FS * 0.001 * (MS + 5))
    rng = np.random.default_rng(7)
    t = np.arange(n) / FS
    x = 20.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    prn = codes.prns(sig.code)[2]
    c = 1.0 - 2.0 * codes.chips(sig.code, prn).astype(np.float64)
    rate = len(c) * sig.fs / sig.n                                           # chips per second (code length per coherent block)
    idx = np.floor((t * rate - 300.25) % len(c)).astype(np.int64)
    x += 6.0 * c[idx] * np.exp(2j * np.pi * (coff + 1250.0) * t)
    iq = np.empty(2 * n, dtype=np.int8)
    iq[0::2] = np.clip(np.round(x.real), -127, 127)
    iq[1::2] = np.clip(np.round(x.imag), -127, 127)
    return iq, prn


def main():
    for name in (sys.argv[1:] or list(CASES)):
        coff = CASES[name]
        sig = signals.get(name)
        iq, prn = recording(name, coff)
        with tempfile.NamedTemporaryFile(suffix=".iq", delete=False) as f:
            iq.tofile(f)
            path = f.name
        try:
            times, lines = [], None
            for _ in range(4):
                t0 = time.perf_counter()
                lines = cli.run(name, [path, str(FS), str(coff)], out=io.StringIO())
                times.append(time.perf_counter() - t0)
        finally:
            os.unlink(path)
        items = acquire.parse_list_ranges(sig.default_items, sep=sig.item_sep) if sig.default_items else codes.prns(sig.code)
        dop = acquire.doppler_grid(list(sig.default_doppler))
        B = max(sig.blocks(MS), 0)
        cells = len(items) * len(dop) * sig.nfft
        best = max(lines, key=lambda l: float(l.split()[5]))
        print(json.dumps({"case": "acquire-%s FILE %.0f %.0f (85 ms of int8 IQ, all defaults)" % (name, FS, coff), "items": len(items),
                          "doppler_bins": len(dop), "blocks": B, "lags": sig.nfft, "cells": cells, "cell_blocks": cells * B,
                          "file_bytes": int(iq.nbytes), "first_call_s": times[0], "steady_call_ms": 1e3 * min(times[1:]),
                          "cell_blocks_per_s_end_to_end": cells * B / min(times[1:]), "injected_prn": int(prn), "strongest_line": best}), flush=True)


if __name__ == "__main__":
    main()
