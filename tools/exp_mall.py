#!/usr/bin/env python3
"""Does the Infinity Cache (256 MB) serve a reader data that a producer kernel has just written?  Fill a buffer of S bytes, read it back,
for S from 16 MiB to 2 GiB (gacq_stream_probe kinds fill / read / fill_read): if written lines were retained, fill + read of a small
buffer would run well above the large-buffer rate.  Decides whether an in-launch producer/consumer ring for the split engines' Z'
round trip could take the read side off HBM."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_dsp_tools_amd import acquire

eng = acquire.Engine(0)
for mib in (16, 32, 64, 128, 192, 256, 512, 2048):
    n = mib << 20
    reps = max(4, min(200, (8 << 30) // n))
    r = {k: eng.stream_probe(k, n, reps) for k in ("fill", "read", "fill_read", "fill_plain", "read_plain", "fill_read_plain")}
    # time of the read half of a fill + read pair, from the pair rate and the fill-only rate
    for sfx in ("", "_plain"):
        t_pair, t_fill = 2.0 * n / (r["fill_read" + sfx] * 1e9), n / (r["fill" + sfx] * 1e9)
        r["read_after_fill%s_GBps" % sfx] = n / max(t_pair - t_fill, 1e-9) / 1e9
    print(json.dumps({"MiB": mib, **{k: round(v, 1) for k, v in r.items()}}), flush=True)
eng.close()
