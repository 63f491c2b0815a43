#!/usr/bin/env python3
"""Where the waves of the N = 16384 correlate kernel spend their cycles (diagnostic): needs a variant built with -DGACQ_PHASE_TIMING16,
   tools/build_variant.sh timing16 -DGACQ_PHASE_TIMING16 gacq_lds16k.hip ; python tools/variant.py timing16 tools/phase_timing16.py
(radix-32 form; the radix-16 form: build gacq_ldsfft.hip with the flag instead and pass lds_variant=16)
Prints shader-clock cycles per row, wave and phase (lane 0 of every wave, summed over all workgroups)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import _native as nat
from gnss_dsp_tools_amd import acquire, signals, synth

PHASES16 = ["item tail: reduction, C loads", "DMA wait + x read + C*x", "wave-private 1024-pt inverse (3 LDS round trips)", "barrier B", "gather + barrier A",
            "DMA issue (next row)", "last radix-16 pass + magnitudes", ""]
PHASES32 = ["item tail: reduction, C loads", "DMA wait + staged-row reads issued", "(unused)", "C*x + two DFT16 + transpose + table B + DFT32 + exchange stores",
            "barrier 1", "gather + barrier 2", "DMA issue + table A + last DFT32 + magnitudes", ""]


def main():
    fn = nat.lib.gacq_debug_phase16
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 128)()
    eng = acquire.Engine(0)
    eng.use_torch_stream()
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    PHASES = PHASES16 if eng.get_option("lds_variant") == 16 else PHASES32
    sig = signals.get("beidou-b1i")
    items = list(range(1, 64))
    dop = acquire.doppler_grid([-10000.0, 10000.0, 100.0])
    B = sig.blocks(10)
    x = synth.make_iq(sig, B, 99, synth.default_sats(items), nsamp=sig.samples_needed(B))
    xd = torch.from_numpy(x[None]).cuda()
    for _ in range(2):
        eng.search_batch_dev(sig, xd, items, dop, B)
    torch.cuda.synchronize()
    fn(buf, 1)
    reps = 3
    for _ in range(reps):
        eng.search_batch_dev(sig, xd, items, dop, B)
    torch.cuda.synchronize()
    fn(buf, 1)
    rows = reps * len(items) * len(dop) * B
    a = np.array(list(buf), dtype=np.float64).reshape(16, 8) / rows
    print("cycles per row, by wave (rows) and phase (columns); wave w sits on SIMD (0,2,1,3)[w % 4]")
    print("wave " + " ".join("%8d" % i for i in range(7)) + "    total")
    a = a[a.sum(1) > 0]
    for w in range(len(a)):
        print("%4d " % w + " ".join("%8.0f" % a[w, i] for i in range(7)) + " %8.0f" % a[w].sum())
    print("mean " + " ".join("%8.0f" % a[:, i].mean() for i in range(7)) + " %8.0f" % a.sum(1).mean())
    for i, name in enumerate(PHASES[:7]):
        print("   phase %d: %s" % (i, name))
    eng.close()


if __name__ == "__main__":
    main()
