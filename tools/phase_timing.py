#!/usr/bin/env python3
"""Where a workgroup of split_inner_corr_kernel spends its cycles (diagnostic): needs a library built with -DGACQ_PHASE_TIMING
(gnss-dsp-tools_amd/build/timing/libgacq.so, see tools/phase_timing.sh) swapped into lib/.  Prints shader-clock cycles per row and
phase, averaged over all workgroups (thread 0 of each)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_dsp_tools_amd import _native as nat
from gnss_dsp_tools_amd import acquire, signals, synth

PHASES = {0: "loads of C (X cached) + C*conj(X) + DFT-11 + LDS writes", 1: "barrier", 2: "pass 2: LDS reads, twiddles, DFT-12", 3: "barrier",
          4: "pass 2: LDS writes", 5: "barrier", 6: "pass 3: LDS reads, twiddles, DFT-15", 7: "(last pass has no barrier)", 8: "row stores to global", 9: "barrier (end of row)"}


def main():
    fn = nat.lib.gacq_debug_phase_cycles
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 32)()
    eng = acquire.Engine(0)
    eng.use_torch_stream()
    if len(sys.argv) > 1:
        eng.set_option("split_dt", int(sys.argv[1]))
    print("split_dt option: %d (0 = library default)" % eng.get_option("split_dt"))
    for name, items in (("gps-l5i", list(range(1, 33))), ("beidou-b2ad", list(range(1, 64))), ("galileo-e6b", list(range(1, 51)))):
        sig = signals.get(name)
        dop = acquire.doppler_grid([-7000.0, 7000.0, 200.0])
        x = synth.make_iq(sig, 1, 99, synth.default_sats(items), nsamp=sig.samples_needed(1))
        xd = torch.from_numpy(x[None]).cuda()
        for _ in range(3):
            eng.search_batch_dev(sig, xd, items, dop, 1)
        torch.cuda.synchronize()
        fn(buf, 1)
        reps = 5
        for _ in range(reps):
            eng.search_batch_dev(sig, xd, items, dop, 1)
        torch.cuda.synchronize()
        fn(buf, 1)
        rows = reps * len(items) * len(dop) * 31
        tot = sum(buf[i] for i in range(32))
        print("%s: %d inner rows, %.0f cycles per row and workgroup" % (name, rows, tot / rows))
        for i in range(32):
            if buf[i]:
                print("   phase %2d %-62s %8.0f cycles  %5.1f %%" % (i, PHASES.get(i, ""), buf[i] / rows, 100.0 * buf[i] / tot))
    eng.close()


if __name__ == "__main__":
    main()
