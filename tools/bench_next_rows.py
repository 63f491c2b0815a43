#!/usr/bin/env python3
"""Timing of the SURVEY 8f "next" rows that have no bench line of their own, with the numpy oracle timed on the same
inputs (test infrastructure: the oracle is only the CPU yardstick here, exactly as in bench.py's cpu_baseline):

  long-code searches   acquire-gps-l2cl.py (75 candidates, 20 ms blocks), acquire-glonass-l1-p.py (1000 candidates, 4 ms blocks)
  tracking correlators early/prompt/late of 12 satellites over one 1 ms block (track-gps-l1.py:48-50), one launch

Host-buffer entry points (H2D + kernels + D2H per call), like the reference's call surface."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gnss_dsp_tools_amd import acquire, longcode, synth, tracking
from oracle import longcode_oracle, tracking_oracle


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def main():
    eng = acquire.Engine(0)
    rows = []

    # ---- L2CL: fs = 4.096 MHz (the rate acquire-gps-l2cl.py resamples to), 100 ms -> 5 blocks of 81920 samples
    fs, ms = 4096000.0, 100
    n = int(fs * 0.020)
    x = synth.make_longcode_iq("gps.l2cl", 7, 511500.0, longcode.L2CL_LENGTH, fs, (ms // 20) * n, 11, 0.4, 1234.0, 37 * 10230 + 4000.25)
    t_gpu, got = timeit(lambda: longcode.search_l2cl(x, 7, 1234.0, 4000.25, ms, fs, engine=eng), 10)
    t_cpu, want = timeit(lambda: longcode_oracle.search_l2cl(x.astype(np.complex128), 7, 1234.0, 4000.25, ms, fs), 1)
    rows.append({"case": "gps-l2cl 75 candidates x 5 blocks x 81920 samples", "gpu_ms": t_gpu * 1e3, "cpu_oracle_ms": t_cpu * 1e3,
                 "speedup": t_cpu / t_gpu, "k": [int(got[1]), int(want[1])], "metric_rel_err": abs(got[0] - want[0]) / want[0],
                 "candidate_samples_per_s": 75 * (ms // 20) * n / t_gpu})

    # ---- GLONASS P: fs = 16.384 MHz, 20 ms -> 5 blocks of 65536 samples, 1000 candidates
    fs, ms = 16384000.0, 20
    n = int(fs * 0.004)
    x = synth.make_longcode_iq("glonass.p", 0, 5110000.0, longcode.P_LENGTH, fs, (ms // 4) * n, 12, 0.4, 562500.0 * 2 + 800.0,
                               5110 * 321 + 10 * 100.5)
    t_gpu, got = timeit(lambda: longcode.search_glonass_p(x, 2, 800.0, 100.5, ms, fs, engine=eng), 5)
    t_cpu, want = timeit(lambda: longcode_oracle.search_glonass_p(x.astype(np.complex128), 2, 800.0, 100.5, ms, fs), 1)
    rows.append({"case": "glonass-l1-p 1000 candidates x 5 blocks x 65536 samples", "gpu_ms": t_gpu * 1e3, "cpu_oracle_ms": t_cpu * 1e3,
                 "speedup": t_cpu / t_gpu, "k": [int(got[1]), int(want[1])], "metric_rel_err": abs(got[0] - want[0]) / want[0],
                 "candidate_samples_per_s": 1000 * (ms // 4) * n / t_gpu})

    # ---- tracking: E/P/L of 12 GPS L1 C/A satellites over one 1 ms block at 4.096 MHz
    fs = 4096000.0
    nb = 4096
    rng = np.random.default_rng(5)
    xb = (rng.standard_normal(nb) + 1j * rng.standard_normal(nb)).astype(np.complex64)
    prns = np.arange(1, 13)
    code_p = rng.uniform(0, 1023, 12)
    cf = 1.023e6 / fs * (1 + rng.uniform(-2e-6, 2e-6, 12))
    t_gpu, got = timeit(lambda: tracking.early_prompt_late("gps.ca", xb, prns, code_p, cf, 0.05, engine=eng), 200)

    def cpu():
        out = np.empty((12, 3), dtype=np.complex128)
        for i, p in enumerate(prns):
            for j, off in enumerate((-0.05, 0.0, 0.05)):
                out[i, j] = tracking_oracle.correlate("gps.ca", xb.astype(np.complex128), int(p), 0.0, code_p[i] + off, cf[i])
        return out
    t_cpu, want = timeit(cpu, 5)
    rows.append({"case": "tracking E/P/L x 12 satellites, 4096-sample block, one launch", "gpu_us": t_gpu * 1e6, "cpu_oracle_us": t_cpu * 1e6,
                 "note": "the oracle is the interpreted loop; the reference JIT-compiles the same loop with numba when it is installed "
                         "(not in this image), so cpu_oracle_us overstates the reference's cost by roughly two orders of magnitude", "max_rel_err": float(np.max(np.abs(got - want)) / np.max(np.abs(want)))})
    for r in rows:
        print(json.dumps(r))
    eng.close()


if __name__ == "__main__":
    main()
