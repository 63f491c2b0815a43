#!/usr/bin/env python3
"""Goldens for the tracking correlators from the reference's own correlate() functions (build container only)."""
import importlib
import json
import os
import sys
import warnings

import numpy as np

REF = os.environ.get("GNSS_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def main():
    cases = []
    rng_seed = 20250829 + 700
    plan = [("gps.ca", "ca_code", 5, 0, 512.37 - 0.05, 1.023e6 / 4.0e6 * (1 + 2e-6), 4000, False),
            ("gps.ca", "ca_code", 31, 0, -0.02, 1.023e6 / 5.456e6, 5456, False),
            ("beidou.b1i", "b1i_code", 12, 0, 2045.99, 2.046e6 / 8.0e6, 4096, False),
            ("gps.l5i", "l5i_code", 3, 0, 100.5, 10.23e6 / 25.0e6, 6000, False),
            ("gps.l1cd", "l1cd_code", 9, 0, 77.3, 1.023e6 / 8.0e6, 6000, True),
            ("galileo.e1b", "e1b_code", 11, 0, 4091.9, 1.023e6 / 8.184e6, 8184, True),
            ("galileo.e1c", "e1c_code", 24, 0, 17.25 + 0.2, 1.023e6 / 8.184e6, 5000, True),
            ("gps.l1cp", "l1cp_code", 2, 0, 3000.6, 1.023e6 / 8.0e6, 8000, True),
            ("gps.l2cm", "l2cm_code", 15, 0, 10229.7 - 0.5, 511500.0 / 4.0e6, 8000, False),
            ("gps.l2cl", "l2cl_code", 30, 120000, 0.25, 511500.0 / 4.0e6, 6000, False)]
    for k, (code, fn, prn, chips, frac, incr, n, has_boc) in enumerate(plan):
        mod = importlib.import_module("gnsstools." + code)
        rng = np.random.Generator(np.random.PCG64(rng_seed + k))
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        c = getattr(mod, fn)(prn)
        xx = x.astype(np.complex128)
        if has_boc:
            p = mod.correlate(xx, prn, chips, frac, incr, c, mod.boc11)
        else:
            p = mod.correlate(xx, prn, chips, frac, incr, c)
        cases.append({"code": code, "prn": prn, "chips": chips, "frac": frac, "incr": incr, "n": n, "seed": rng_seed + k,
                      "re": float(complex(p).real), "im": float(complex(p).imag)})
        print(code, prn, p)
    json.dump({"generator": "tools/make_goldens_tracking.py", "cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "tracking_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
