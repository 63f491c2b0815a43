"""File input of the acquire scripts: (ms + 5) ms of interleaved signed 8-bit I/Q (acquire-gps-l1.py:78-83,
gnsstools/io.py:3-12).  Everything after the read -- carrier-offset wipe-off, FIR low-pass, resample -- runs on the GPU
(csrc/gacq_frontend.hip through Engine.frontend_dev / gacq_longcode_search_int8); the numpy restatement the GPU front-end is
checked against lives with the other checkers in oracle/frontend_oracle.py."""
import numpy as np


def read_iq_int8(fp, n):
    """n complex samples as an int8 array [n, 2] (I, Q); None on a short read, like io.get_samples_complex (gnsstools/io.py:5-6)."""
    raw = fp.read(2 * n)
    if len(raw) != 2 * n:
        return None
    return np.frombuffer(raw, dtype=np.int8).reshape(n, 2)
