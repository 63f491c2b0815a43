"""PRN chips and sampled replicas from the native generators (libgacq.so, host C++).

Mirror of the reference's per-signal modules: ``chips(code, prn)`` == ``<mod>.<sig>_code(prn)``
and ``replica(code, prn, n, boc)`` == ``<mod>.code(prn,0,0,L/n,n) [* nco.boc11(0,0,L/n,n)]``
(gnsstools/gps/ca.py:101-112, gnsstools/nco.py:12-19).  Needs no GPU.
"""
import ctypes

import numpy as np

from . import _native as nat


def names():
    return [nat.lib.gacq_code_name(i).decode() for i in range(nat.lib.gacq_code_count())]


def code_length(code):
    return nat.check(nat.lib.gacq_code_length(code.encode()))


def chip_rate(code):
    return nat.lib.gacq_code_chip_rate(code.encode())


def prns(code):
    buf = (ctypes.c_int * 1024)()
    n = nat.check(nat.lib.gacq_code_prns(code.encode(), buf, 1024))
    return list(buf[:n])


def chips(code, prn):
    L = code_length(code)
    out = np.empty(L, dtype=np.uint8)
    nat.check(nat.lib.gacq_code_chips(code.encode(), int(prn), out.ctypes.data_as(nat.c_uint8_p), L))
    return out


def replica(code, prn, n, boc=False):
    out = np.empty(int(n), dtype=np.float32)
    nat.check(nat.lib.gacq_code_replica(code.encode(), int(prn), int(n), int(bool(boc)), out.ctypes.data_as(nat.c_float_p)))
    return out
