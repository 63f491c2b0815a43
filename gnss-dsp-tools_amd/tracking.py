"""Batched code correlators for the tracking hand-off (SURVEY.md section 8f "next #4").

Mirror of the reference's per-signal ``<module>.correlate(x, prn, chips, frac, incr, c[, boc11])``
(gnsstools/gps/ca.py:120-128 and the BOC / CBOC / TMBOC / RZ variants), evaluated for many (PRN, start phase,
rate) triples over one block of samples in a single launch -- e.g. early/prompt/late of every tracked satellite.
The feedback loops themselves (track-*.py) are sequential and are not part of this package."""
import numpy as np

from . import _native as nat
from . import acquire

# which subcarrier the reference's correlate() of each code module applies
KIND = {"gps.l1cd": 1, "beidou.b1cd": 1, "beidou.b1cp": 1,          # boc11[int(bp)]           gps/l1cd.py:101-112
        "galileo.e1b": 2, "galileo.e1c": 2,                          # CBOC                      galileo/e1b.py:45-58
        "gps.l1cp": 3,                                               # TMBOC                     gps/l1cp.py:210-228
        "gps.l2cm": 4, "gps.l2cl": 5}                                # RZ [1,0] / [0,1]          gps/l2cm.py:81-92, l2cl.py


def _is_device_block(x):
    """a 1-D contiguous complex64 torch tensor on the GPU (e.g. a slice of Engine.frontend_dev's output)"""
    return hasattr(x, "is_cuda") and x.is_cuda


def correlate_batch(code, x, prns, chips, frac, incr, engine=None):
    """K correlators over the same block x: returns complex128[K].  prns/chips/frac/incr broadcast to a common length.
    x: numpy complex block (staged with the specs in one copy), or a complex64 torch tensor ALREADY on the GPU -- the front-end's
    output or the samples an acquisition just searched -- in which case only the K specs travel (gacq_correlate_batch_dev)."""
    eng = engine or acquire.default_engine()
    prns, chips, frac, incr = np.broadcast_arrays(np.atleast_1d(prns), np.atleast_1d(chips), np.atleast_1d(frac), np.atleast_1d(incr))
    K = len(prns)
    if _is_device_block(x):
        import ctypes
        import torch
        if not (x.dtype == torch.complex64 and x.dim() == 1 and x.is_contiguous()):
            raise ValueError("device block must be a contiguous 1-D complex64 CUDA tensor")
        p = np.ascontiguousarray(prns, dtype=np.int32)
        c = np.ascontiguousarray(chips, dtype=np.float64)
        f = np.ascontiguousarray(frac, dtype=np.float64)
        r = np.ascontiguousarray(incr, dtype=np.float64)
        out = np.empty(K, dtype=np.complex128)
        nat.check(nat.lib.gacq_correlate_batch_dev(eng._ctx, ctypes.c_void_p(x.data_ptr()), x.numel(), code.encode(), KIND.get(code, 0),
                                                   p.ctypes.data_as(nat.c_int_p), c.ctypes.data_as(nat.c_double_p),
                                                   f.ctypes.data_as(nat.c_double_p), r.ctypes.data_as(nat.c_double_p), K,
                                                   out.ctypes.data_as(nat.c_double_p)), eng._ctx)
        return out
    xc = np.ascontiguousarray(x, dtype=np.complex64)
    p = np.ascontiguousarray(prns, dtype=np.int32)
    c = np.ascontiguousarray(chips, dtype=np.float64)
    f = np.ascontiguousarray(frac, dtype=np.float64)
    r = np.ascontiguousarray(incr, dtype=np.float64)
    out = np.empty(K, dtype=np.complex128)
    nat.check(nat.lib.gacq_correlate_batch(eng._ctx, xc.ctypes.data_as(nat.c_float_p), len(xc), code.encode(), KIND.get(code, 0),
                                           p.ctypes.data_as(nat.c_int_p), c.ctypes.data_as(nat.c_double_p),
                                           f.ctypes.data_as(nat.c_double_p), r.ctypes.data_as(nat.c_double_p), K,
                                           out.ctypes.data_as(nat.c_double_p)), eng._ctx)
    return out


def correlate(code, x, prn, chips, frac, incr, engine=None):
    """One correlator, the reference's argument order: <module>.correlate(x, prn, chips, frac, incr, ...)."""
    return complex(correlate_batch(code, x, [prn], [chips], [frac], [incr], engine)[0])


def early_prompt_late(code, x, prns, code_p, cf, spacing, engine=None):
    """E/P/L of several satellites in one launch: the three correlate() calls of track-gps-l1.py:48-50 (spacing 0.05),
    track-galileo-e1b.py:41-43 (0.2), track-gps-l2cm.py:41-43 (0.5).  Returns complex128[len(prns), 3]."""
    prns = np.atleast_1d(prns)
    code_p = np.broadcast_to(np.asarray(code_p, dtype=np.float64), prns.shape)
    cf = np.broadcast_to(np.asarray(cf, dtype=np.float64), prns.shape)
    off = np.array([-spacing, 0.0, spacing])
    out = correlate_batch(code, x, np.repeat(prns, 3), 0.0, (code_p[:, None] + off[None, :]).ravel(), np.repeat(cf, 3), engine)
    return out.reshape(len(prns), 3)


class EplPlan:
    """Early/prompt/late of a fixed satellite set, called once per block by a tracking loop (track-gps-l1.py:48-50): the PRN list, the
    tap spacing and every array the C call needs are laid out once, a call only writes the 3 K start phases and rates in place and
    hands the block over -- numpy's per-call broadcasting / repeating / pointer conversions cost as much as the GPU work itself on a
    resident 1 ms block (tools/exp_epl_latency.py).  x: a complex64 torch CUDA tensor (resident block) or a numpy block."""

    def __init__(self, code, prns, spacing, engine=None):
        import ctypes
        self.eng = engine or acquire.default_engine()
        self.code = code.encode()
        self.kind = KIND.get(code, 0)
        self.n = len(prns)
        self._off = np.array([-spacing, 0.0, spacing])
        self._prn = np.ascontiguousarray(np.repeat(np.asarray(prns, dtype=np.int32), 3))
        self._chips = np.zeros(3 * self.n)
        self._frac = np.empty((self.n, 3))
        self._incr = np.empty((self.n, 3))
        self._out = np.empty((self.n, 3), dtype=np.complex128)
        self._args = (self._prn.ctypes.data_as(nat.c_int_p), self._chips.ctypes.data_as(nat.c_double_p), self._frac.ctypes.data_as(nat.c_double_p),
                      self._incr.ctypes.data_as(nat.c_double_p), 3 * self.n, self._out.ctypes.data_as(nat.c_double_p))
        self._void = ctypes.c_void_p

    def __call__(self, x, code_p, cf):
        """complex128 [len(prns), 3] (a view of the plan's own result buffer: copy it to keep it across calls)."""
        np.add(np.asarray(code_p, dtype=np.float64).reshape(-1, 1), self._off, out=self._frac)
        self._incr[:] = np.asarray(cf, dtype=np.float64).reshape(-1, 1)
        if _is_device_block(x):
            rc = nat.lib.gacq_correlate_batch_dev(self.eng._ctx, self._void(x.data_ptr()), x.numel(), self.code, self.kind, *self._args)
        else:
            xc = np.ascontiguousarray(x, dtype=np.complex64)
            rc = nat.lib.gacq_correlate_batch(self.eng._ctx, xc.ctypes.data_as(nat.c_float_p), len(xc), self.code, self.kind, *self._args)
        if rc < 0:
            nat.check(rc, self.eng._ctx)
        return self._out
