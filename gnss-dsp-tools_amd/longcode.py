"""Time-domain long-code searches that follow an FFT acquisition (SURVEY.md section 8f "next #3"):

    acquire-gps-l2cl.py:15-30        search(x, prn, doppler, l2cm_code_phase, ms)  -> (metric, k)     75 candidates
    acquire-glonass-l1-p.py:15-33    search(x, chan, doppler, ca_code_phase, ms)   -> (metric, k)   1000 candidates
    acquire-glonass-l2-p.py:15-33    same with 437.5 kHz channel spacing

The reference reads `fs` from a module global; here it is an explicit argument.  Per candidate k and block the start code
phase is formed on the host in fp64 exactly like the reference's expression, everything per-sample runs on the GPU
(gacq_longcode_search).  x is the complex input at the file rate after the carrier-offset wipe-off (nco.mix) -- or, with
`coffset=`, the file's raw int8 I/Q, wiped off on the GPU as well (gacq_longcode_search_int8; what the CLI uses)."""
import ctypes

import numpy as np

from . import _native as nat
from . import acquire

L2CL_LENGTH = 767250
P_LENGTH = 5110000


def _run(engine, x, fs, code, prn, carrier_hz, phase0, blocks, n, coffset=None):
    """coffset None: x is complex (the caller already wiped off the carrier offset, as the reference's search() expects);
    coffset given: x is the file's raw int8 I/Q [nsamp, 2] and the wipe-off nco.mix(x,-coffset/fs,0) runs on the GPU."""
    eng = engine or acquire.default_engine()
    K = phase0.shape[0]
    if blocks <= 0:
        return np.zeros(K)
    if hasattr(x, "is_cuda") and x.is_cuda:
        # complex64 samples already on the GPU (Engine.mix_int8_dev / frontend output): only the start phases travel
        import torch
        if not (x.dtype == torch.complex64 and x.dim() == 1 and x.is_contiguous()) or coffset is not None:
            raise ValueError("device input must be a contiguous 1-D complex64 CUDA tensor, already wiped off (coffset=None)")
        if x.numel() < blocks * n:
            raise ValueError("operands could not be broadcast together: search needs %d samples, x has %d" % (blocks * n, x.numel()))
        ph = np.ascontiguousarray(phase0, dtype=np.float64)
        q = np.empty(K, dtype=np.float64)
        nat.check(nat.lib.gacq_longcode_search_dev(eng._ctx, ctypes.c_void_p(x.data_ptr()), x.numel(), float(fs), code.encode(), int(prn),
                                                   float(carrier_hz), ph.ctypes.data_as(nat.c_double_p), K, int(blocks), int(n),
                                                   q.ctypes.data_as(nat.c_double_p)), eng._ctx)
        return q
    x = np.asarray(x)
    if len(x) < blocks * n:
        raise ValueError("operands could not be broadcast together: search needs %d samples, x has %d" % (blocks * n, len(x)))
    ph = np.ascontiguousarray(phase0, dtype=np.float64)
    q = np.empty(K, dtype=np.float64)
    if coffset is None:
        xc = np.ascontiguousarray(x[:blocks * n], dtype=np.complex64)
        nat.check(nat.lib.gacq_longcode_search(eng._ctx, xc.ctypes.data_as(nat.c_float_p), len(xc), float(fs), code.encode(), int(prn),
                                               float(carrier_hz), ph.ctypes.data_as(nat.c_double_p), K, int(blocks), int(n),
                                               q.ctypes.data_as(nat.c_double_p)), eng._ctx)
    else:
        if x.dtype != np.int8 or x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("raw input must be an int8 array [nsamp, 2]")
        xi = np.ascontiguousarray(x[:blocks * n])
        nat.check(nat.lib.gacq_longcode_search_int8(eng._ctx, ctypes.c_void_p(xi.ctypes.data), len(xi), float(fs), float(coffset),
                                                    code.encode(), int(prn), float(carrier_hz), ph.ctypes.data_as(nat.c_double_p), K,
                                                    int(blocks), int(n), q.ctypes.data_as(nat.c_double_p)), eng._ctx)
    return q


def _best(q):
    """strict '>' scan from (0, 0): first maximum wins, all-zero keeps the initial ints (acquire-gps-l2cl.py:20,27-29).
    np.argmax returns the first maximum, which is what the scan keeps; a NaN in q (never greater than anything) takes the scan itself."""
    q = np.asarray(q)
    if q.size == 0:
        return 0, 0
    if np.isnan(q).any():
        m_metric, m_k = 0, 0
        for k, v in enumerate(q):
            if v > m_metric:
                m_metric, m_k = v, k
        return m_metric, m_k
    k = int(np.argmax(q))
    return (q[k], k) if q[k] > 0 else (0, 0)


def l2cl_start_phases(l2cm_code_phase, blocks):
    """chips = (k + block) * 10230 + l2cm_code_phase (acquire-gps-l2cl.py:24); start phase (chips % code_length) + frac with frac = 0
    (gps/l2cl.py:59).  Formed for all (k, block) at once: the same operations in the same order per element as the reference's scalar
    loop (integer product, one addition, np.mod = Python's floored %), so the values are bit for bit the loop's
    (tests/test_longcode.py::test_start_phase_tables_equal_the_scalar_loops)."""
    kb = np.arange(75, dtype=np.int64)[:, None] + np.arange(max(blocks, 0), dtype=np.int64)[None, :]
    chips = kb * 10230 + l2cm_code_phase
    return np.ascontiguousarray(np.mod(chips, L2CL_LENGTH) + 0, dtype=np.float64)


def search_l2cl(x, prn, doppler, l2cm_code_phase, ms, fs, engine=None, coffset=None):
    blocks = ms // 20
    n = int(fs * 0.020)
    phase0 = l2cl_start_phases(l2cm_code_phase, blocks)
    return _best(_run(engine, x, fs, "gps.l2cl", prn, doppler, phase0, blocks, n, coffset))


def glonass_p_start_phases(ca_code_phase, blocks, n, incr):
    """cp = 5110 k + 10 ca_code_phase, then cp += n * incr block after block (acquire-glonass-l1-p.py:24-31; p.code(0, cp, incr, n):
    chips = 0, frac = cp): one column of the (k, block) table per block, every element going through the scalar loop's sequence of
    fp64 additions."""
    phase0 = np.empty((1000, max(blocks, 0)), dtype=np.float64)
    cp = 5110 * np.arange(1000, dtype=np.int64) + 10 * ca_code_phase
    cp = cp.astype(np.float64) if cp.dtype != np.float64 else cp
    step = n * incr
    for block in range(blocks):
        phase0[:, block] = (0 % P_LENGTH) + cp
        cp = cp + step
    return phase0


def search_glonass_p(x, chan, doppler, ca_code_phase, ms, fs, band="l1", engine=None, coffset=None):
    spacing = {"l1": 562500, "l2": 437500}[band]                    # acquire-glonass-l1-p.py:18 / -l2-p.py:18
    blocks = ms // 4
    n = int(fs * 0.004)
    incr = 5110000.0 / fs
    phase0 = glonass_p_start_phases(ca_code_phase, blocks, n, incr)
    return _best(_run(engine, x, fs, "glonass.p", 0, spacing * chan + doppler, phase0, blocks, n, coffset))
