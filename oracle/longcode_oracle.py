"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy (fp64) restatement of the reference's time-domain long-code searches
(acquire-gps-l2cl.py:15-30, acquire-glonass-l1-p.py:15-33, acquire-glonass-l2-p.py:15-33).  Pinned by
tests/golden/longcode_cases.json (tools/make_goldens_longcode.py ran the reference's own search() functions)."""
import numpy as np

from . import acq_oracle, codes_oracle


def _code(chips01, chips, frac, incr, n):
    """<sig>.code(prn, chips, frac, incr, n)   (gnsstools/gps/l2cl.py:57-63, glonass/p.py:27-32)."""
    L = len(chips01)
    idx = (chips % L) + frac + incr * np.arange(n)
    idx = np.mod(np.floor(idx).astype('int'), L)
    return 1.0 - 2.0 * chips01[idx].astype(np.float64)


def search_l2cl(x, prn, doppler, l2cm_code_phase, ms, fs):
    c01 = codes_oracle.chips("gps.l2cl", prn)
    blocks = ms // 20
    n = int(fs * 0.020)
    w = acq_oracle.nco(-doppler / fs, 0, n)
    incr = 511500 / fs
    m_metric, m_k = 0, 0
    for k in range(75):
        q = 0
        for block in range(blocks):
            c = _code(c01, (k + block) * 10230 + l2cm_code_phase, 0, incr, n)
            q = q + np.absolute(np.sum(x[n * block:n * (block + 1)] * c * w))
        if q > m_metric:
            m_metric, m_k = q, k
    return m_metric, m_k


def search_glonass_p(x, chan, doppler, ca_code_phase, ms, fs, band="l1"):
    c01 = codes_oracle.chips("glonass.p", 0)
    spacing = {"l1": 562500, "l2": 437500}[band]
    blocks = ms // 4
    n = int(fs * 0.004)
    w = acq_oracle.nco(-(spacing * chan + doppler) / fs, 0, n)
    m_metric, m_k = 0, 0
    for k in range(1000):
        q = 0
        cp = 5110 * k + 10 * ca_code_phase
        for block in range(blocks):
            incr = 5110000.0 / fs
            c = _code(c01, 0, cp, incr, n)
            q = q + np.absolute(np.sum(x[n * block:n * (block + 1)] * c * w))
            cp += n * incr
        if q > m_metric:
            m_metric, m_k = q, k
    return m_metric, m_k
