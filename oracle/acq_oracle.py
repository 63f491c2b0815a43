"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

numpy (fp64) restatement of the reference's FFT parallel code-phase acquisition search -- the
function ``search(x, prn, doppler_search, ms)`` that every acquire-*.py defines inline
(acquire-gps-l1.py:18-40 and its 27 variants) -- folded into ONE
function with the six knobs that distinguish the variants.  Loop order, dtypes, tie rules
and the order of floating-point operations follow the reference statement by statement so
that results agree to fp64 round-off (~1e-15).

Parity status: PINNED.  tools/make_goldens.py imports the real reference ``search()`` of nine
structurally distinct scripts in the build container and stores (metric, code, doppler) and
full ``q`` rows under tests/golden/; tests/test_oracle_golden.py holds this file to them.
The FFT itself is third-party arithmetic (scipy.fftpack -> pocketfft; version not pinned by
the reference, README:10); scipy 1.15.3 produced the goldens and scipy.fft is used here.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import scipy.fft as _fft

TABLE_SIZE = 1024
# gnsstools/nco.py:3-4 -- 1024-entry complex exponential table
PHASOR_TABLE = np.exp(2 * (np.pi) * (1j) * np.arange(TABLE_SIZE) * (1.0 / TABLE_SIZE))


def nco_indices(f, p, n):
    """Table indices of nco.nco(f,p,n): floor((p + f*i)*1024) mod 1024 (gnsstools/nco.py:6-9)."""
    ph = p + f * np.arange(n)
    k = np.floor(ph * TABLE_SIZE).astype('int')
    return np.mod(k, TABLE_SIZE)


def nco(f, p, n):
    """gnsstools/nco.py:6-10."""
    return PHASOR_TABLE[nco_indices(f, p, n)]


def boc11(chips, frac, incr, n):
    """gnsstools/nco.py:12-19 -- square subcarrier, first half chip -1, second +1."""
    levels = np.array([-1, 1])
    t = (chips % 2) + frac + incr * np.arange(n)
    k = np.floor(t * 2).astype('int')
    return levels[np.mod(k, 2)]


def sample_code(chips01, chips, frac, incr, n):
    """<sig>.code(prn,chips,frac,incr,n): nearest-chip (floor) resampling to +-1.0
    (gnsstools/gps/ca.py:106-112; identical body in every signal module)."""
    L = len(chips01)
    t = (chips % L) + frac + incr * np.arange(n)
    k = np.mod(np.floor(t).astype('int'), L)
    return 1.0 - 2.0 * np.asarray(chips01, dtype=np.float64)[k]


def doppler_grid(doppler_search):
    """np.arange(min,max,incr): half-open grid (acquire-gps-l1.py:26)."""
    lo, hi, step = doppler_search
    return np.arange(lo, hi, step)


def code_spectrum(chips01, n, pad, boc):
    """c = fft(code [*boc] [++ zeros(n)])   (acquire-gps-l1.py:22-24, acquire-galileo-e1b.py:23-26,
    acquire-beidou-b1i.py:22-24)."""
    L = len(chips01)
    incr = float(L) / n
    c = sample_code(chips01, 0, 0, incr, n)
    if boc:
        c = c * boc11(0, 0, incr, n)
    if pad:
        c = np.concatenate((c, np.zeros(n)))
    return _fft.fft(c)


def accumulate_row(x, C, f, n, pad, blocks):
    """q for one Doppler bin: sum over blocks of |ifft(C * conj(fft(x_block * w)))|
    (acquire-gps-l1.py:27-33 / acquire-beidou-b1i.py:27-33)."""
    span = 2 * n if pad else n
    q = np.zeros(span)
    w = nco(f, 0, span)
    for b in range(blocks):
        seg = x[(b * n):(b * n + span)]
        seg = seg * w
        r = _fft.ifft(C * np.conj(_fft.fft(seg)))
        q = q + np.absolute(r)
    return q


def search(x, chips01, doppler_search, blocks, *, fs, n, pad=False, boc=False, normalised=False,
           fold=False, bias_hz=0.0):
    """The reference search() with the variant knobs exposed.

    chips01: {0,1} chips of the PRN; blocks: number of non-coherent blocks (the ms->B rule is
    applied by the caller exactly like each script does); bias_hz: carrier bias added to the
    Doppler inside the NCO (GLONASS: 562500*chan, acquire-glonass-l1.py:28).
    Returns (m_metric, m_code, m_doppler) -- ints (0,0,0) when no bin beats metric 0,
    as in the reference (acquire-gps-l1.py:25,40)."""
    L = len(chips01)
    C = code_spectrum(chips01, n, pad, boc)
    m_metric, m_code, m_doppler = 0, 0, 0
    for doppler in doppler_grid(doppler_search):
        if bias_hz:
            f = -(bias_hz + doppler) / fs
        else:
            f = -doppler / fs
        q = accumulate_row(x, C, f, n, pad, blocks)
        idx = np.argmax(q)
        metric = q[idx] / np.mean(q) if normalised else q[idx]
        if metric > m_metric:
            m_metric = metric
            m_code = L * (float(idx) / n)
            m_doppler = doppler
    if fold:
        m_code = m_code % L
    return m_metric, m_code, m_doppler


def search_row(x, chips01, doppler, blocks, *, fs, n, pad=False, boc=False, bias_hz=0.0):
    """Full accumulated magnitude row q for one Doppler value (debugging aid / golden rows)."""
    C = code_spectrum(chips01, n, pad, boc)
    f = -(bias_hz + doppler) / fs if bias_hz else -doppler / fs
    return accumulate_row(x, C, f, n, pad, blocks)


# Variant knobs per reference script (restated from the search() bodies, lines 18-42 of each).
#   name: (code module, fs, n, pad, boc, normalised, fold, blocks(ms), bias multiplier)
VARIANTS = {
    "gps-l1":       ("gps.ca",        4096000.0,  4096, False, False, True,  False, lambda ms: ms, 0.0),
    "xona-x1":      ("xona.x1p",      4096000.0,  4096, False, False, True,  False, lambda ms: ms, 0.0),
    "xona-x5p":     ("xona.x5p",      30690000.0, 30690, False, False, True, False, lambda ms: ms, 0.0),
    "glonass-l1":   ("glonass.ca",    16384000.0, 16384, False, False, False, False, lambda ms: ms, 562500.0),
    "glonass-l2":   ("glonass.ca",    16384000.0, 16384, False, False, False, False, lambda ms: ms, 437500.0),
    "gps-l1cd":     ("gps.l1cd",      8192000.0,  81920, False, True,  False, True,  lambda ms: ms // 10, 0.0),
    "gps-l1cp":     ("gps.l1cp",      8192000.0,  81920, False, True,  False, True,  lambda ms: ms // 10, 0.0),
    "beidou-b1cd":  ("beidou.b1cd",   8192000.0,  81920, False, True,  False, True,  lambda ms: ms // 10, 0.0),
    "beidou-b1cp":  ("beidou.b1cp",   8192000.0,  81920, False, True,  False, True,  lambda ms: ms // 10, 0.0),
    "galileo-e1b":  ("galileo.e1b",   8192000.0,  32768, True,  True,  False, True,  lambda ms: ms // 4 - 1, 0.0),
    "galileo-e1c":  ("galileo.e1c",   8192000.0,  32768, True,  True,  False, True,  lambda ms: ms // 4 - 1, 0.0),
    "beidou-b1i":   ("beidou.b1i",    8192000.0,  8192,  True,  False, False, True,  lambda ms: ms, 0.0),
    "beidou-b2i":   ("beidou.b1i",    8192000.0,  8192,  True,  False, False, True,  lambda ms: ms, 0.0),
    "gps-l2cm":     ("gps.l2cm",      4096000.0,  81920, True,  False, False, True,  lambda ms: ms // 20 - 1, 0.0),
    "gps-l5i":      ("gps.l5i",       30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "gps-l5q":      ("gps.l5q",       30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "galileo-e5ai": ("galileo.e5ai",  30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "galileo-e5aq": ("galileo.e5aq",  30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "galileo-e5bi": ("galileo.e5bi",  30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "galileo-e5bq": ("galileo.e5bq",  30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "beidou-b2ad":  ("beidou.b2ad",   30690000.0, 30690, True,  False, False, True,  lambda ms: 80, 0.0),
    "beidou-b2ap":  ("beidou.b2ap",   30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "beidou-b2bi":  ("beidou.b2bi",   30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "beidou-b2bq":  ("beidou.b2bq",   30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "beidou-b3i":   ("beidou.b3i",    30690000.0, 30690, True,  False, False, True,  lambda ms: ms, 0.0),
    "glonass-l3ocd": ("glonass.l3ocd", 30690000.0, 30690, True, False, False, True,  lambda ms: ms, 0.0),
    "glonass-l3ocp": ("glonass.l3ocp", 30690000.0, 30690, True, False, False, True,  lambda ms: ms, 0.0),
    "galileo-e6b":  ("galileo.e6b",   15345000.0, 15345, True,  False, False, True,  lambda ms: ms, 0.0),
    "galileo-e6c":  ("galileo.e6c",   15345000.0, 15345, True,  False, False, True,  lambda ms: ms, 0.0),
}


def search_script(name, x, item, doppler_search, ms, chips01=None):
    """search(x, prn_or_chan, doppler_search, ms) of acquire-<name>.py."""
    from . import codes_oracle
    code, fs, n, pad, boc, normalised, fold, blocks, bias = VARIANTS[name]
    if chips01 is None:
        chips01 = codes_oracle.chips(code, 0 if bias else item)
    return search(x, chips01, doppler_search, blocks(ms), fs=fs, n=n, pad=pad, boc=boc,
                  normalised=normalised, fold=fold, bias_hz=(bias * item if bias else 0.0))


def search_script_blocks(name, x, item, doppler_search, blocks, chips01=None):
    """search() of acquire-<name>.py with the block count B given directly instead of through the script's ms -> B rule
    (BASELINE config 4 runs B2aD at B = 1; the script itself fixes B = 80, acquire-beidou-b2ad.py:29)."""
    from . import codes_oracle
    code, fs, n, pad, boc, normalised, fold, _, bias = VARIANTS[name]
    if chips01 is None:
        chips01 = codes_oracle.chips(code, 0 if bias else item)
    return search(x, chips01, doppler_search, blocks, fs=fs, n=n, pad=pad, boc=boc,
                  normalised=normalised, fold=fold, bias_hz=(bias * item if bias else 0.0))
