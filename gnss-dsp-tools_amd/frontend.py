"""Front-end of the acquire scripts (host side, numpy/scipy): int8 IQ file -> complex samples at the signal's
internal rate.  This is component C10 of SURVEY.md -- it runs once per file, is NOT part of the search hot
path, and is "next #1" in section 8f (to be moved to the GPU after the hot path meets its bar).  It is here so that
the CLI shim is a complete drop-in for `acquire-<name>.py file fs coffset`.

Steps and their reference lines (acquire-gps-l1.py):
  read (ms+5) ms of interleaved int8 I/Q            :78-83, gnsstools/io.py:3-12
  wipe off the carrier offset with the table NCO    :87,    gnsstools/nco.py:30-41 (50-bit fixed-point phase)
  161-tap Hann low-pass, forward-backward           :92-93
  linear-interpolation resample to the internal fs  :94-96
"""
import numpy as np
import scipy.signal

NT = 1024
_TABLE = np.exp(2 * np.pi * 1j * np.arange(NT) * (1.0 / NT))


def read_iq_int8(fp, n):
    """n complex samples from interleaved signed 8-bit I/Q; None on a short read (gnsstools/io.py:3-12)."""
    raw = fp.read(2 * n)
    if len(raw) != 2 * n:
        return None
    s = np.frombuffer(raw, dtype=np.int8).reshape(n, 2)
    x = np.empty(n, dtype=np.complex64)
    x.real = s[:, 0]
    x.imag = s[:, 1]
    return x


def mix_fixed_point(x, f, p=0.0):
    """x[i] *= table[(phase_i >> 50) & 1023], phase_i = floor(p*1024*2^50) + i*floor(f*1024*2^50) in int64
    (wrapping, exactly like the reference's integer accumulator)."""
    dp = np.int64(int(np.floor(p * NT * (1 << 50))))
    df = np.int64(int(np.floor(f * NT * (1 << 50))))
    with np.errstate(over="ignore"):
        ph = dp + np.arange(len(x), dtype=np.int64) * df
    # the reference multiplies in place into the complex64 sample array: every product is rounded to fp32
    return (x * _TABLE[(ph >> 50) & (NT - 1)]).astype(x.dtype if np.iscomplexobj(x) else np.complex128)


def condition(x, fs, coffset, sig, ms_pad):
    """Carrier wipe-off, low-pass and resample to the signal's internal rate; returns ms_pad ms of complex128."""
    x = mix_fixed_point(x, -coffset / fs, 0)
    per_ms = int(round(sig.fs * 0.001))
    fsr = sig.fs / fs
    h = scipy.signal.firwin(161, sig.fir_cutoff / (fs / 2), window='hann')
    x = scipy.signal.filtfilt(h, [1], x)
    t = (1 / fsr) * np.arange(ms_pad * per_ms)
    src = np.arange(len(x))
    return np.interp(t, src, np.real(x)) + 1j * np.interp(t, src, np.imag(x))
