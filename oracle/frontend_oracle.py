"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy/scipy restatement of the front-end every acquire script runs on the file
before search() (acquire-gps-l1.py:78-96): carrier-offset wipe-off with the fixed-point table NCO (gnsstools/nco.py:30-41),
161-tap Hann low-pass applied forward-backward (scipy.signal.filtfilt), linear-interpolation resample to the internal rate.
Pinned by tests/golden/cli_gps_l1.json ("conditioned": the same steps evaluated with the reference's own io/nco primitives,
tools/make_goldens_cli.py).  The product runs these steps on the GPU (csrc/gacq_frontend.hip); only tests import this file."""
import numpy as np
import scipy.signal

NT = 1024
_TABLE = np.exp(2 * np.pi * 1j * np.arange(NT) * (1.0 / NT))


def iq_to_complex(iq_int8):
    """[n, 2] int8 -> complex64 (gnsstools/io.py:7-11)."""
    s = np.asarray(iq_int8, dtype=np.int8).reshape(-1, 2)
    x = np.empty(len(s), dtype=np.complex64)
    x.real = s[:, 0]
    x.imag = s[:, 1]
    return x


def mix_fixed_point(x, f, p=0.0):
    """x[i] *= table[(phase_i >> 50) & 1023], phase_i = floor(p*1024*2^50) + i*floor(f*1024*2^50) in int64
    (wrapping, exactly like the reference's integer accumulator, gnsstools/nco.py:30-41)."""
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
