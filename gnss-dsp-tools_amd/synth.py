"""Seeded synthetic IQ for tests, goldens and the benchmark.

    x = noise + sum_k a_k * code_k[(i - delay_k) mod n] * exp(2 pi j (bias_k + f_k) i / fs)

noise = complex standard normal (sigma = 1 per component) from numpy's PCG64; up to four
"satellites" with amplitudes {0.5, 0.35, 0.25, 0.18}, off-grid Dopplers {1537, -3262, 4111, -409} Hz
and delays {1201, 77, 3000, 2222} samples (mod n); everything else is noise-only, which is what
exercises the near-tie behaviour of argmax / strict '>'.  The replica uses the same floor sampling
as the reference's code() (+ BOC(1,1) where the signal has it).  Output is complex64 -- feed the
oracle ``x.astype(complex128)`` so both sides see identical sample values.
"""
import numpy as np

from . import codes
from . import signals as _signals

AMPLITUDES = (0.5, 0.35, 0.25, 0.18)
DOPPLERS_HZ = (1537.0, -3262.0, 4111.0, -409.0)
DELAYS = (1201, 77, 3000, 2222)
BASE_SEED = 20250829


def default_sats(items):
    """Pick up to four items spread over the list: (item, amplitude, doppler_hz, delay_samples)."""
    items = list(items)
    if not items:
        return []
    picks = []
    for k in range(min(4, len(items))):
        it = items[(k * len(items)) // 4 + (len(items) // 8 if len(items) >= 8 else 0)]
        if it not in picks:
            picks.append(it)
    return [(it, AMPLITUDES[k], DOPPLERS_HZ[k], DELAYS[k]) for k, it in enumerate(picks)]


def make_iq(name, blocks, seed, sats, nsamp=None, dtype=np.complex64):
    """One epoch of samples at the signal's internal rate, long enough for `blocks` blocks (+1 spare block)."""
    sig = _signals.get(name) if isinstance(name, str) else name
    n = sig.n
    if nsamp is None:
        nsamp = (blocks + 1) * n + n
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)
    i = np.arange(nsamp)
    for item, amp, dop, delay in sats:
        prn = 0 if sig.bias_hz else item
        rep = codes.replica(sig.code, prn, n, sig.boc).astype(np.float64)
        carrier_hz = (sig.bias_hz * item if sig.bias_hz else 0.0) + dop
        x += amp * rep[(i - (delay % n)) % n] * np.exp(2j * np.pi * carrier_hz * i / sig.fs)
    return x.astype(dtype)


def make_epochs(name, blocks, seed, sats, nepoch, nsamp=None):
    """[nepoch, nsamp] complex64: independent noise per epoch, same satellites."""
    return np.stack([make_iq(name, blocks, seed + 1000 * e, sats, nsamp) for e in range(nepoch)])


def make_longcode_iq(code, prn, chip_rate, L, fs, nsamp, seed, amp, carrier_hz, start_chips):
    """Input for the time-domain long-code searches: noise + amp * code(start_chips + chip_rate/fs * i) * carrier, complex64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)
    i = np.arange(nsamp)
    c = codes.chips(code, prn)
    idx = np.mod(np.floor(start_chips + (chip_rate / fs) * i).astype(np.int64), L)
    x += amp * (1.0 - 2.0 * c[idx]) * np.exp(2j * np.pi * carrier_hz * i / fs)
    return x.astype(np.complex64)
