"""Signal descriptors: the knobs that distinguish the FFT ``search()`` variants of the reference.

One entry per ``acquire-<name>.py`` script that uses the FFT parallel code-phase search
(SURVEY.md section 2.2).  Every number below is hard-coded inside the corresponding reference
``search()`` (lines 18-42 of each script) or its option defaults / output format (main body):

  fs, n            acquire-gps-l1.py:19-20           pad      acquire-beidou-b1i.py:24,30
  boc              acquire-galileo-e1b.py:25-26      metric   acquire-gps-l1.py:35 vs b1i.py:36
  fold             acquire-beidou-b1i.py:39          blocks   acquire-galileo-e1b.py:19,
                                                              acquire-gps-l1cd.py:19, acquire-gps-l2cm.py:19,
                                                              acquire-beidou-b2ad.py:29
  bias_hz          acquire-glonass-l1.py:28 (562500*chan), acquire-glonass-l2.py:28 (437500*chan)
  fmt              the worker() format string, e.g. acquire-gps-l1.py:103
"""
from dataclasses import dataclass
from typing import Callable, Optional, Tuple


@dataclass(frozen=True)
class Signal:
    name: str                 # script stem: acquire-<name>.py
    code: str                 # reference code module, "gps.ca" == gnsstools/gps/ca.py
    fs: float                 # internal sampling rate (Hz)
    n: int                    # samples per coherent block
    pad: bool                 # zero-padded replica, N = 2n, overlapping windows
    boc: bool                 # BOC(1,1) subcarrier on the replica
    normalised: bool          # metric = max/mean (True) or raw max (False)
    fold: bool                # code phase reported modulo code_length
    blocks: Callable[[int], int]   # ms -> number of non-coherent blocks B
    fmt: str                  # result line format (item, doppler, metric, code)
    default_items: str        # default --prn / --channel
    default_doppler: Tuple[float, float, float]
    fir_cutoff: float         # front-end low-pass cutoff (Hz), out of the hot path
    bias_hz: float = 0.0      # per-item carrier bias multiplier (GLONASS FDMA channel spacing)
    item_sep: str = "-"       # range separator of the item list option
    item_opt: str = "--prn"

    @property
    def nfft(self) -> int:
        return 2 * self.n if self.pad else self.n

    def samples_needed(self, blocks: int) -> int:
        """x must cover block windows x[b*n:(b+1)*n] (or x[b*n:(b+2)*n] when padded)."""
        return (blocks + (1 if self.pad else 0)) * self.n if blocks > 0 else 0


def _ms(ms):
    return ms


_F_PRN3_NORM = 'prn %3d doppler % 7.1f metric % 5.2f code_offset %6.1f'
_F_PRN3 = 'prn %3d doppler % 7.1f metric % 7.1f code_offset %6.1f'
_F_PRN2 = 'prn %2d doppler % 7.1f metric % 7.1f code_offset %6.1f'
_F_CHAN = 'chan % 2d doppler % 7.1f metric % 7.1f code_offset %7.2f'

_D200 = (-7000.0, 7000.0, 200.0)
_D20 = (-7000.0, 7000.0, 20.0)
_G200 = (-9000.0, 9000.0, 200.0)
_G50 = (-9000.0, 9000.0, 50.0)
_X200 = (-50000.0, 50000.0, 200.0)

_FS_10 = 3 * 10230000.0      # acquire-gps-l5i.py:19
_N_10 = 3 * 10230            # acquire-gps-l5i.py:20


def _sig10(name, code, fmt, items, dop=_D200, blocks=_ms):
    """10.23 Mcps family: fs = 30.69 MS/s, n = 30690, padded, raw metric, folded code phase."""
    return Signal(name, code, _FS_10, _N_10, True, False, False, True, blocks, fmt, items, dop, 12e6)


SIGNALS = {s.name: s for s in [
    # --- unpadded, normalised (a1) -------------------------------------------------------------
    Signal("gps-l1", "gps.ca", 4096000.0, 4096, False, False, True, False, _ms, _F_PRN3_NORM, "1-32", _D200, 1.5e6),
    Signal("xona-x1", "xona.x1p", 4096000.0, 4096, False, False, True, False, _ms, _F_PRN3_NORM, "0", _X200, 1.5e6),
    Signal("xona-x5p", "xona.x5p", _FS_10, _N_10, False, False, True, False, _ms, _F_PRN3_NORM, "0", _X200, 12e6),
    # --- GLONASS FDMA (a4): item = frequency channel, one shared code ---------------------------
    Signal("glonass-l1", "glonass.ca", 16384000.0, 16384, False, False, False, False, _ms, _F_CHAN, "-7:7", _D200, 6e6,
           bias_hz=562500.0, item_sep=":", item_opt="--channel"),
    Signal("glonass-l2", "glonass.ca", 16384000.0, 16384, False, False, False, False, _ms, _F_CHAN, "-7:7", _D200, 6e6,
           bias_hz=437500.0, item_sep=":", item_opt="--channel"),
    # --- BOC(1,1), unpadded 10 ms (a3) ------------------------------------------------------------
    Signal("gps-l1cd", "gps.l1cd", 8192000.0, 81920, False, True, False, True, lambda ms: ms // 10, _F_PRN3, "1-32", _D20, 4e6),
    Signal("gps-l1cp", "gps.l1cp", 8192000.0, 81920, False, True, False, True, lambda ms: ms // 10, _F_PRN3, "1-32", _D20, 4e6),
    Signal("beidou-b1cd", "beidou.b1cd", 8192000.0, 81920, False, True, False, True, lambda ms: ms // 10, _F_PRN3, "1-63", _D20, 4e6),
    Signal("beidou-b1cp", "beidou.b1cp", 8192000.0, 81920, False, True, False, True, lambda ms: ms // 10, _F_PRN3, "1-63", _D20, 4e6),
    # --- BOC(1,1), padded 4 ms (a3) -----------------------------------------------------------------
    Signal("galileo-e1b", "galileo.e1b", 8192000.0, 32768, True, True, False, True, lambda ms: ms // 4 - 1, _F_PRN2, "1-50", _G50, 4e6),
    Signal("galileo-e1c", "galileo.e1c", 8192000.0, 32768, True, True, False, True, lambda ms: ms // 4 - 1, _F_PRN2, "1-50", _G50, 4e6),
    # --- padded, raw (a2) -----------------------------------------------------------------------------
    Signal("beidou-b1i", "beidou.b1i", 8192000.0, 8192, True, False, False, True, _ms, _F_PRN2, "1-63", _D200, 3e6),
    Signal("beidou-b2i", "beidou.b1i", 8192000.0, 8192, True, False, False, True, _ms, _F_PRN3, "1-63", _D200, 3e6),
    Signal("gps-l2cm", "gps.l2cm", 4096000.0, 81920, True, False, False, True, lambda ms: ms // 20 - 1, _F_PRN3, "1-32", _D20, 1.5e6),
    _sig10("gps-l5i", "gps.l5i", _F_PRN2, "1-32"),
    _sig10("gps-l5q", "gps.l5q", _F_PRN2, "1-32"),
    _sig10("galileo-e5ai", "galileo.e5ai", _F_PRN2, "1-50", _G200),
    _sig10("galileo-e5aq", "galileo.e5aq", _F_PRN2, "1-50", _G200),
    _sig10("galileo-e5bi", "galileo.e5bi", _F_PRN2, "1-50", _G200),
    _sig10("galileo-e5bq", "galileo.e5bq", _F_PRN2, "1-50", _G200),
    _sig10("beidou-b2ad", "beidou.b2ad", _F_PRN3, "1-63", blocks=lambda ms: 80),   # B fixed, ms ignored
    _sig10("beidou-b2ap", "beidou.b2ap", _F_PRN3, "1-63"),
    _sig10("beidou-b2bi", "beidou.b2bi", _F_PRN3, ""),       # default: every PRN in the table
    _sig10("beidou-b2bq", "beidou.b2bq", _F_PRN3, ""),
    _sig10("beidou-b3i", "beidou.b3i", _F_PRN2, "1-63"),
    _sig10("glonass-l3ocd", "glonass.l3ocd", _F_PRN2, "0-63"),
    _sig10("glonass-l3ocp", "glonass.l3ocp", _F_PRN2, "0-63"),
    Signal("galileo-e6b", "galileo.e6b", 3 * 5115000.0, 3 * 5115, True, False, False, True, _ms, _F_PRN2, "1-50", _G200, 6e6),
    Signal("galileo-e6c", "galileo.e6c", 3 * 5115000.0, 3 * 5115, True, False, False, True, _ms, _F_PRN2, "1-50", _G200, 6e6),
]}


def get(name: str) -> Signal:
    """Look a signal up by script stem ('gps-l1'), tolerating 'acquire-gps-l1.py' / 'acquire_gps_l1'."""
    key = name.lower().replace("_", "-")
    if key.startswith("acquire-"):
        key = key[len("acquire-"):]
    if key.endswith(".py"):
        key = key[:-3]
    if key not in SIGNALS:
        raise KeyError("unknown signal %r; known: %s" % (name, ", ".join(sorted(SIGNALS))))
    return SIGNALS[key]
