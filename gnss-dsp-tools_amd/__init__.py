"""gnss-dsp-tools_amd -- MI355X-native GNSS acquisition engine.

Drop-in for ONE hot path of pmonta/GNSS-DSP-tools: the FFT parallel code-phase search
``search(x, prn, doppler_search, ms) -> (metric, code, doppler)`` of every acquire-*.py
(acquire-gps-l1.py:18-40).  Host side is Python (like the reference); all arithmetic of the path
runs in hand-written HIP kernels + rocFFT behind the C ABI of include/gacq.h (libgacq.so).

  signals    descriptors of the 29 FFT search() variants (script stem -> knobs)
  codes      PRN chips / replicas from the native generators (bit-exact with gnsstools/*)
  acquire    Engine, search(), search_all(), result formatting
  sharded    PRN x Doppler grid sharding over one-process-per-GPU ranks (torch.distributed)
  synth      seeded synthetic IQ
"""
from . import signals  # noqa: F401

__all__ = ["signals", "codes", "acquire", "sharded", "synth"]
