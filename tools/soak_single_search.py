"""20 s of back-to-back 32-PRN search_all calls on four alternating inputs through the latency path (BAR upload, result watching):
every result must equal the first one for that input, host RSS must stay flat.  usage (GPU box): tools/soak_single_search.py"""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, psutil
from gnss_dsp_tools_amd import acquire, signals, synth
sig = signals.get("gps-l1"); items = list(range(1, 33)); ds = [-5000.0, 5000.0, 250.0]
xs = [synth.make_iq(sig, 1, 5 + k, synth.default_sats(items), nsamp=4096) for k in range(4)]
eng = acquire.Engine(0)
ref = [eng.search_all(sig, x, items, ds, 1) for x in xs]
p = psutil.Process(os.getpid()); r0 = p.memory_info().rss
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 20.0:
    for k, x in enumerate(xs):
        got = eng.search_all(sig, x, items, ds, 1)
        assert got == ref[k], (n, k)
        n += 1
dt = time.perf_counter() - t0
print("%d calls in %.1f s (%.1f us per call), all equal to the first results; RSS %d -> %d MiB" % (n, dt, dt / n * 1e6, r0 >> 20, p.memory_info().rss >> 20))
eng.close()
