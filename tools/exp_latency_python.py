import sys, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np
from gnss_dsp_tools_amd import acquire, signals, synth, _native as nat
sig = signals.get("gps-l1"); items = list(range(1, 33)); ds = [-5000.0, 5000.0, 250.0]
x = synth.make_iq(sig, 1, 5, synth.default_sats(items), nsamp=4096)
eng = acquire.Engine(0)
for _ in range(20): eng.search_all(sig, x, items, ds, 1)
def med(fn, n=300):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
print("python search_all: %.1f us" % med(lambda: eng.search_all(sig, x, items, ds, 1)))
s, idx, bias = eng._plan(sig, items)
dop = acquire.doppler_grid(ds)
xc = np.ascontiguousarray(x[:4096], dtype=np.complex64)
res = (nat.Result * len(idx))()
xp = xc.ctypes.data_as(nat.c_float_p); ip = idx.ctypes.data_as(nat.c_int_p); dp = dop.ctypes.data_as(nat.c_double_p)
call = lambda: nat.lib.gacq_search(s._h, xp, len(xc), ip, len(idx), dp, len(dop), None, 1, res)
print("bare ctypes gacq_search: %.1f us" % med(call))
print("doppler_grid: %.1f us, _plan: %.1f us, ascontiguous: %.1f us, tuples: %.1f us" % (med(lambda: acquire.doppler_grid(ds)), med(lambda: eng._plan(sig, items)),
      med(lambda: np.ascontiguousarray(x[:4096], dtype=np.complex64)), med(lambda: [acquire._as_tuple(r) for r in res])))
eng.close()
