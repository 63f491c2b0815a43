import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gnss_dsp_tools_amd import acquire, signals, synth, codes
name = sys.argv[1]; sig = signals.get(name)
torch.cuda.init()
eng = acquire.Engine(0); eng.use_torch_stream()
for items in ([1], [1, 2], list(range(1, 33))):
    t0 = time.perf_counter(); eng._plan(sig, items); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(name, "signal build", len(items), "items: %.1f ms" % (1e3 * (t1 - t0)))
dop = acquire.doppler_grid([-1000.0, 1000.0, 500.0])
need = (1 + (1 if sig.pad else 0)) * sig.n
xd = torch.from_numpy(synth.make_epochs(sig, 1, 5, [], 1, nsamp=need)).cuda()
for tie in (0, 1):
    e2 = acquire.Engine(0); e2.use_torch_stream(); e2.set_option("tie_safe", tie)
    e2._plan(sig, [1, 2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); e2.search_batch_dev(sig, xd, [1, 2], dop, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    e2.search_batch_dev(sig, xd, [1, 2], dop, 1); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, "tie_safe", tie, "first search %.1f ms, second %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    e2.close()
