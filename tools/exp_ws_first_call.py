import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gnss_dsp_tools_amd import acquire, signals, synth
sig = signals.get("beidou-b2ad"); items = list(range(1, 64)); dop = acquire.doppler_grid([-7000.0, 7000.0, 200.0]); B = 80
need = (B + 1) * sig.n
xs = synth.make_epochs(sig, 1, 5, [], 1, nsamp=need)
xd = torch.from_numpy(xs).cuda()
for ws in (4 << 30, 32 << 30, 4 << 30, 32 << 30):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        eng = acquire.Engine(0, workspace_bytes=ws)
        eng.use_torch_stream()
        t1 = time.perf_counter()
        pk = eng.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pk2 = eng.search_batch_dev(sig, xd, items, dop, B)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        eng.close()
        t4 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    print("ws %d GiB: create %.1f ms, first search %.1f ms, second search %.1f ms, close %.1f ms" % ((ws >> 30,) + tuple(1e3 * min(t[i] for t in ts) for i in range(4))))
