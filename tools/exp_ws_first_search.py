import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gnss_dsp_tools_amd import acquire, signals, synth
name = sys.argv[1]; sig = signals.get(name); items = list(range(1, int(sys.argv[2]) + 1)); dop = acquire.doppler_grid([-7000.0, 7000.0, 200.0]); B = 80
xs = synth.make_epochs(sig, 1, 5, [], 1, nsamp=(B + 1) * sig.n)
xd = torch.from_numpy(xs).cuda(); torch.cuda.synchronize()
for ws in (4, 8, 16, 32, 64):
    best = None
    for rep in range(2):
        eng = acquire.Engine(0, workspace_bytes=ws << 30); eng.use_torch_stream(); eng._plan(sig, items); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.search_batch_dev(sig, xd, items, dop, B); torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.search_batch_dev(sig, xd, items, dop, B); torch.cuda.synchronize(); t2 = time.perf_counter()
        eng.close()
        cur = (1e3 * (t1 - t0), 1e3 * (t2 - t1))
        best = cur if best is None or cur[0] < best[0] else best
    print("%s %d items ws %2d GiB: first search %.1f ms, second %.1f ms" % ((name, len(items), ws) + best))
