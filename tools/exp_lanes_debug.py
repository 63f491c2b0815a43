import gc, json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
import gnss_dsp_tools_amd as g
from gnss_dsp_tools_amd import acquire, sharded
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
jobs = bench.build_jobs(bench.CONFIGS[5], 1, dev)
extra = "--extra-engine" in sys.argv
if extra:
    eng = acquire.Engine(0); eng.use_torch_stream(dev)
lanes = []
for _ in range(2):
    st = torch.cuda.Stream(dev)
    e2 = acquire.Engine(0)
    with torch.cuda.stream(st):
        lanes.append((st, e2, sharded.ShardedSearch(engine=e2)))
L = [((lambda st=st: torch.cuda.stream(st)), shl) for st, _, shl in lanes]
def timed(lanes_, k, nogc):
    run = bench.make_run_steps(lanes_, jobs)
    if nogc:
        gc.collect(); gc.disable()
    run(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(k); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    gc.enable()
    return (t2 - t0) / k * 1e3, (t1 - t0) / k * 1e3
for rep in range(3):
    for nl in (1, 2):
        for nogc in (False, True):
            ms, host = timed(L[:nl], 100, nogc)
            print(json.dumps({"lanes": nl, "gc_disabled": nogc, "extra_engine": extra, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(host, 3)}))
