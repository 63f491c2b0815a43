"""One rank of the CPU (gloo) sharded-search test; launched by tests/test_sharded_gloo.py.
usage: _gloo_worker.py <out.json> <name> <items csv> <doppler csv> <ms>   (RANK/WORLD_SIZE/MASTER_* in env)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_local(name, x, items, dopplers, blocks):
    """Per-rank stand-in for Engine.search_batch_dev built on the oracle: best over the local Doppler slice."""
    from gnss_dsp_tools_amd import acquire
    from oracle import acq_oracle, codes_oracle
    code, fs, n, pad, boc, norm, _fold, _b, bias = acq_oracle.VARIANTS[name]
    xs = x.numpy().astype(np.complex128)
    out = np.zeros((xs.shape[0], len(items)), dtype=acquire.PEAK_DTYPE)
    out["idx"] = -1
    out["d_index"] = -1
    for e in range(xs.shape[0]):
        for p, it in enumerate(items):
            chips = codes_oracle.chips(code, 0 if bias else it)
            for k, dop in enumerate(dopplers):
                q = acq_oracle.search_row(xs[e], chips, dop, blocks, fs=fs, n=n, pad=pad, boc=boc, bias_hz=bias * it)
                i = int(np.argmax(q))
                m = q[i] / np.mean(q) if norm else q[i]
                if m > out[e, p]["metric"]:
                    out[e, p] = (m, i, k)
    return torch.from_numpy(out.view(np.float64).reshape(xs.shape[0], len(items), 2).copy())


_HIP = {}


def hip_local(name, x, items, dopplers, blocks):
    """The real per-rank compute: this process's own HIP engine on cuda:0 (GLOO_HIP=1; several ranks then share the one GPU of the
    test box, the exchange still runs over gloo on host tensors)."""
    from gnss_dsp_tools_amd import acquire
    if "eng" not in _HIP:
        _HIP["eng"] = acquire.Engine(0)
        _HIP["eng"].use_torch_stream()
    out = _HIP["eng"].search_batch_dev(name, x.cuda(), items, dopplers, blocks)
    torch.cuda.synchronize()
    return out.cpu()


def main_mix(out_path, spec):
    """BASELINE config 5 in miniature: several different signals, every grid Doppler-sliced over the ranks, ONE all-gather
    (search_jobs_async with two steps in flight, as bench.py --config 5 runs it)."""
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    try:
        from gnss_dsp_tools_amd import acquire, sharded, signals, synth
        jobs = []
        for name, items, ds, ms in spec:
            sig = signals.get(name)
            B = sig.blocks(ms)
            xs = synth.make_epochs(sig, B, 6060, [(items[0], 0.4, 1537.0, 1201)], 1, nsamp=sig.samples_needed(B))
            jobs.append({"name": name, "x": torch.from_numpy(xs), "items": items, "dopplers": acquire.doppler_grid(ds), "blocks": B})
        sh = sharded.ShardedSearch(engine=None, local_fn=hip_local if os.environ.get("GLOO_HIP") else oracle_local)
        p1 = sh.search_jobs_async(jobs)
        p2 = sh.search_jobs_async(jobs)                     # second step queued before the first merge
        g = p1.shards()
        assert g.shape[0] == dist.get_world_size() and bool((g.view(g.shape[0], -1, 2)[:, :, 0] > 0).all())
        merged = p1.wait()
        again = p2.wait()
        res = []
        for job, m, m2 in zip(jobs, merged, again):
            assert torch.equal(m, m2)
            res.append(sh.results(job["name"], job["items"], m, job["dopplers"])[0])
        if dist.get_rank() == 0:
            with open(out_path, "w") as f:
                json.dump([[[float(v) for v in r] for r in job] for job in res], f)
    finally:
        dist.destroy_process_group()


def main():
    if os.environ.get("GLOO_MIX"):
        return main_mix(sys.argv[1], json.loads(sys.argv[2]))
    out_path, name = sys.argv[1], sys.argv[2]
    items = [int(v) for v in sys.argv[3].split(",")]
    ds = [float(v) for v in sys.argv[4].split(",")]
    ms = int(sys.argv[5])
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    try:
        from gnss_dsp_tools_amd import acquire, sharded, signals, synth
        sig = signals.get(name)
        blocks = sig.blocks(ms)
        dop = acquire.doppler_grid(ds)
        xs = synth.make_epochs(sig, blocks, 5150, [(items[0], 0.4, 1537.0, 1201)], 2)
        sh = sharded.ShardedSearch(engine=None, local_fn=hip_local if os.environ.get("GLOO_HIP") else oracle_local)
        if os.environ.get("GLOO_JOBS"):
            # two jobs, one exchange: the same search plus a second Doppler grid
            dop2 = acquire.doppler_grid([ds[0] + 37.0, ds[1], ds[2] * 2])
            jobs = [{"name": name, "x": torch.from_numpy(xs), "items": items, "dopplers": dop, "blocks": blocks},
                    {"name": name, "x": torch.from_numpy(xs), "items": items[:1], "dopplers": dop2, "blocks": blocks}]
            m1, m2 = sh.search_jobs(jobs)
            res = sh.results(name, items, m1, dop)
            res2 = sh.results(name, items[:1], m2, dop2)
            res = [a + b for a, b in zip(res, res2)]          # per epoch: job-1 results followed by job-2's
        elif os.environ.get("GLOO_ASYNC"):
            p1 = sh.search_batch_async(name, torch.from_numpy(xs), items, dop, blocks)
            p2 = sh.search_batch_async(name, torch.from_numpy(xs), items, dop, blocks)      # second search in flight before the first merge
            res = sh.results(name, items, p1.wait(), dop)
            assert sh.results(name, items, p2.wait(), dop) == res
        else:
            merged = sh.search_batch(name, torch.from_numpy(xs), items, dop, blocks)
            res = sh.results(name, items, merged, dop)
        if dist.get_rank() == 0:
            with open(out_path, "w") as f:
                json.dump([[[float(v) for v in r] for r in ep] for ep in res], f)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
