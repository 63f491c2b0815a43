#!/usr/bin/env python3
"""Benchmark of the acquisition hot path on MI355X.

Workload (BASELINE.json configs[1]): GPS L1 C/A, all 32 PRNs, 1 ms coherent, fs = 4.096 MS/s
(the reference's hard-coded rate), Doppler grid np.arange(-5000, 5000, 250) = 40 bins,
n = N = 4096 code-phase lags -> 5 242 880 cells per 1 ms epoch.  One "step" = one pass of the hot
path over a batch of EPOCHS independent 1 ms sample blocks that are already resident in HBM
(synthetic seeded IQ): table-NCO mix -> forward FFT -> x conj code spectrum ->
inverse FFT -> |.| -> peak/mean per Doppler bin -> best per PRN.

N GPUs (one process per GPU, torch.distributed/RCCL): the PRN x Doppler grid is sharded by Doppler
slice, every rank processes its slice for N*EPOCHS epochs (per-GPU work fixed -> "weak" scaling),
then ONE all-gather of the per-shard peak records and a device-side merge.

Prints ONE JSON line (rank 0).  `value` = cells/s of the whole job with inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
S = 8                           # bytes per complex64


def a_pipe_bytes(N, P, D, B, F=1):
    """Algorithmic bytes of ONE search under the stage-boundary model (SURVEY.md section 8d / BASELINE.md section 3):
    A_pipe = S*N*(4*D*B*F + P + 5*P*D*B) + 8*N*P*D*(B-1)."""
    return S * N * (4 * D * B * F + P + 5 * P * D * B) + 8 * N * P * D * (B - 1)


def stage_bytes(N, P, D, B, F=1):
    """Per-stage split of A_pipe for one search (each stage reads its input once and writes its output once)."""
    return {
        "mix_nco": S * N * 2 * D * B * F,            # read x window + write mixed block
        "rocfft_forward": S * N * 2 * D * B * F,     # read + write
        "conj_mul": S * N * (P + 2 * P * D * B),     # read C_p, read X, write Y
        "rocfft_inverse": S * N * 2 * P * D * B,     # read + write
        "mag_peak": S * N * P * D * B + 8 * N * P * D * (B - 1),
        # fused LDS engine: the correlate kernel covers conj-mul + inverse FFT + magnitude/peak,
        # the forward kernel covers mix + forward FFT
        "lds_correlate": S * N * (P + 5 * P * D * B) + 8 * N * P * D * (B - 1),
        "lds_forward": S * N * 4 * D * B * F,
    }


def cells_step_1gpu(E, P, D, N):
    return E * P * D * N


def host_info():
    """CPU model and library versions of the host the CPU baseline runs on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    import scipy
    return {"cpu_model": model, "cpu_count": os.cpu_count(), "numpy": np.__version__, "scipy": scipy.__version__}


def cpu_baseline(sig, xs, items, ds, ms, budget_s=12.0):
    """The oracle (numpy fp64 restatement of the reference, reference loop order) on this host, 1 core."""
    from oracle import acq_oracle           # checker / baseline only
    n_cells_epoch = len(items) * len(np.arange(*ds)) * sig.nfft
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:          # bounded sample: ~budget_s seconds of CPU work
        x = xs[done % xs.shape[0]].astype(np.complex128)
        for it in items:
            acq_oracle.search_script(sig.name, x, it, ds, ms)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done * n_cells_epoch / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%d epoch(s) x %d PRNs x %d Doppler bins x %d lags, numpy/scipy fp64 oracle in the reference's loop order "
                      "(per-PRN forward FFT), %.1f s on %d-core host" % (done, len(items), len(np.arange(*ds)), sig.nfft, dt, os.cpu_count())}


def _pool_worker(task):
    name, x, it, ds, ms = task
    from oracle import acq_oracle
    return acq_oracle.search_script(name, x, it, ds, ms)


def cpu_baseline_pool(sig, xs, items, ds, ms, reps=60):
    """Same oracle through multiprocessing.Pool(cpu_count()) with one task per PRN and x pickled per task -- the
    reference's own parallel harness (acquire-gps-l1.py:98-108)."""
    import multiprocessing as mp
    cores = os.cpu_count()
    n_cells_epoch = len(items) * len(np.arange(*ds)) * sig.nfft
    ctx = mp.get_context("fork")
    with ctx.Pool(min(cores, len(items))) as pool:
        x = xs[0].astype(np.complex128)
        pool.map(_pool_worker, [(sig.name, x, it, ds, ms) for it in items])          # warm-up: code caches, page faults
        t0 = time.perf_counter()
        for r in range(reps):
            x = xs[r % xs.shape[0]].astype(np.complex128)
            pool.map(_pool_worker, [(sig.name, x, it, ds, ms) for it in items])
        dt = time.perf_counter() - t0
    return {"value": reps * n_cells_epoch / dt, "unit": "cells/s", "cores": min(cores, len(items)), "kind": "port",
            "sample": "%d epoch(s) via multiprocessing.Pool(%d).map over %d PRNs (one task per PRN, x pickled per task, like "
                      "acquire-gps-l1.py:105-108), %.2f s" % (reps, min(cores, len(items)), len(items), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=64, help="1 ms epochs per GPU per step")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 rocFFT pipeline, 2 LDS FFT kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=1,
                    help="engine contexts / HIP streams the independent steps alternate between (2 fills the correlate kernel's tail, "
                         "+3.5 %%, but overlapping launches make per-kernel durations meaningless for the roofline line)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-epoch host-call latency probe (profiling runs: keeps every launch the bench workload)")
    ap.add_argument("--force-gather", action="store_true", help="run the all-gather + merge even on 1 rank (test aid)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gnss_dsp_tools_amd import acquire, sharded, signals, synth

    sig = signals.get("gps-l1")
    items = list(range(1, 33))
    ds = [-5000.0, 5000.0, 250.0]
    ms = 1
    B = sig.blocks(ms)
    dop = acquire.doppler_grid(ds)
    P, D, N = len(items), len(dop), sig.nfft
    E_total = args.epochs * world                       # weak scaling: per-GPU work is fixed
    sats = synth.default_sats(items)
    # a few distinct seeded epochs tiled to the batch (content does not change the work)
    base = synth.make_epochs(sig, B, synth.BASE_SEED + 2, sats, min(8, E_total), nsamp=B * sig.n)
    xs = np.concatenate([base] * ((E_total + len(base) - 1) // len(base)))[:E_total]
    x_dev = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)

    eng = acquire.Engine(local_rank, engine=args.engine)
    eng.use_torch_stream(dev)                           # same stream as the RCCL collective -> ordered
    sh = sharded.ShardedSearch(engine=eng, always_gather=args.force_gather)

    def step():
        return sh.search_batch(sig, x_dev, items, dop, B)

    # Steps are independent searches (a receiver scanning a recording keeps several batches in flight).  With N > 1 the
    # all-gather of step i runs under the kernels of step i+1 (asynchronous collective, merge deferred by one step); with
    # --lanes 2 the steps also alternate between two engine contexts with their own HIP streams and workspaces, so the next
    # step's kernels fill the CUs the tail of the correlate kernel leaves idle.  Every step runs all of its kernels, the
    # exchange and the merge inside the timed region.
    lanes = []
    for _ in range(max(1, args.lanes)):
        st = torch.cuda.Stream(dev)
        e2 = acquire.Engine(local_rank, engine=args.engine)
        with torch.cuda.stream(st):
            lanes.append((st, e2, sharded.ShardedSearch(engine=e2, always_gather=args.force_gather)))

    def run_steps(k):
        pend = [None] * len(lanes)
        out = None
        for i in range(k):
            st, _, shl = lanes[i % len(lanes)]
            with torch.cuda.stream(st):
                nxt = shl.search_batch_async(sig, x_dev, items, dop, B)      # next search queued before the previous merge:
                if pend[i % len(lanes)] is not None:                          # its kernels cover the previous exchange
                    out = pend[i % len(lanes)].wait()
                pend[i % len(lanes)] = nxt
        for (st, _, _), p in zip(lanes, pend):
            if p is not None:
                with torch.cuda.stream(st):
                    out = p.wait()
        return out

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    merged = run_steps(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    merged = run_steps(args.steps)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness spot check of what was just computed (not timed) ----------------------------
    res = sh.results(sig, items, merged[:1], dop)[0]
    detected = {it: r for it, r in zip(items, res)}
    for it, amp, f, delay in sats:
        if amp >= 0.25 and not os.environ.get("GACQ_LIB"):          # profiling-only ablation builds compute garbage on purpose
            want_code = 1023 * (((-delay) % sig.n) / sig.n)
            assert abs(detected[it][1] - want_code) < 1e-9, ("bench self-check failed", it, detected[it], want_code)

    # ---- roofline of the dominant kernel: HIP events on the launch stream, separate profiled pass ---
    cells_step = E_total * P * D * N
    bounds = sharded.doppler_bounds(D, world)
    D_local = bounds[rank + 1] - bounds[rank]
    eng.set_profiling(True)
    eng.reset_stage_times()
    prof_steps = max(3, min(10, args.steps))
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize(dev)
    stages = eng.stage_times()
    eng.set_profiling(False)
    sb = stage_bytes(N, P, D_local, B)
    per_stage = {}
    for name, (tot_ms, n) in stages.items():
        if n:
            key = "lds_forward" if (name == "mix_nco" and stages["lds_correlate"][1]) else name
            per_stage[name] = {"avg_ms": tot_ms / n, "launches_per_step": n / prof_steps,
                               "alg_bytes_per_launch": sb.get(key, 0) * E_total * (prof_steps / n) if key in sb else None}
    dominant = max((k for k in per_stage if per_stage[k]["alg_bytes_per_launch"]), key=lambda k: per_stage[k]["avg_ms"] * per_stage[k]["launches_per_step"])
    dk = per_stage[dominant]
    achieved = dk["alg_bytes_per_launch"] / (dk["avg_ms"] * 1e-3) / 1e9
    traffic = None
    pipe_busy = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel_stage") == dominant and tj.get("epochs") == E_total and world == 1:
                traffic = tj.get("hbm_bytes_per_launch")
                pipe_busy = tj.get("valu_pipe_busy")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "avg_kernel_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes_per_launch"],
                "model": "stage-boundary A_pipe share of this kernel (SURVEY.md 8d); fused kernels keep stage boundaries in LDS, "
                         "so achieved can exceed what HBM alone could deliver -- see traffic for measured HBM bytes"}

    # the fused LDS kernel is VALU-bound: useful FP32 work vs the 157.3 TFLOP/s vector peak (MI355X_MICROARCH.md)
    valu = None
    if dominant == "lds_correlate":
        rows_launch = E_total * P * D_local * B / dk["launches_per_step"]
        flops = rows_launch * (5.0 * N * np.log2(N) + 6.0 * N + 4.0 * N)     # inverse FFT + C*X + |.|
        valu = {"useful_flop_per_launch": flops, "achieved_TFLOPs": flops / (dk["avg_ms"] * 1e-3) / 1e12, "peak_TFLOPs": 157.3,
                "frac": flops / (dk["avg_ms"] * 1e-3) / 1e12 / 157.3,
                # fraction of all SIMD cycles spent issuing VALU work, from the SQ counters of the same command (profiles/)
                "pipe_busy_pmc": pipe_busy}

    # host-buffer entry point (gacq_search: H2D + launches + D2H + sync), the drop-in search() call surface; not part of `value`
    latency = None
    if world == 1 and not args.no_latency:
        eng.set_stream(None)
        xh = base[0]
        def median_call(fn, n=60):
            ts = []
            for _ in range(n):
                t1 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t1)
            return float(np.median(ts))
        for _ in range(3):
            eng.search_all(sig, xh, items, ds, ms)
        t_all = median_call(lambda: eng.search_all(sig, xh, items, ds, ms))             # median of 60 calls
        for _ in range(3):
            eng.search(sig, xh, 7, ds, ms)              # builds the 1-PRN signal (code spectrum, FFT plan) once
        t_one = median_call(lambda: eng.search(sig, xh, 7, ds, ms))
        latency = {"search_all_32prn_us": t_all * 1e6, "search_1prn_us": t_one * 1e6,
                   "cells_per_s_single_epoch_pcie_inclusive": P * D * N / t_all}
        eng.use_torch_stream(dev)
        # host-resident batches streamed through pinned double buffers (H2D of batch i+1 under the kernels of batch i)
        from gnss_dsp_tools_amd import stream
        st = stream.EpochStreamer(eng, sig, items, dop, B, E_total, xs.shape[1], depth=3, device=dev)
        nb = 24
        for _ in st.run(xs for _ in range(3)):
            pass
        t1 = time.perf_counter()
        last = None
        for last in st.run(xs for _ in range(nb)):
            pass
        t_stream = (time.perf_counter() - t1) / nb
        latency["streamed_batches_pcie_inclusive"] = {"cells_per_s": cells_step_1gpu(E_total, P, D, N) / t_stream, "ms_per_batch": t_stream * 1e3,
                                                      "epochs_per_batch": E_total, "h2d_bytes_per_batch": int(xs.nbytes)}

    out = None
    if rank == 0:
        a_pipe_step = a_pipe_bytes(N, P, D, B) * E_total
        out = {
            "metric": "acquisition cells/s (PRN x Doppler x code-phase), GPS L1 C/A 1 ms",
            "value": cells_step * args.steps / dt,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "GPS L1 C/A all 32 PRNs, 1 ms coherent (B=1), fs=4.096 MS/s, n=N=4096, "
                                   "Doppler arange(-5000,5000,250)=40 bins; %d epochs/step/GPU batched, inputs resident in HBM" % args.epochs,
                       "prns": P, "doppler_bins": D, "lags": N, "blocks": B, "epochs_per_step": E_total,
                       "cells_per_step": cells_step, "sharding": "doppler-slice x%d + 1 all-gather of peaks per step (async, overlapped with the next step's kernels)" % world,
                       "engine": {0: "auto", 1: "rocfft", 2: "lds-fft"}[args.engine],
                       "steps_in_flight": len(lanes)},
            "roofline": roofline,
            "valu": valu,
            "host_call_latency": latency,
            "pipeline": {"a_pipe_bytes_per_step": a_pipe_step,
                         # compulsory I/O only: samples in, code spectra, 16-byte peak records out (SURVEY.md 8d "A_min")
                         "a_min_bytes_per_step": E_total * B * N * 8 + P * N * 8 + E_total * P * 16, "achieved_GBps": a_pipe_step / (dt / args.steps) / 1e9,
                         "frac_of_8TBps": a_pipe_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS,
                         "us_per_search": dt / args.steps / E_total * 1e6, "stages": per_stage},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sig, base, items, ds, ms)
            out["cpu_baseline"]["host"] = host_info()
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            try:
                out["cpu_baseline_pool"] = cpu_baseline_pool(sig, base, items, ds, ms)
            except Exception as exc:                       # the pool leg is informative only
                out["cpu_baseline_pool"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for _, e2, _ in lanes:
        e2.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
