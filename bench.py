#!/usr/bin/env python3
"""Benchmark of the acquisition hot path on MI355X.

Default workload (BASELINE.json configs[1], the one the metric is quoted on): GPS L1 C/A, all 32 PRNs, 1 ms coherent,
fs = 4.096 MS/s (the reference's hard-coded rate), Doppler grid np.arange(-5000, 5000, 250) = 40 bins,
n = N = 4096 code-phase lags -> 5 242 880 cells per 1 ms epoch.  One "step" = one pass of the hot path over a batch of
EPOCHS independent sample blocks that are already resident in HBM (synthetic seeded IQ):
table-NCO mix -> forward FFT -> x conj code spectrum -> inverse FFT -> |.| -> peak/mean per Doppler bin -> best per PRN.

--config 3 runs BASELINE configs[2] (E1B + E1C as one 72-row family with shared forward transforms); --config 4 / 5 run the
multi-signal shapes of configs[3] / configs[4] (L5I + B2aD; GPS L1 + E1B + B1I + GLONASS) through ShardedSearch.search_jobs: every signal's Doppler grid is sliced over the ranks, ONE all-gather carries all
signals' peak records.

N GPUs (one process per GPU, torch.distributed/RCCL): --scaling weak (default) keeps per-GPU work fixed (N x EPOCHS epochs,
D/N bins per rank); --scaling strong keeps the total work fixed (EPOCHS epochs).  Steps are independent searches; the
all-gather of step i is issued asynchronously and merged after step i+1 has been queued.

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
VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector peak (one FMA per lane per cycle)
S = 8                           # bytes per complex64

# name, items, doppler_search, ms (int) or ("B", blocks) for an engine-level block count
CONFIGS = {
    2: {"label": "GPS L1 C/A all 32 PRNs, 1 ms coherent (B=1), fs=4.096 MS/s, n=N=4096, Doppler arange(-5000,5000,250)=40 bins",
        "epochs": 1024, "seed": 2,
        "jobs": [("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1)]},
    3: {"label": "Galileo E1B + E1C as one family (forward transforms shared), PRN 1-36 each, BOC(1,1), 4092-chip memory codes, fs=8.192 MS/s, "
                 "n=32768, N=65536 (padded), ms=8 (B=1), Doppler arange(-4000,4000,125)=64 bins",
        "epochs": 2, "seed": 3,
        "jobs": [(("galileo-e1b", "galileo-e1c"), (list(range(1, 37)), list(range(1, 37))), [-4000.0, 4000.0, 125.0], 8)]},
    4: {"label": "GPS L5I PRN 1-32 + BeiDou B2aD PRN 1-63, 10.23 Mcps, fs=30.69 MS/s, n=30690, N=61380 (padded), B=1, "
                 "Doppler arange(-7000,7000,200)=70 bins",
        "epochs": 2, "seed": 4,
        "jobs": [("gps-l5i", list(range(1, 33)), [-7000.0, 7000.0, 200.0], ("B", 1)),
                 ("beidou-b2ad", list(range(1, 64)), [-7000.0, 7000.0, 200.0], ("B", 1))]},
    5: {"label": "cold start: GPS L1 (32 PRNs, B=10, N=4096) + Galileo E1B (50 PRNs, B=1, N=65536) + BeiDou B1I (63 PRNs, B=10, "
                 "N=16384) + GLONASS L1 (15 channels, B=10, N=16384), Doppler arange(-10000,10000,100)=200 bins, ms=10",
        "epochs": 1, "seed": 5,
        "jobs": [("gps-l1", list(range(1, 33)), [-10000.0, 10000.0, 100.0], 10),
                 ("galileo-e1b", list(range(1, 51)), [-10000.0, 10000.0, 100.0], 10),
                 ("beidou-b1i", list(range(1, 64)), [-10000.0, 10000.0, 100.0], 10),
                 ("glonass-l1", list(range(-7, 8)), [-10000.0, 10000.0, 100.0], 10)]},
}


class ClockSampler:
    """Shader clock and socket power of the benchmark's GPU while a timed region runs: a thread reads the amdgpu sysfs files (the
    starred level of pp_dpm_sclk = what rocm-smi prints as sclk, hwmon power1_input in microwatts) every 25 ms; no subprocess, no
    runtime call.  The chip clocks to its power budget (MI355X_MICROARCH.md, DVFS), so a roofline priced at 2.4 GHz under-reads a kernel
    that holds the part at its power limit: frac_at_measured_clock divides by the clock that was actually there."""

    def __init__(self, dev_index):
        import glob
        self.sclk_path = self.power_path = None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cards = []
        for c in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                cards.append((os.path.basename(os.path.realpath(c)), c))
        pick = [c for a, c in cards if want and a.lower() == want.lower()] or ([cards[0][1]] if len(cards) == 1 else [])
        if pick:
            self.sclk_path = os.path.join(pick[0], "pp_dpm_sclk")
            pw = glob.glob(os.path.join(pick[0], "hwmon", "hwmon*", "power1_input")) + glob.glob(os.path.join(pick[0], "hwmon", "hwmon*", "power1_average"))
            self.power_path = pw[0] if pw else None
        self.samples, self._stop, self._thread = [], False, None

    def _read(self):
        mhz = watts = None
        try:
            for line in open(self.sclk_path).read().splitlines():
                if line.rstrip().endswith("*"):
                    mhz = float(line.split(":")[1].strip().split("M")[0])
            if self.power_path:
                watts = float(open(self.power_path).read()) * 1e-6
        except Exception:
            pass
        return mhz, watts

    def start(self):
        if not self.sclk_path:
            return self
        import threading

        def loop():
            while not self._stop:
                self.samples.append(self._read())
                time.sleep(0.025)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop = True
        if self._thread:
            self._thread.join()
        mhz = [m for m, _ in self.samples if m]
        w = [x for _, x in self.samples if x]
        if not mhz:
            return None
        return {"sclk_mhz_mean": float(np.mean(mhz)), "sclk_mhz_min": float(np.min(mhz)), "sclk_mhz_max": float(np.max(mhz)),
                "power_w_mean": float(np.mean(w)) if w else None, "samples": len(mhz),
                "source": "amdgpu sysfs (pp_dpm_sclk starred level, hwmon power1_input), one read per 25 ms during the sustained timed region"}


def a_pipe_bytes(N, P, D, B, F=1):
    """Algorithmic bytes of ONE search under the stage-boundary model (SURVEY.md section 8d / BASELINE.md section 3):
    A_pipe = S*N*(4*D*B*F + P + 5*P*D*B) + 8*N*P*D*(B-1)."""
    return S * N * (4 * D * B * F + P + 5 * P * D * B) + 8 * N * P * D * (B - 1)


def a_min_bytes(N, P, B, nsamp):
    """Compulsory I/O of one search: samples in, code spectra, 16-byte peak records out (SURVEY.md 8d "A_min")."""
    return nsamp * S + P * N * S + P * 16


def engine_kind(N):
    """Which hand-written engine serves this FFT length."""
    if N in (4096, 16384):
        return "lds"                 # whole transform in one workgroup's registers + LDS
    if N % 31 == 0:
        return "split31"             # prime-factor form 31 x 11 x Nb x 9 (gacq_pfa.hip), one Z' round trip through HBM
    return "split_lds"               # outer DFT-R + 4096-point LDS inner transforms, one Z' round trip through HBM


def stage_model(kind, stage, N, P, D, B, F, E, fused16k, fused4k=False):
    """What bounds a launch of `stage` and its algorithmic work for E epochs of one job:
    ("valu", useful FP32 flop) for the LDS-resident transform kernels, ("hbm", bytes) for the kernels on either side of the
    split engines' one unavoidable round trip (inner IFFT rows Z' written once, read once: 8*N bytes each way per
    correlation row).  None: latency-bound helper kernels."""
    rows = E * P * D * B                     # correlation rows
    frows = E * F * D * B                    # forward rows
    fft = 5.0 * N * np.log2(N)
    if kind == "lds":
        if stage == "lds_correlate":
            per_row = fft + 6.0 * N + 4.0 * N                 # inverse FFT + C*X + |.|
            if fused16k:
                per_row += fft + 6.0 * N                      # + mix + forward FFT in the same kernel
            # fused 4096 kernel: ONE forward transform per (epoch, Doppler bin) is useful work; the copies the other item
            # chunks of the same unit recompute are overhead and do not count
            return "valu", rows * per_row + (frows * (fft + 6.0 * N) if fused4k else 0.0)
        if stage == "mix_nco":
            return "valu", frows * (fft + 6.0 * N)
    else:
        if stage == "lds_correlate":                          # K2 + inner inverse transforms: writes Z'
            return "hbm", rows * S * N
        if stage == "mag_peak":                               # outer inverse DFT + |.| + reduce: reads Z'
            return "hbm", rows * S * N
        if stage == "mix_nco":                                # outer forward DFT (+ inner forward): x in, X out
            return "hbm", frows * 2 * S * N
    return None, None


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


def cpu_baseline(jobs, budget_s=14.0):
    """The oracle (numpy fp64 restatement of the reference, reference loop order: one search() per item with its own
    forward FFTs) on this host, 1 core, on a bounded sample: items of every job in turn until the budget is spent.  The
    whole-workload rate combines the per-signal rates by their share of the cells (a harmonic mean by work)."""
    from oracle import acq_oracle           # checker / baseline only
    t_job = [0.0] * len(jobs)
    c_job = [0] * len(jobs)
    n_job = [0] * len(jobs)
    t0 = time.perf_counter()
    k = 0
    while True:
        for ji, job in enumerate(jobs):
            sig = job["sig"]
            oname, it = job["cpu_items"][k % len(job["cpu_items"])]
            x = job["host"][k % job["host"].shape[0]].astype(np.complex128)
            t1 = time.perf_counter()
            acq_oracle.search_script_blocks(oname, x, it, job["ds"], job["B"])
            t_job[ji] += time.perf_counter() - t1
            c_job[ji] += len(job["dop"]) * sig.nfft
            n_job[ji] += 1
        k += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cells_epoch = [j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs]
    t_epoch = sum(c / (cj / tj) for c, cj, tj in zip(cells_epoch, c_job, t_job))
    return {"value": sum(cells_epoch) / t_epoch, "unit": "cells/s", "cores": 1, "kind": "port",
            "per_signal_cells_per_s": {j["label"]: cj / tj for j, cj, tj in zip(jobs, c_job, t_job)},
            "sample": "%s, numpy/scipy fp64 oracle in the reference's loop order (per-item forward FFTs), %.1f s on 1 of %d "
                      "cores; whole-workload rate = cells per epoch / sum over signals of (cells / measured rate)"
                      % (", ".join("%d item search(es) of %s (%d Doppler bins x %d lags x B=%d)" % (n, j["label"], len(j["dop"]), j["sig"].nfft, j["B"])
                                   for j, n in zip(jobs, n_job)), dt, os.cpu_count())}


def _pool_worker(task):
    name, x, it, ds, B = task
    from oracle import acq_oracle
    return acq_oracle.search_script_blocks(name, x, it, ds, B)


def cpu_baseline_pool(job, runs=5, reps=12):
    """Same oracle through multiprocessing.Pool(cpu_count()) with one task per PRN and x pickled per task -- the
    reference's own parallel harness (acquire-gps-l1.py:98-108).  Config 2 only (one signal).  BASELINE.md section 4: median of
    >= 5 timed runs after a warm-up; every run is `reps` epochs."""
    import multiprocessing as mp
    sig, items, dop, B, ds = job["sig"], job["items"], job["dop"], job["B"], job["ds"]
    cores = os.cpu_count()
    n_cells_epoch = len(items) * len(dop) * sig.nfft
    ctx = mp.get_context("fork")
    rates = []
    with ctx.Pool(min(cores, len(items))) as pool:
        x = job["host"][0].astype(np.complex128)
        for _ in range(2):
            pool.map(_pool_worker, [(sig.name, x, it, ds, B) for it in items])      # warm-up: code caches, page faults
        t_all = time.perf_counter()
        for _ in range(runs):
            t0 = time.perf_counter()
            for r in range(reps):
                x = job["host"][r % job["host"].shape[0]].astype(np.complex128)
                pool.map(_pool_worker, [(sig.name, x, it, ds, B) for it in items])
            rates.append(reps * n_cells_epoch / (time.perf_counter() - t0))
        dt = time.perf_counter() - t_all
    return {"value": float(np.median(rates)), "unit": "cells/s", "cores": min(cores, len(items)), "kind": "port", "runs": rates,
            "sample": "median of %d runs of %d epoch(s) each via multiprocessing.Pool(%d).map over %d PRNs (one task per PRN, x pickled per "
                      "task, like acquire-gps-l1.py:105-108) after 2 warm-up epochs, %.2f s in all" % (runs, reps, min(cores, len(items)), len(items), dt)}


def pmc_passes(argv, kernel, passes, timeout_s=240):
    """Hardware counters of `kernel` for this very command line, collected live: one `rocprofv3 --pmc` child run of bench.py per
    counter group (counter passes are never combined with tracing; FETCH_SIZE and WRITE_SIZE need separate passes --
    MI355X_MICROARCH.md, HBM / rocprofv3 section).  Returns {counter: average per launch of the kernels whose name contains
    `kernel`} plus the launch count; raises on any failure so that the caller can fall back and say so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    out = {}
    tmp = tempfile.mkdtemp(prefix="gacq_pmc_", dir="/tmp")
    try:
        for counters in passes:
            d = os.path.join(tmp, counters[0])
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + argv
            env = dict(os.environ, TMPDIR="/tmp", GACQ_BENCH_PMC_CHILD="1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 --pmc %s: rc %d: %s" % (counters[0], r.returncode, (r.stderr or r.stdout)[-300:]))
            per = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if kernel in row["Kernel_Name"]:
                        key = (row["Counter_Name"], row["Dispatch_Id"])
                        per[key] = per.get(key, 0.0) + float(row["Counter_Value"] or 0)
            for c in counters:
                vals = [v for (cn, _), v in per.items() if cn == c]
                if not vals:
                    raise RuntimeError("no %s rows for kernel %s" % (c, kernel))
                out[c] = sum(vals) / len(vals)
                out["launches"] = len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def condense(o, wall_s):
    """The short form of a side run's JSON line that the headline line carries under other_configs."""
    r, sus = o["roofline"], o.get("sustained")
    d = {"baseline_config": o["config"]["baseline_config"], "workload": o["config"]["workload"], "signals": o["config"]["signals"],
         "value": sus["value"] if sus else o["value"], "unit": "cells/s",
         "ms_per_step": sus["ms_per_step"] if sus else o["ms_per_step"], "steps_timed": sus["steps"] if sus else o["steps"],
         "seconds_timed": sus["seconds"] if sus else o["ms_per_step"] * o["steps"] * 1e-3,
         "cells_per_step": o["config"]["cells_per_step"], "cell_blocks_per_step": o["config"]["cell_blocks_per_step"],
         "dominant_kernel": r["kernel"], "dominant_signal": r["signal"], "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"],
         "roofline_unit": r["unit"], "frac": r["frac"], "avg_kernel_ms": r["avg_kernel_ms"], "stream_ceiling": r.get("stream_ceiling"),
         "algorithmic_per_launch": r.get("alg_bytes_per_launch", r.get("useful_flop_per_launch")),
         "compulsory_hbm_bytes_per_launch": r.get("compulsory_bytes_per_launch"),
         "traffic_measured_bytes_per_launch": r.get("traffic"),
         "traffic_measured_in_this_run": bool((r.get("traffic_source") or {}).get("measured_in_this_run")),
         "ms_per_step_by_signal": {pj["signal"]: sum(st["ms_per_step"] for st in pj["stages"].values()) for pj in o["pipeline"]["per_signal"]},
         "steps_in_flight": o["config"].get("steps_in_flight"), "one_step_in_flight": o["config"].get("one_step_in_flight"),
         "wall_s": wall_s}
    if d["traffic_measured_bytes_per_launch"] and d["compulsory_hbm_bytes_per_launch"]:
        d["traffic_over_compulsory"] = d["traffic_measured_bytes_per_launch"] / d["compulsory_hbm_bytes_per_launch"]
    return d


FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X data sheet, FP64 vector (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz); SURVEY App. A plans with 79


def reference_precision(args, env, f32_value, epochs=1024, seconds=1.0):
    """The headline workload (config 2, the same 1024-epoch step) in the reference's own arithmetic type: engine 5 keeps every value
    complex128 / fp64 on the device.  For N = 4096, B = 1 that is ONE kernel per search (fused4k_c128_kernel: table-NCO mix, forward
    transform, C_p * conj(.), inverse transform, |.|, row reduction, all in fp64 in one workgroup's registers + LDS).  Timed for >=
    `seconds` with its own roofline (useful FP64 flop over the kernel's HIP-event time against the FP64 vector peak); the fp32 engine is
    timed in the SAME loop (same epochs, same synchronisation cadence) for a like-for-like ratio, and the two engines' peak records are
    compared on the same epochs (locations must be identical -- tie-safe locations -- and metrics within 1e-5)."""
    from gnss_dsp_tools_amd import acquire
    dev = env["dev"]
    job = build_jobs(CONFIGS[2], epochs, dev)[0]
    sig, items, dop, B = job["sig"], job["items"], job["dop"], job["B"]
    cells = epochs * len(items) * len(dop) * sig.nfft
    rate, peaks, res = {}, {}, {}
    for label, which in (("f64", 5), ("f32", 0)):
        eng = acquire.Engine(env["local_rank"], engine=which)
        eng.use_torch_stream(dev)
        try:
            for _ in range(2):
                pk = eng.search_batch_dev(sig, job["x"], items, dop, B)
            torch.cuda.synchronize(dev)
            peaks[label] = pk.cpu().numpy().view(acquire.PEAK_DTYPE).reshape(epochs, len(items))
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for _ in range(2):
                    eng.search_batch_dev(sig, job["x"], items, dop, B)
                torch.cuda.synchronize(dev)
                n += 2
            dt = time.perf_counter() - t0
            rate[label] = {"value": n * cells / dt, "ms_per_step": dt / n * 1e3, "steps_timed": n, "seconds_timed": dt}
            if which == 5:
                # the kernel's own duration from HIP events on the launch stream (stage 6 = the fused row kernel)
                eng.set_profiling(True)
                eng.reset_stage_times()
                for _ in range(4):
                    eng.search_batch_dev(sig, job["x"], items, dop, B)
                torch.cuda.synchronize(dev)
                st = eng.stage_times()
                eng.set_profiling(False)
                k_ms, k_n = st["lds_correlate"]
                rows = epochs * len(dop) * (len(items) + 1)                      # inverse rows + ONE forward row per (epoch, Doppler bin)
                Nf = sig.nfft
                flop = epochs * (len(items) * len(dop) * (5 * Nf * 12 + 10 * Nf) + len(dop) * (5 * Nf * 12 + 6 * Nf))
                if k_n:
                    avg_ms = k_ms / k_n
                    ach = flop / (avg_ms * 1e-3) / 1e12
                    res["roofline"] = {"kernel": "fused4k_c128_kernel", "bound": "valu", "unit": "TFLOP/s", "achieved": ach,
                                       "peak": FP64_VECTOR_PEAK_TFLOPS, "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "avg_kernel_ms": avg_ms,
                                       "useful_flop_per_launch": flop, "rows_per_launch": rows,
                                       "note": "useful FP64 flop of the rows one launch transforms (5 N log2 N per transform, 6 N per complex "
                                               "product, 4 N for |.|; one forward transform per (epoch, Doppler bin) counts) over the kernel's "
                                               "HIP-event duration, against the FP64 vector peak"}
        finally:
            eng.close()
    res.update({"engine": "5: complex128 on the device (fp64 table-NCO mix, transforms, conj-multiply, magnitudes, metric and Doppler scan), the "
                          "reference's arithmetic type; N = 4096, B = 1: one fused kernel per search (fused4k_c128_kernel)", "dtype": "f64",
                "workload": "BASELINE config 2, %d epochs/step resident in HBM" % epochs, "unit": "cells/s"})
    res.update(rate["f64"])
    res["f32_same_loop"] = rate["f32"]
    res["f32_over_f64"] = rate["f32"]["value"] / rate["f64"]["value"]
    a, b = peaks["f32"], peaks["f64"]
    res["f32_vs_f64_on_the_same_%d_epochs" % epochs] = {
        "searches": int(a.size), "peak_location_mismatches": int(((a["idx"] != b["idx"]) | (a["d_index"] != b["d_index"])).sum()),
        "max_rel_metric_error": float(np.max(np.abs(a["metric"] - b["metric"]) / np.abs(b["metric"])))}
    return res


def reference_precision_other_configs(args, env, seconds=0.6):
    """BASELINE configs 3, 4 and 5 in the reference's arithmetic type: engine 5 (complex128 on the device) on each config's own step,
    >= `seconds` timed per config, the fp32 engines timed in the same loop, and the two engines' peak records compared on the same
    epochs (identical locations, metrics within 1e-5).  N = 16384 / 65536 (configs 3 and 5) run the hand-written split form (round 6:
    forward spectra shared by the items, ONE Z' round trip of 32 N bytes per row and block on the LDS-resident complex128 transform),
    N = 61380 (config 4) the Cooley-Tukey form 31 x 1980 with hand-written fp64 DFT-31 stages around rocFFT's native length-1980
    transforms (no Bluestein); the five-stage rocFFT double-precision pipeline they replaced is timed in the same loop (option
    fused_c128 = 0).  GPS L1 with B > 1 still runs that pipeline.  Every record carries its own roofline block: HIP-event time of the Z' writer +
    reader (or of the pipeline's five stages) against the bytes they must move."""
    from gnss_dsp_tools_amd import acquire
    dev = env["dev"]
    out = []
    for k in (3, 4, 5):
        cfg = CONFIGS[k]
        E = cfg["epochs"]
        jobs = build_jobs(cfg, E, dev)
        cells = sum(E * j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
        rec = {"baseline_config": k, "workload": cfg["label"], "epochs_per_step": E, "unit": "cells/s", "dtype": "f64"}
        peaks = {}
        split_jobs = [j for j in jobs if j["sig"].nfft in (16384, 65536, 61380, 30690)]          # lengths with a hand-written complex128 form
        try:
            for label, which, opt in (("f64", 5, 1), ("f64_pipeline", 5, 0), ("f32", 0, 1)):
                if label == "f64_pipeline" and not split_jobs:
                    continue
                eng = acquire.Engine(env["local_rank"], engine=which)
                eng.use_torch_stream(dev)
                eng.set_option("fused_c128", opt)
                try:
                    def step():
                        res = []
                        for j in jobs:
                            if j["family"]:
                                res.append(eng.search_family_batch_dev(j["family"], j["x"], j["items"], j["dop"], j["B"]))
                            else:
                                res.append(eng.search_batch_dev(j["sig"], j["x"], j["items"], j["dop"], j["B"]))
                        return res
                    res = step()
                    torch.cuda.synchronize(dev)
                    peaks[label] = np.concatenate([r.cpu().numpy().view(acquire.PEAK_DTYPE).reshape(-1) for r in res])
                    n, t0 = 0, time.perf_counter()
                    while time.perf_counter() - t0 < seconds or n < 2:
                        step()
                        torch.cuda.synchronize(dev)
                        n += 1
                    dt = time.perf_counter() - t0
                    rec[label if label != "f64" else "f64_timing"] = {"value": n * cells / dt, "ms_per_step": dt / n * 1e3, "steps_timed": n, "seconds_timed": dt}
                    if label == "f64":
                        # HIP events around the stages (one context for all signals of the step: the stage sums are the step's)
                        eng.set_profiling(True)
                        eng.reset_stage_times()
                        for _ in range(3):
                            step()
                        torch.cuda.synchronize(dev)
                        stg = {kk: v[0] / 3.0 for kk, v in eng.stage_times().items() if v[1]}
                        eng.set_profiling(False)
                        # R x 4096: Z' written once and read once (2 x 16 N per row and block); 31 x M: conj-multiply out, rocFFT inverse in
                        # and out, reader in (4 x 16 N)
                        z_bytes = sum((2.0 if j["sig"].nfft in (16384, 65536) else 4.0) * 16 * j["sig"].nfft * E * j["P"] * len(j["dop"]) * j["B"] for j in split_jobs)
                        if split_jobs:
                            ms_z = stg.get("lds_correlate", 0.0) + stg.get("mag_peak", 0.0) + stg.get("conj_mul", 0.0) + stg.get("rocfft_inverse", 0.0)
                            r31 = any(j["sig"].nfft in (61380, 30690) for j in split_jobs)
                            rec["roofline"] = {"bound": "hbm", "kernels": ("conj_mul64_kernel + rocFFT double inverse (length N / 31) + c128_r31_reader_kernel" if r31 else
                                                                           "c128_split_corr_kernel (Z' writer) + c128_split_reader_kernel"), "unit": "GB/s",
                                               "signals": [j["label"] for j in split_jobs], "alg_bytes_per_step": z_bytes, "kernels_ms_per_step": ms_z,
                                               "achieved": z_bytes / (ms_z * 1e-3) / 1e9 if ms_z else None, "peak": HBM_PEAK_GBPS,
                                               "frac": (z_bytes / (ms_z * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ms_z else None, "stage_ms_per_step": stg,
                                               "model": "the correlation-side stage boundaries of the hand-written form -- R x 4096: Z' written and read once, 2 x 16 N "
                                                        "bytes per correlation row and block; 31 x M: conj-multiply out, inner inverse transform in and out, "
                                                        "reader in, 4 x 16 N -- over the HIP-event time of those launches of a step (all signals of the step; "
                                                        "the stages of signals still on the rocFFT pipeline share the timers)"}
                        else:
                            p_bytes = sum(5.0 * 32 * j["sig"].nfft * E * j["P"] * len(j["dop"]) * j["B"] for j in jobs)
                            rec["roofline"] = {"bound": "hbm", "kernels": "rocFFT double-precision pipeline: conj-multiply, inverse transform (>= 2 passes), magnitudes",
                                               "unit": "GB/s", "alg_bytes_per_step": p_bytes, "ms_per_step": dt / n * 1e3,
                                               "achieved": p_bytes / (dt / n) / 1e9, "peak": HBM_PEAK_GBPS, "frac": p_bytes / (dt / n) / 1e9 / HBM_PEAK_GBPS,
                                               "model": "five stage boundaries of 32 N bytes (16 N in, 16 N out) per correlation row and block over the step time"}
                finally:
                    eng.close()
            rec.update(rec.pop("f64_timing"))
            rec["f32_same_loop"] = rec.pop("f32")
            rec["f32_over_f64"] = rec["f32_same_loop"]["value"] / rec["value"]
            if "f64_pipeline" in rec:
                rec["rocfft_pipeline_same_loop"] = rec.pop("f64_pipeline")
                rec["split_form_over_rocfft_pipeline"] = rec["value"] / rec["rocfft_pipeline_same_loop"]["value"]
                a, b = peaks["f64"], peaks["f64_pipeline"]
                rec["split_form_vs_rocfft_pipeline_on_the_same_epochs"] = {
                    "searches": int(a.size), "peak_location_mismatches": int(((a["idx"] != b["idx"]) | (a["d_index"] != b["d_index"])).sum()),
                    "max_rel_metric_error": float(np.max(np.abs(a["metric"] - b["metric"]) / np.abs(b["metric"])))}
            a, b = peaks["f32"], peaks["f64"]
            rec["f32_vs_f64_on_the_same_epochs"] = {
                "searches": int(a.size), "peak_location_mismatches": int(((a["idx"] != b["idx"]) | (a["d_index"] != b["d_index"])).sum()),
                "max_rel_metric_error": float(np.max(np.abs(a["metric"] - b["metric"]) / np.abs(b["metric"])))}
        except Exception as exc:
            rec["error"] = repr(exc)[:300]
        out.append(rec)
    return out


_STREAM_CEILINGS = {}


def ab_n16384(args, env, rounds=4, reps=6):
    """In-run A/B of the two forms of the one-workgroup N = 16384 transform (GACQ_OPT_LDS_VARIANT: radix-32, 512 threads x 32 points,
    gacq_lds16k.hip, the default; 16 = radix-16, 1024 threads x 16 points) on config 5's two N = 16384 signals: same process, same engine
    context, same resident samples, the option flipped between short bursts so that box, clocks and allocation are shared."""
    from gnss_dsp_tools_amd import acquire
    dev = env["dev"]
    jobs = [j for j in build_jobs(CONFIGS[5], 1, dev) if j["sig"].nfft == 16384]
    eng = acquire.Engine(env["local_rank"])
    eng.use_torch_stream(dev)
    ms = {}
    try:
        for variant in (16, -1):                 # first use: code spectra, workspaces
            eng.set_option("lds_variant", variant)
            for j in jobs:
                eng.search_batch_dev(j["sig"], j["x"], j["flat"], j["dop"], j["B"])
        torch.cuda.synchronize(dev)
        for _ in range(rounds):
            for variant in (16, -1):
                eng.set_option("lds_variant", variant)
                for j in jobs:
                    eng.search_batch_dev(j["sig"], j["x"], j["flat"], j["dop"], j["B"])          # settle
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        eng.search_batch_dev(j["sig"], j["x"], j["flat"], j["dop"], j["B"])
                    torch.cuda.synchronize(dev)
                    ms.setdefault((j["label"], variant), []).append((time.perf_counter() - t0) / reps * 1e3)
    finally:
        eng.set_option("lds_variant", -1)
        eng.close()
    out = {"what": "ms per search of config 5's N = 16384 signals (B1I: 63 items x 200 bins x 10 blocks through forward + correlate kernels; "
                   "GLONASS L1: 15 channels x 200 x 10 through the fused kernel), median of %d alternating bursts of %d calls" % (rounds, reps)}
    for (label, variant), v in sorted(ms.items()):
        out["%s_radix%d_ms" % (label.replace("-", "_"), 16 if variant == 16 else 32)] = float(np.median(v))
    return out


def stream_ceilings(eng):
    """GB/s a tuned streaming kernel reaches on THIS device, measured now (gacq_stream_probe: eight 16-byte non-temporal accesses in
    flight per lane, 1 GiB per launch, 8 timed launches per figure): fill (stores), read, copy (read + write bytes).  Cached per process."""
    if not _STREAM_CEILINGS:
        for kind in ("fill", "read", "copy"):
            _STREAM_CEILINGS[kind] = eng.stream_probe(kind, 1 << 30, 8)
    return dict(_STREAM_CEILINGS)


def next_rows(args, env, seconds=0.5):
    """The SURVEY section 8f rows that sit either side of the FFT search -- the front-end in front of it, the long-code searches and the
    tracking correlators behind it -- timed for >= `seconds` each on device-resident inputs (the reference keeps one x in memory along
    that chain), so that the driver-visible line carries them."""
    from gnss_dsp_tools_amd import acquire, longcode, signals, synth, tracking
    dev = env["dev"]
    eng = acquire.Engine(env["local_rank"])
    eng.use_torch_stream(dev)
    rows = {}

    def timed(fn, sync=True):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds or n < 3:
            fn()
            n += 1
            if sync and n % 8 == 0:
                torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n, n
    try:
        # front-end: 85 ms of int8 IQ at 69.984 MS/s (the reference's example recording rate and default --time 80 + 5 ms,
        # acquire-gps-l1.py:50-52,67,80) -> fixed-point NCO mix -> filtfilt(161 taps) -> linear-interpolation resample, resident input
        fs, coff, ms_pad = 69984000.0, -9334875.0, 85
        n = int(fs * 0.001 * ms_pad)
        rng = np.random.default_rng(1)
        iq = torch.from_numpy(rng.integers(-90, 90, size=2 * n, dtype=np.int8)).to(dev)
        sig = signals.get("gps-l1")
        out = eng.frontend_dev(sig, iq, fs, coff, ms_pad)
        dt, reps = timed(lambda: eng.frontend_dev(sig, iq, fs, coff, ms_pad))
        L = n + 2 * 483
        alg = n * (2 + 8) + (n * 8 + L * 8) + (L * 8 + n * 8) + out.numel() * (16 + 8)       # mix, FIR forward, FIR backward, resample
        mac = 2.0 * 161 * 4 * L                                                              # two passes x 161 taps x one real x complex MAC (4 flop)
        rows["frontend"] = {"what": "int8 IQ 69.984 MS/s x 85 ms -> mix -> filtfilt(161) -> resample to %.3f MS/s (gps-l1), resident input" % (sig.fs / 1e6),
                            "ms": dt * 1e3, "calls_timed": reps, "input_bytes_per_s": 2.0 * n / dt, "Msamples_in_per_s": n / dt / 1e6,
                            "algorithmic_GBps": alg / dt / 1e9, "frac_hbm_8TBps": alg / dt / 1e9 / HBM_PEAK_GBPS,
                            "fir_TFLOPps": mac / dt / 1e12, "frac_fp32_vector_peak": mac / dt / 1e12 / VALU_PEAK_TFLOPS, "x_realtime": ms_pad * 1e-3 / dt,
                            "bound": "valu (direct-form FIR: 2 x 161 real x complex MACs per sample)"}
        # long-code searches on a device-resident block (gacq_longcode_search_dev)
        fs_l, ms_l = 4096000.0, 100
        nl = int(fs_l * 0.020)
        xl = torch.from_numpy(synth.make_longcode_iq("gps.l2cl", 7, 511500.0, longcode.L2CL_LENGTH, fs_l, (ms_l // 20) * nl, 11, 0.4, 1234.0,
                                                     37 * 10230 + 4000.25)).to(dev)
        dt, reps = timed(lambda: longcode.search_l2cl(xl, 7, 1234.0, 4000.25, ms_l, fs_l, engine=eng), sync=False)
        got = longcode.search_l2cl(xl, 7, 1234.0, 4000.25, ms_l, fs_l, engine=eng)
        rows["longcode_l2cl"] = {"what": "acquire-gps-l2cl.py search(): 75 candidates x 5 blocks x 81920 samples, resident input", "ms": dt * 1e3,
                                 "calls_timed": reps, "candidate_samples_per_s": 75 * (ms_l // 20) * nl / dt, "k_found": int(got[1]), "k_injected": 37}
        fs_p, ms_p = 16384000.0, 20
        npp = int(fs_p * 0.004)
        xp = torch.from_numpy(synth.make_longcode_iq("glonass.p", 0, 5110000.0, longcode.P_LENGTH, fs_p, (ms_p // 4) * npp, 12, 0.4,
                                                     562500.0 * 2 + 800.0, 5110 * 321 + 10 * 100.5)).to(dev)
        dt, reps = timed(lambda: longcode.search_glonass_p(xp, 2, 800.0, 100.5, ms_p, fs_p, engine=eng), sync=False)
        got = longcode.search_glonass_p(xp, 2, 800.0, 100.5, ms_p, fs_p, engine=eng)
        rows["longcode_glonass_p"] = {"what": "acquire-glonass-l1-p.py search(): 1000 candidates x 5 blocks x 65536 samples, resident input",
                                      "ms": dt * 1e3, "calls_timed": reps, "candidate_samples_per_s": 1000 * (ms_p // 4) * npp / dt,
                                      "k_found": int(got[1]), "k_injected": 321}
        # tracking correlators: E/P/L of 12 GPS L1 C/A satellites over one resident 1 ms block (track-gps-l1.py:48-50), one launch per call
        rng = np.random.default_rng(5)
        xb = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
        xbd = torch.from_numpy(xb).to(dev)
        prns = np.arange(1, 13)
        code_p = rng.uniform(0, 1023, 12)
        cf = 1.023e6 / 4096000.0 * (1 + rng.uniform(-2e-6, 2e-6, 12))
        plan = tracking.EplPlan("gps.ca", prns, 0.05, engine=eng)
        dt_dev, reps = timed(lambda: plan(xbd, code_p, cf), sync=False)
        dt_host, _ = timed(lambda: plan(xb, code_p, cf), sync=False)
        rows["tracking_epl"] = {"what": "early/prompt/late of 12 satellites over one 4096-sample block (36 correlators), one launch per call "
                                        "(tracking.EplPlan through Python; the bare C call is ~4 us less)",
                                "us_resident_block": dt_dev * 1e6, "us_host_block": dt_host * 1e6, "calls_timed": reps,
                                "fraction_of_the_1ms_block": dt_dev / 1e-3}
    finally:
        eng.close()
    return rows


def emulate_ranks(args, env):
    """PROJECTION, not a measurement (no multi-GPU node was available): every rank's share of an N-rank job is run on this one
    GPU, one rank after the other -- the same Doppler slices ShardedSearch cuts (sharded.doppler_bounds), the same epochs (weak
    scaling: N x EPOCHS, strong: EPOCHS), the same kernels.  Reports each slice's step time and dominant-kernel roofline
    fraction, the slowest slice (which would pace a real step), the exchange size, and what that projects to."""
    from gnss_dsp_tools_amd import acquire, sharded
    dev, N = env["dev"], args.emulate_ranks
    cfg = CONFIGS[args.config]
    epochs = args.epochs or cfg["epochs"]
    E_total = epochs * N if args.scaling == "weak" else epochs
    jobs = build_jobs(cfg, E_total, dev)
    eng = acquire.Engine(env["local_rank"], engine=args.engine)
    eng.use_torch_stream(dev)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    def step(r, nranks, e_count):
        for job in jobs:
            b = sharded.doppler_bounds(len(job["dop"]), nranks)
            dsl = job["dop"][b[r]:b[r + 1]]
            x = job["x"][:e_count]
            if job["family"]:
                eng.search_family_batch_dev(job["family"], x, job["items"], dsl, job["B"])
            else:
                eng.search_batch_dev(job["sig"], x, job["items"], dsl, job["B"])

    def timed(r, nranks, e_count, seconds=0.5):
        """best of two windows of >= seconds/2 each (the first window after a change of shape can still see the clock settle)"""
        for _ in range(2):
            step(r, nranks, e_count)
        torch.cuda.synchronize(dev)
        best = None
        for _ in range(2):
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds / 2 or n < 3:
                step(r, nranks, e_count)
                torch.cuda.synchronize(dev)
                n += 1
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None else min(best, dt)
        return best

    def stage_fracs(r, nranks, e_count):
        """dominant stage (most time per step) of this slice and its roofline fraction"""
        eng.set_profiling(True)
        best = None
        for job in jobs:
            b = sharded.doppler_bounds(len(job["dop"]), nranks)
            D_local = b[r + 1] - b[r]
            eng.reset_stage_times()
            for _ in range(3):
                x = job["x"][:e_count]
                if job["family"]:
                    eng.search_family_batch_dev(job["family"], x, job["items"], job["dop"][b[r]:b[r + 1]], job["B"])
                else:
                    eng.search_batch_dev(job["sig"], x, job["items"], job["dop"][b[r]:b[r + 1]], job["B"])
            torch.cuda.synchronize(dev)
            st = eng.stage_times()
            N_, P, B, F = job["sig"].nfft, job["P"], job["B"], job["F"]
            fused16k = job["kind"] == "lds" and N_ == 16384 and F == P
            fused4k = job["kind"] == "lds" and N_ == 4096 and not st["mix_nco"][1] and st["lds_correlate"][1] > 0
            for sname, (tot_ms, nl) in st.items():
                if not nl:
                    continue
                bound, work = stage_model(job["kind"], sname, N_, P, D_local, B, F, e_count, fused16k, fused4k)
                if bound is None:
                    continue
                ms_step = tot_ms / 3
                rate = work / (ms_step * 1e-3)
                frac = rate / 1e12 / VALU_PEAK_TFLOPS if bound == "valu" else rate / 1e9 / HBM_PEAK_GBPS
                if best is None or ms_step > best["ms_per_step"]:
                    best = {"signal": job["label"], "stage": sname, "bound": bound, "ms_per_step": ms_step, "frac": frac, "D_local": D_local}
        eng.set_profiling(False)
        return best

    cells_rank0 = lambda nranks, e: sum(e * j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
    t_one = timed(0, 1, epochs)                                   # the 1-GPU reference point: whole grid, EPOCHS epochs
    one = {"ms_per_step": t_one * 1e3, "value": cells_rank0(1, epochs) / t_one, "dominant": stage_fracs(0, 1, epochs)}
    per_rank = []
    for r in range(N):
        t = timed(r, N, E_total)
        per_rank.append({"rank": r, "ms_per_step": t * 1e3, "doppler_bins": [sharded.doppler_bounds(len(j["dop"]), N)[r + 1] - sharded.doppler_bounds(len(j["dop"]), N)[r] for j in jobs],
                         "dominant": stage_fracs(r, N, E_total)})
    eng.close()
    slowest = max(p["ms_per_step"] for p in per_rank)
    total_cells = cells_rank0(N, E_total)
    exch = N * sum(16 * E_total * j["P"] for j in jobs)            # every rank receives N shards of 16-byte records
    proj = total_cells / (slowest * 1e-3)
    return {"projection": True,
            "note": "PROJECTION, NOT A MEASUREMENT: the %d ranks' slices were run one after the other on ONE MI355X; a real step is paced by "
                    "the slowest rank plus one all-gather of %d bytes per rank (latency-bound, overlapped with the next step)" % (N, exch),
            "baseline_config": args.config, "workload": cfg["label"], "ranks_emulated": N, "scaling": args.scaling,
            "epochs_per_step": E_total, "one_gpu": one, "per_rank": per_rank, "slowest_slice_ms": slowest,
            "projected_value": proj, "unit": "cells/s", "projected_speedup_over_one_gpu": proj / one["value"],
            "projected_efficiency": proj / one["value"] / N, "exchange_bytes_received_per_rank_per_step": exch}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (2 = the headline metric)")
    ap.add_argument("--epochs", type=int, default=0, help="epochs per GPU per step (default: 1024 / 2 / 2 / 1 for config 2 / 3 / 4 / 5)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: N x EPOCHS epochs per step; strong: EPOCHS epochs per step whatever N is")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 rocFFT pipeline, 2 LDS FFT kernels, 3/4 split engines")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="gacq_set_option tuning switch (e.g. lds_variant=3)")
    ap.add_argument("--distinct-epochs", type=int, default=0, help="N = 4096 signals: distinct seeded epochs per step, tiled to the batch (default: all 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=2,
                    help="steps in flight: engine contexts, each on a HIP stream with a hardware queue of its own, that consecutive "
                         "(independent) steps alternate between.  2 (default): the end of one step -- ragged last round of workgroups, the "
                         "small launches that close a search -- runs under the next step's kernels, and memory-bound and arithmetic-bound "
                         "searches of a mixed step overlap; the line carries the one-step-in-flight time measured in the same process "
                         "(roofline.ab.steps_in_flight).  Kernel durations for the roofline come from the profiling pass, which keeps ONE "
                         "step in flight: overlapped launches have no meaningful duration of their own")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-epoch host-call latency probe (profiling runs: keeps every launch the bench workload)")
    ap.add_argument("--sustained-s", type=float, default=1.5, help="length of the second, sustained timed region (0 = skip)")
    ap.add_argument("--preroll-s", type=float, default=0.3, help="untimed load before the warm-up steps so the GPU has clocked up (0 = none)")
    ap.add_argument("--no-self-check", action="store_true", help="variant builds that compute garbage on purpose (tools/build_variant.sh)")
    ap.add_argument("--force-gather", action="store_true", help="run the all-gather + merge even on 1 rank (test aid)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the live rocprofv3 --pmc child runs that measure the dominant kernel's HBM traffic (N = 1 only; ~15 s each)")
    ap.add_argument("--no-others", action="store_true",
                    help="headline run only: skip the short runs of configs 3, 4, 5 and the complex128 (engine 5) leg that the default "
                         "single-GPU config-2 run appends as other_configs / reference_precision")
    ap.add_argument("--emulate-ranks", type=int, default=0, metavar="N",
                    help="PROJECTION, not a measurement: on this one GPU, time every rank's slice of an N-rank job (the Doppler slices "
                         "ShardedSearch would cut) one after the other and report the slowest slice, per-slice roofline and exchange bytes")
    ap.add_argument("--dry-bins", type=int, default=8, help="--dry-run-cpu: Doppler bins of the miniature grid (fewer than ranks leaves ranks without a slice)")
    ap.add_argument("--dry-run-cpu", default="", metavar="FILE.py:FUNCTION",
                    help="NOT A MEASUREMENT: run the N-rank launch path on CPU tensors over gloo with FUNCTION(name, x, items, dopplers, blocks) "
                         "-> peaks as the per-rank compute (tests pass an oracle-backed stand-in); no GPU is touched")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)     # launched by torch.distributed.run
    if args.dry_run_cpu:
        import importlib.util
        path, fname = args.dry_run_cpu.rsplit(":", 1)
        spec = importlib.util.spec_from_file_location("gacq_dry_run_compute", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        env = {"world": world, "rank": rank, "local_rank": local_rank, "use_dist": use_dist, "dev": torch.device("cpu")}
        out = run_dry(args, env, getattr(mod, fname))
        if rank == 0:
            print(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    env = {"world": world, "rank": rank, "local_rank": local_rank, "use_dist": use_dist, "dev": dev}

    if args.emulate_ranks:
        out = emulate_ranks(args, env)
        if rank == 0:
            print(json.dumps(out))
        return

    t_start = time.perf_counter()
    out = run(args, env)
    # The default single-GPU headline run also measures the other BASELINE configurations (short runs of the same code path:
    # >= 1 s timed each, dominant kernel, roofline fraction, HBM traffic measured next to the algorithmic bytes) and the
    # headline workload in the reference's own arithmetic type (complex128, engine 5), so that the driver-recorded line
    # carries them.
    headline = args.config == 2 and world == 1 and args.engine == 0 and not args.no_others and not os.environ.get("GACQ_BENCH_PMC_CHILD")
    if headline and rank == 0:
        others = []
        for k in (3, 4, 5):
            a = argparse.Namespace(**vars(args))
            a.config, a.epochs, a.steps, a.warmup, a.sustained_s, a.preroll_s = k, 0, 5, 1, 1.0, 0.2
            a.no_cpu_baseline = a.no_latency = True
            t1 = time.perf_counter()
            try:
                others.append(condense(run(a, env), time.perf_counter() - t1))
            except Exception as exc:                     # the headline line must survive a failing side run
                others.append({"baseline_config": k, "error": repr(exc)[:300]})
        out["other_configs"] = others
        try:
            out["reference_precision"] = reference_precision(args, env, out["value"])
            out["reference_precision"]["other_configs"] = reference_precision_other_configs(args, env)
        except Exception as exc:
            out.setdefault("reference_precision", {})["error"] = repr(exc)[:300]
        try:
            out["next_rows"] = next_rows(args, env)
        except Exception as exc:
            out["next_rows"] = {"error": repr(exc)[:300]}
        # in-run A/Bs of the levers this round claims or rules out, inside `roofline` so that they travel with the parsed line
        ab = {}
        try:
            ab["n16384_transform"] = ab_n16384(args, env)
        except Exception as exc:
            ab["n16384_transform"] = {"error": repr(exc)[:300]}
        if out["config"].get("one_step_in_flight"):
            ab["steps_in_flight"] = {"one_ms_per_step": out["config"]["one_step_in_flight"]["ms_per_step"], "%d_ms_per_step" % out["config"]["steps_in_flight"]: out["ms_per_step"]}
        ts = out.get("tie_safe") or {}
        if "ms_per_step_with_tie_safe_off" in ts:
            ab["tie_safe_locations"] = {"on_ms_per_step": out["ms_per_step"], "off_ms_per_step": ts["ms_per_step_with_tie_safe_off"]}
        out["roofline"]["ab"] = ab
        out["bench_wall_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def build_jobs(cfg, E_total, dev, distinct=0):
    """One job per signal of a BASELINE configuration, samples resident in HBM."""
    from gnss_dsp_tools_amd import acquire, signals, synth
    jobs = []
    for name, items, ds, ms in cfg["jobs"]:
        family = name if isinstance(name, tuple) else None           # signals that share everything but their code tables
        sig = signals.get(family[0] if family else name)
        B = ms[1] if isinstance(ms, tuple) else sig.blocks(ms)
        dop = acquire.doppler_grid(ds)
        cpu_items = [(n, it) for n, lst in zip(family, items) for it in lst] if family else [(sig.name, it) for it in items]
        flat = [it for lst in items for it in lst] if family else list(items)
        sats = synth.default_sats(items[0] if family else items)     # family: satellites of the first signal (first rows of the stack)
        nsamp = sig.samples_needed(B)
        # N = 4096 (the headline): every epoch of the step is its own seeded noise realisation (32 MB for 1024 epochs), so that the
        # ~2e-4 near-ties per noise-only search of real data are in the timed region and the tie-safe re-evaluation actually runs
        # (round 4 tiled 8 epochs 128 x: 256 distinct noise-only searches, none of them ambiguous).  The long lengths tile two epochs
        # (content does not change the work there and an epoch is 0.5-1.5 MB of host arithmetic).
        base = synth.make_epochs(sig, B, synth.BASE_SEED + cfg["seed"] + 100 * len(jobs), sats, min((distinct or 1024) if sig.nfft <= 4096 else 2, E_total), nsamp=nsamp)
        xs = np.concatenate([base] * ((E_total + len(base) - 1) // len(base)))[:E_total]
        jobs.append({"sig": sig, "name": sig, "family": family, "items": items, "flat": flat, "P": len(flat), "cpu_items": cpu_items, "ds": ds, "ms": ms,
                     "B": B, "dop": dop, "dopplers": dop, "blocks": B, "sats": sats, "host": base, "xs": xs,
                     "x": torch.from_numpy(np.ascontiguousarray(xs)).to(dev), "F": len(flat) if sig.bias_hz else 1, "kind": engine_kind(sig.nfft),
                     "label": "+".join(family) if family else sig.name})
    return jobs


def make_run_steps(lanes, jobs):
    """k independent steps, alternating between `lanes` = [(context factory, ShardedSearch)].  With N > 1 the all-gather of step i
    runs under the local compute of step i+1 (asynchronous collective, merge deferred by one step)."""
    def run_steps(k):
        pend = [None] * len(lanes)
        out = None
        for i in range(k):
            ctx, shl = lanes[i % len(lanes)]
            with ctx():
                nxt = shl.search_jobs_async(jobs)                             # next search queued before the previous merge:
                if pend[i % len(lanes)] is not None:                          # its kernels cover the previous exchange
                    out = pend[i % len(lanes)].wait()
                pend[i % len(lanes)] = nxt
        for (ctx, _), p in zip(lanes, pend):
            if p is not None:
                with ctx():
                    out = p.wait()
        return out
    return run_steps


def make_timed(run_steps, dev, use_dist):
    """timed(k) -> (last merged result, seconds): exactly k steps bracketed by a device synchronise + barrier on both sides, MAX over
    ranks (the contract the driver's scaling runs rely on)."""
    def sync_dev():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def timed(k):
        # the interpreter's cyclic collector is run BEFORE the region and held off inside it: a full collection of this process (tens
        # of thousands of tracked objects after the imports) takes ~35 ms and otherwise lands in whichever region the allocation count
        # happens to cross its threshold in (tools/exp_cfg4_steps.py: one 38.8 ms step among 160 of 3.2 ms)
        import gc
        gc.collect()
        gc.disable()
        try:
            run_steps(8)                 # untimed: the collection above idled the GPU for tens of ms, long enough for its clocks to drop
            sync_dev()
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            out = run_steps(k)
            sync_dev()
            if use_dist:
                dist.barrier()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return out, dt
    return timed


def shard_census(sh, jobs, world, dev):
    """One more step with the un-merged exchange buffer kept: every rank must hold `world` shards, and noise alone gives every
    (epoch, item) of every shard a positive metric, so an all-positive shard is one that really arrived.  Returns (shards, merged)."""
    pj = sh.search_jobs_async(jobs)
    g = pj.shards()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    assert g.shape[0] == world, ("exchange buffer has %d shards, world is %d" % (g.shape[0], world))
    recs = g.view(world, -1, 2)
    off = 0
    for job in jobs:                           # per job: a rank that owns Doppler bins must have sent positive metrics, one that owns none
        from gnss_dsp_tools_amd import sharded as _sh      # (grid with fewer bins than ranks) "nothing found" records: metric 0
        b = _sh.doppler_bounds(len(job["dop"]), world)
        cnt = int(job["x"].shape[0]) * job["P"]
        for r in range(world):
            m = recs[r, off:off + cnt, 0]
            if b[r + 1] > b[r]:
                assert bool((m > 0).all()), "shard %d of the all-gather arrived empty" % r
            else:
                assert bool((m == 0).all()), "shard %d owns no Doppler bin but sent peaks" % r
        off += cnt
    return int(g.shape[0]), pj.wait()


def run_dry(args, env, local_fn):
    """--dry-run-cpu MODULE.py:FUNCTION -- NOT A MEASUREMENT.  The driver-style launch of `bench.py --gpus N` (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment, one process per rank) on CPU tensors over gloo, with the per-rank compute supplied by
    the caller (tests/ pass an oracle-backed stand-in; the product engine needs a GPU and has no CPU fallback).  Exercises what the
    8-GPU scaling run depends on before hardware ever sees it: the rank environment, Doppler slicing per rank (ShardedSearch), the
    single all-gather with two steps in flight, the tie-exact merge, the shard census, the barrier-bracketed MAX-reduced timing and
    the shape of the JSON line.  The workload is a miniature of the chosen BASELINE configuration (3 items, 8 Doppler bins)."""
    world, rank, dev = env["world"], env["rank"], env["dev"]
    from gnss_dsp_tools_amd import acquire, sharded, signals, synth
    cfg = CONFIGS[args.config]
    epochs = args.epochs or 1
    E_total = epochs * world if args.scaling == "weak" else epochs
    jobs = []
    members = []
    for name, items, ds, ms in cfg["jobs"]:
        if isinstance(name, tuple):            # a family (E1B + E1C): the stacked device signal needs a GPU; here one job per member, same samples
            members += [(n, it, ds, ms) for n, it in zip(name, items)]
        else:
            members.append((name, items, ds, ms))
    for name, items, ds, ms in members:
        sig = signals.get(name)
        B = min(2, ms[1] if isinstance(ms, tuple) else sig.blocks(ms))
        items = list(items)[:3]
        dop = acquire.doppler_grid(ds)[:args.dry_bins]
        xs = synth.make_epochs(sig, B, synth.BASE_SEED + cfg["seed"], synth.default_sats(items), E_total, nsamp=sig.samples_needed(B))
        jobs.append({"sig": sig, "name": sig.name, "family": None, "items": items, "P": len(items), "dop": dop, "dopplers": dop, "blocks": B, "B": B,
                     "x": torch.from_numpy(xs), "label": sig.name})
    cells_step = sum(E_total * j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
    sh = sharded.ShardedSearch(local_fn=local_fn, always_gather=args.force_gather)
    import contextlib
    run_steps = make_run_steps([(contextlib.nullcontext, sh)], jobs)
    timed = make_timed(run_steps, dev, env["use_dist"])
    run_steps(args.warmup)
    merged, dt = timed(args.steps)
    shards_seen = 1
    if not sh._solo():
        shards_seen, merged = shard_census(sh, jobs, world, dev)
    # every rank holds the same merged records; rank 0 checks them against a single-rank scan of the whole grid
    ok = True
    if rank == 0:
        for job, m in zip(jobs, merged):
            whole = local_fn(job["name"], job["x"], job["items"], job["dop"], job["B"])
            ok = ok and bool((m.view(torch.int64) == whole.view(torch.int64)).all())
    if rank != 0:
        return None
    return {"dry_run": True, "note": "NOT A MEASUREMENT: CPU tensors over gloo, per-rank compute supplied by the caller (" + args.dry_run_cpu + ")",
            "metric": "acquisition cells/s (PRN x Doppler x code-phase), " + "+".join(j["label"] for j in jobs), "value": None, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64 (stand-in)", "data": "synthetic",
            "config": {"workload": "miniature of BASELINE config %d (3 items, %d Doppler bins, %d epoch(s)/step)" % (args.config, args.dry_bins, E_total),
                       "baseline_config": args.config, "signals": [j["label"] for j in jobs], "epochs_per_step": E_total, "cells_per_step": cells_step,
                       "doppler_bins_per_rank": [[sharded.doppler_bounds(len(j["dop"]), world)[r + 1] - sharded.doppler_bounds(len(j["dop"]), world)[r]
                                                  for r in range(world)] for j in jobs],
                       "shards_seen_by_every_rank": shards_seen, "merged_equals_single_rank_scan": ok}}


def run(args, env):
    """One benchmark of one BASELINE configuration: returns the JSON line's dict (rank 0; other ranks return None)."""
    world, rank, local_rank, use_dist, dev = env["world"], env["rank"], env["local_rank"], env["use_dist"], env["dev"]
    from gnss_dsp_tools_amd import acquire, sharded, signals, synth

    cfg = CONFIGS[args.config]
    epochs = args.epochs or cfg["epochs"]
    E_total = epochs * world if args.scaling == "weak" else epochs
    jobs = build_jobs(cfg, E_total, dev, args.distinct_epochs)
    cells_step = sum(E_total * j["P"] * len(j["dop"]) * j["sig"].nfft for j in jobs)
    cell_blocks_step = sum(E_total * j["P"] * len(j["dop"]) * j["sig"].nfft * j["B"] for j in jobs)

    def make_engine():
        e = acquire.Engine(local_rank, engine=args.engine)
        for kv in args.option:
            k, v = kv.split("=")
            e.set_option(k, int(v))
        return e

    eng = make_engine()
    eng.use_torch_stream(dev)                           # same stream as the RCCL collective -> ordered
    sh = sharded.ShardedSearch(engine=eng, always_gather=args.force_gather)
    exchanged = not sh._solo()

    # Steps are independent searches (a receiver scanning a recording keeps several batches in flight).  With N > 1 the
    # all-gather of step i runs under the kernels of step i+1 (asynchronous collective, merge deferred by one step); with
    # --lanes 2 the steps also alternate between two engine contexts with their own HIP streams and workspaces.  Every step
    # runs all of its kernels, the exchange and the merge inside the timed region.
    lanes, own_queues, lane_queue_note = [], [], "one hardware queue per lane (streams created with a full CU mask)"
    for _ in range(max(1, args.lanes)):
        # a stream with a hardware queue of its own: plain HIP streams are multiplexed over a few queues, and two lanes that land on the
        # same one would run one after the other (acquire.MaskedStream; tools/exp_lanes_debug.py: the overlap was there or not by luck)
        try:
            own_queues.append(acquire.MaskedStream(local_rank))
            st = own_queues[-1].torch_stream
        except Exception as exc:                       # no CU-mask streams on this system: plain streams (the lanes may then share a queue)
            lane_queue_note = "plain torch streams (gacq_stream_create_cu_mask failed: %s)" % repr(exc)[:160]
            st = torch.cuda.Stream(dev)
        e2 = make_engine()
        with torch.cuda.stream(st):
            lanes.append((st, e2, sharded.ShardedSearch(engine=e2, always_gather=args.force_gather)))

    run_steps = make_run_steps([((lambda st=st: torch.cuda.stream(st)), shl) for st, _, shl in lanes], jobs)
    timed = make_timed(run_steps, dev, use_dist)

    # Clock ramp: after the idle seconds of start-up (signal build, H2D) the GPU needs tens of milliseconds of load before it
    # clocks up (tools/exp_cfg2.py timeline: 0.66 -> 0.45 -> 0.41 ms per step over the first ~100 steps after 0.5 s idle, flat
    # 0.41-0.42 without the idle gap).  An untimed pre-roll of the same steps puts the chip in its sustained state before the
    # W warm-up steps and the K timed steps; its length is reported in the JSON line.
    preroll = {"seconds": 0.0, "steps": 0}
    if args.preroll_s > 0:
        run_steps(1)                                     # first call: code spectra, FFT plans, workspaces (not load, not counted)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        per = 16
        while time.perf_counter() - t0 < args.preroll_s:
            t1 = time.perf_counter()
            run_steps(per)                               # long uninterrupted stretches: the ramp needs tens of ms of continuous load
            torch.cuda.synchronize(dev)
            preroll["steps"] += per
            if time.perf_counter() - t1 < 0.05:
                per *= 2
        preroll["seconds"] = time.perf_counter() - t0
    run_steps(args.warmup)
    merged, dt = timed(args.steps)
    one_lane = None
    if len(lanes) > 1:
        # the same K steps with ONE step in flight (lane 0 only), same process, right after the timed region: what the second step in
        # flight is worth (roofline.ab.steps_in_flight) and the step time the profiling pass's kernel durations belong to
        _, dt_one = make_timed(make_run_steps([((lambda st=lanes[0][0]: torch.cuda.stream(st)), lanes[0][2])], jobs), dev, use_dist)(args.steps)
        one_lane = {"steps_in_flight": 1, "ms_per_step": dt_one / args.steps * 1e3, "value": cells_step * args.steps / dt_one}
    plain_ratio = None
    if use_dist and world == 1:
        # launched by torch.distributed.run with one rank: the same K steps once more without the barrier / MAX-reduce bracket of the
        # distributed contract -- the two must agree, or the launch path itself costs something
        _, dt_plain = make_timed(run_steps, dev, False)(args.steps)
        plain_ratio = dt_plain / dt

    # Second, longer timed region: the K-step region above can be a few milliseconds and sits in boost clocks; this one runs
    # the same steps for >= --sustained-s seconds so that DVFS cannot flatter the number (reported beside `value`).
    sustained = None
    clocks = None
    if args.sustained_s > 0:
        k_sus, per_step = args.steps, dt / args.steps
        for _ in range(4):                               # the first estimate of the step time may be off: repeat until long enough
            k_sus = max(args.steps, int(np.ceil(1.05 * args.sustained_s / per_step)))
            if use_dist:
                kt = torch.tensor([k_sus], dtype=torch.int64, device=dev)
                dist.all_reduce(kt, op=dist.ReduceOp.MAX)
                k_sus = int(kt.item())
            sampler = ClockSampler(local_rank).start() if rank == 0 else None
            _, dt_sus = timed(k_sus)
            clocks = sampler.stop() if sampler else None
            per_step = dt_sus / k_sus
            if dt_sus >= args.sustained_s:
                break
        sustained = {"steps": k_sus, "seconds": dt_sus, "ms_per_step": dt_sus / k_sus * 1e3, "value": cells_step * k_sus / dt_sus, "clocks": clocks}
        if one_lane is not None:
            # the one-step-in-flight comparison over a region of the same length (same clocks regime as `sustained`)
            _, dt_one_sus = make_timed(make_run_steps([((lambda st=lanes[0][0]: torch.cuda.stream(st)), lanes[0][2])], jobs), dev, use_dist)(k_sus)
            one_lane["sustained"] = {"steps": k_sus, "seconds": dt_one_sus, "ms_per_step": dt_one_sus / k_sus * 1e3, "value": cells_step * k_sus / dt_one_sus}

    # ---- correctness of what was just computed (not timed) ----------------------------------------------------------
    # (a) every rank received `world` shards: one more step with the un-merged exchange buffer kept -- noise alone gives
    #     every (epoch, item) of every shard a positive metric, so an all-positive shard is one that really arrived
    shards_seen = 1
    if exchanged:
        shards_seen, merged = shard_census(sh, jobs, world, dev)
    # (b) the strong injected satellites sit at their delays in epoch 0 (padded searches see two code periods: n-d or 2n-d)
    if not args.no_self_check:
        for job, m in zip(jobs, merged):
            pk = m[:1].cpu().numpy().view(acquire.PEAK_DTYPE).reshape(job["P"])
            n = job["sig"].n
            for it, amp, f, delay in job["sats"]:
                if amp >= 0.25:
                    got = int(pk["idx"][job["flat"].index(it)])
                    assert got % n == (-delay) % n, ("bench self-check failed", job["sig"].name, it, got, delay)

    # ---- per-kernel durations: HIP events on the launch stream, in a pass that reproduces the timed region ------------------------
    # Stage timers are per engine context, so every job gets its own profiling engine (same stream, same options, same call path) and
    # the pass runs whole steps -- all jobs back to back, pipelined like run_steps(), for >= 0.5 s -- so that the kernels are timed
    # in the thermal / clock state of the timed region and next to the same neighbours.  (Round 2 timed one job at a time in a short
    # loop: for the four-signal config 5 that read 11 % faster than the rocprofv3 trace of the timed region.)
    bounds_of = lambda D: sharded.doppler_bounds(D, world)
    st0, eng0, sh0 = lanes[0]
    prof_steps = max(3, min(10, args.steps), int(np.ceil(0.5 / max(dt / args.steps, 1e-6))))
    if use_dist:
        kt = torch.tensor([prof_steps], dtype=torch.int64, device=dev)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        prof_steps = int(kt.item())
    if len(jobs) == 1:
        prof = [(eng0, sh0)]
    else:
        prof = []
        with torch.cuda.stream(st0):
            for _ in jobs:
                ej = make_engine()
                prof.append((ej, sharded.ShardedSearch(engine=ej, always_gather=args.force_gather)))
    with torch.cuda.stream(st0):
        for (ej, shj), job in zip(prof, jobs):
            shj.search_jobs_async([job]).wait()          # settle (signal build, grid upload) before the events start
        torch.cuda.synchronize(dev)
        for ej, _ in prof:
            ej.set_profiling(True)
            ej.reset_stage_times()
        pend = [None] * len(jobs)
        t_prof = time.perf_counter()
        for _ in range(prof_steps):
            for ji, ((ej, shj), job) in enumerate(zip(prof, jobs)):
                nxt = shj.search_jobs_async([job])
                if pend[ji] is not None:
                    pend[ji].wait()
                pend[ji] = nxt
        for pj_ in pend:
            pj_.wait()
    torch.cuda.synchronize(dev)
    prof_ms_per_step = (time.perf_counter() - t_prof) / prof_steps * 1e3
    per_job = []
    for (ej, _), job in zip(prof, jobs):
        b = bounds_of(len(job["dop"]))
        D_local = b[rank + 1] - b[rank]
        stages = ej.stage_times()
        ej.set_profiling(False)
        N, P, B, F = job["sig"].nfft, job["P"], job["B"], job["F"]
        fused16k = job["kind"] == "lds" and N == 16384 and F == P
        fused4k = job["kind"] == "lds" and N == 4096 and not stages["mix_nco"][1] and stages["lds_correlate"][1] > 0
        st_out = {}
        for sname, (tot_ms, nl) in stages.items():
            if not nl:
                continue
            bound, work = stage_model(job["kind"], sname, N, P, D_local, B, F, E_total, fused16k, fused4k)
            st_out[sname] = {"avg_ms": tot_ms / nl, "launches_per_step": nl / prof_steps, "ms_per_step": tot_ms / prof_steps,
                             "bound": bound, "work_per_step": work}
        per_job.append({"signal": job["label"], "engine": job["kind"], "P": P, "D_local": D_local, "B": B, "N": N, "F": F, "stages": st_out,
                        "fused_forward": bool(fused16k or fused4k)})
    if len(jobs) > 1:
        for ej, _ in prof:
            ej.close()

    # dominant kernel = the (signal, stage) with the most time per step
    cand = [(s["ms_per_step"], pj, sname) for pj in per_job for sname, s in pj["stages"].items() if s["bound"]]
    _, dj, dstage = max(cand, key=lambda c: c[0])
    dk = dj["stages"][dstage]
    work_launch = dk["work_per_step"] / dk["launches_per_step"]
    kernel_name = {("lds", "lds_correlate", 4096): "lds_correlate_kernel", ("lds", "lds_correlate", 16384): "r32_correlate_kernel / r32_fused_kernel",
                   ("lds", "mix_nco", 4096): "lds_forward_kernel", ("lds", "mix_nco", 16384): "r32_forward_kernel",
                   ("split31", "lds_correlate"): "pfa_inner_corr_kernel", ("split_lds", "lds_correlate"): "lds_inner_correlate_kernel",
                   ("split31", "mag_peak"): "pfa_outer_inverse_kernel", ("split_lds", "mag_peak"): "split_outer_inverse_kernel",
                   ("split31", "mix_nco"): "pfa_outer_forward_kernel + pfa_inner_forward_kernel", ("split_lds", "mix_nco"): "split_outer_forward_kernel + lds_inner_forward_kernel"}
    kname = kernel_name.get((dj["engine"], dstage, dj["N"])) or kernel_name.get((dj["engine"], dstage)) or dstage
    if dj["engine"] == "lds" and dstage == "lds_correlate":
        kname = {(4096, True): "lds_fused4k_kernel", (4096, False): "lds_correlate_kernel", (16384, True): "lds16k_fused_kernel",
                 (16384, False): "lds16k_correlate_kernel"}[(dj["N"], dj["fused_forward"])]
        if dj["N"] == 16384 and "lds_variant=16" not in args.option:
            kname = "r32_fused_kernel" if dj["fused_forward"] else "r32_correlate_kernel"
    if dk["bound"] == "valu":
        achieved = work_launch / (dk["avg_ms"] * 1e-3) / 1e12
        roofline = {"bound": "valu", "kernel": kname, "signal": dj["signal"], "achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / VALU_PEAK_TFLOPS, "avg_kernel_ms": dk["avg_ms"], "useful_flop_per_launch": work_launch,
                    "model": "useful FP32 flop of the rows one launch transforms (5 N log2 N per FFT + 6 N per complex product + 4 N for |.|) "
                             "over the kernel's HIP-event duration, against the FP32 vector peak; the kernel keeps every stage boundary in "
                             "LDS/registers, so HBM is not what bounds it (see traffic and pipeline_equivalent)"}
    else:
        achieved = work_launch / (dk["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname, "signal": dj["signal"], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "avg_kernel_ms": dk["avg_ms"], "alg_bytes_per_launch": work_launch,
                    "model": "bytes this kernel must move through HBM: its side of the split engine's one round trip (8 N per correlation "
                             "row), or x in + X out for the forward stage, over the kernel's HIP-event duration"}
    # The kernel's HIP-event time comes from the profiling pass (one engine per job, events around every stage), `ms_per_step` from the
    # timed region: two loops, so the pass carries its own wall-clock step time -- the kernel time of a step can only be read against
    # THAT -- and the line also prices the dominant kernel against the timed region's step time, a bound that needs no second loop.
    k_ms_step = dk["avg_ms"] * dk["launches_per_step"]
    roofline["profiling_pass"] = {"steps": prof_steps, "ms_per_step": prof_ms_per_step, "dominant_kernel_ms_per_step": k_ms_step,
                                  "kernel_time_within_its_own_step": bool(k_ms_step <= prof_ms_per_step * 1.0005)}
    roofline["frac_floor_from_timed_step"] = roofline["frac"] * k_ms_step / max(k_ms_step, dt / args.steps * 1e3)
    if clocks and clocks.get("sclk_mhz_mean"):
        roofline["clocks"] = clocks
        if dk["bound"] == "valu":      # the FP32 vector peak is 256 CUs x 4 SIMDs x 32 lanes x 2 (packed) x 2 flop x 2.4 GHz
            roofline["frac_at_measured_clock"] = roofline["frac"] * 2400.0 / clocks["sclk_mhz_mean"]
    if dk["bound"] == "hbm":
        # `peak` stays the 8 TB/s of MI355X_MICROARCH.md; what a plain streaming kernel reaches on this part is lower and differs by
        # direction (tools/hbm_bandwidth.hip, profiles/r03_hbm_read_write_copy_bandwidth.log): 16-byte stores 4.2-4.7 TB/s, loads 6.5-7.1
        # measured in this run on this device by a tuned streaming kernel (gacq_stream_probe), not constants from an earlier box
        kind = "read" if dstage == "mag_peak" else ("fill" if dstage == "lds_correlate" else "copy")
        try:
            ceil = stream_ceilings(eng0)
            sc = {"side": kind, "GBps": ceil[kind], "all": ceil, "measured_in_this_run": True,
                  "source": "gacq_stream_probe: 8 x 16-byte non-temporal accesses in flight per lane, 1 GiB per launch, 8 timed launches"}
            if roofline["achieved"] <= ceil[kind]:
                sc["frac"] = roofline["achieved"] / ceil[kind]
            else:            # a kernel faster than the probe: the probe is then no ceiling for it, and no fraction > 1 is printed
                sc["exceeded_by_kernel"] = True
            roofline["stream_ceiling"] = sc
        except Exception as exc:
            roofline["stream_ceiling"] = {"error": repr(exc)[:200]}
    # what the dominant kernel must at least move through HBM per launch (inputs once, outputs once): the yardstick for `traffic`
    # The PMC passes below average over EVERY launch of that kernel in a step, so the yardstick does too: jobs of the step that run the
    # same kernel on the same FFT length (config 4: L5I and B2aD) are pooled, compulsory bytes per step over launches per step.
    def compulsory_step(pj):
        job_ = next(j for j in jobs if j["label"] == pj["signal"])
        k_ = pj["stages"][dstage]
        rec_ = 16.0 * E_total * pj["P"] * pj["D_local"]
        spectra_ = float(S * pj["P"] * pj["N"])
        x_rows_ = float(S) * E_total * pj["F"] * pj["D_local"] * pj["B"] * pj["N"]       # forward spectra [E][F][D][B][N]
        x_samples_ = float(S) * E_total * job_["xs"].shape[1]
        if k_["bound"] == "hbm":
            return k_["work_per_step"] + ((x_rows_ + spectra_) if dstage == "lds_correlate" else (rec_ if dstage == "mag_peak" else 0.0))
        return (x_samples_ if pj["fused_forward"] else x_rows_) + spectra_ + rec_
    pool = [pj for pj in per_job if pj["engine"] == dj["engine"] and pj["N"] == dj["N"] and dstage in pj["stages"]
            and pj["fused_forward"] == dj["fused_forward"]]
    roofline["compulsory_bytes_per_launch"] = sum(compulsory_step(pj) for pj in pool) / sum(pj["stages"][dstage]["launches_per_step"] for pj in pool)
    if len(pool) > 1:
        roofline["compulsory_pooled_over"] = [pj["signal"] for pj in pool]
    # Measured HBM traffic of the dominant kernel: rocprofv3 --pmc wraps a command, so rank 0 of a single-GPU run re-runs this
    # very command line (3 steps, no baselines) under it, one counter group per child run, and reads the kernel's FETCH_SIZE /
    # WRITE_SIZE (KiB; FETCH_SIZE doubled: on gfx950 it counts half of a wide coalesced read -- MI355X_MICROARCH.md, HBM
    # section).  If that cannot be done here (no rocprofv3, child failed) the committed PMC summary of the same command under
    # profiles/ is replayed instead and labelled as such; it is dropped when it describes another kernel, batch or world size.
    roofline["traffic"] = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ)       # never nest profilers
    if world == 1 and rank == 0 and not args.no_pmc and not under_profiler and not os.environ.get("GACQ_BENCH_PMC_CHILD"):
        child = ["--gpus", "1", "--config", str(args.config), "--epochs", str(epochs), "--steps", "3", "--warmup", "1", "--engine", str(args.engine),
                 "--no-cpu-baseline", "--no-latency", "--sustained-s", "0", "--preroll-s", "0", "--no-pmc", "--no-others", "--lanes", "1"]
        for kv in args.option:
            child += ["--option", kv]
        if args.no_self_check:
            child.append("--no-self-check")
        passes = [["FETCH_SIZE"], ["WRITE_SIZE"]]
        if dk["bound"] == "valu":
            passes.append(["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU"])
        t_pmc = time.perf_counter()
        try:
            pm = pmc_passes(child, kname.split("<")[0].split(" ")[0], passes)
            roofline["traffic"] = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
            roofline["traffic_source"] = {"measured_in_this_run": True, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this "
                                          "command (separate passes), bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch",
                                          "fetch_size_kib_raw": pm["FETCH_SIZE"], "write_size_kib": pm["WRITE_SIZE"], "launches_averaged": pm["launches"],
                                          "seconds": None}
            if "SQ_ACTIVE_INST_VALU" in pm and pm.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles over 1024 SIMDs (tools/pmc_profile.py)
                roofline["traffic_source"]["valu_pipe_busy_pmc"] = 4.0 * pm["SQ_ACTIVE_INST_VALU"] / (pm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            roofline["traffic_source"]["seconds"] = time.perf_counter() - t_pmc
        except Exception as exc:
            roofline["traffic_source"] = {"measured_in_this_run": False, "live_error": repr(exc)[:300]}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if roofline["traffic"] is None and os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            if tj.get("config", 2) == args.config and tj.get("kernel_stage") == dstage and tj.get("epochs") == E_total:
                roofline["traffic"] = tj.get("hbm_bytes_per_launch")
                src = roofline.get("traffic_source") or {}
                src.update({"measured_in_this_run": False, "file": "profiles/traffic_latest.json",
                            "from": tj.get("source"), "valu_pipe_busy_pmc": tj.get("valu_pipe_busy")})
                roofline["traffic_source"] = src
        except Exception:
            pass
    # the SURVEY 8d stage-boundary figure, kept as a secondary number: how a perfect HBM-bound five-stage pipeline would
    # have to perform to match the measured step (it exceeds the HBM peak for the fused engines, i.e. it is not a fraction)
    a_pipe_step = sum(a_pipe_bytes(j["sig"].nfft, j["P"], len(j["dop"]), j["B"], j["F"]) for j in jobs) * E_total
    roofline["pipeline_equivalent"] = {"a_pipe_bytes_per_step": a_pipe_step, "GBps": a_pipe_step / (dt / args.steps) / 1e9,
                                       "times_hbm_peak": a_pipe_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS}
    if use_dist:
        mine = torch.tensor([roofline["frac"], dk["avg_ms"]], dtype=torch.float64, device=dev)
        allr = torch.empty(world * 2, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        roofline["per_rank"] = [{"rank": r, "frac": float(allr[2 * r]), "avg_kernel_ms": float(allr[2 * r + 1])} for r in range(world)]

    # host-buffer entry points (H2D + launches + D2H + sync): the drop-in search() call surface; not part of `value`
    latency = None
    if world == 1 and not args.no_latency and args.config == 2:
        job = jobs[0]
        sig, items, ds, ms, dop, B = job["sig"], job["items"], job["ds"], job["ms"], job["dop"], job["B"]
        P, D, N = len(items), len(dop), sig.nfft
        eng.set_stream(None)
        xh = job["host"][0]

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
        # the same call with complex128 samples -- the dtype the reference's search() is handed (np.interp output,
        # acquire-gps-l1.py:94-96) and what the drop-in wrapper passes on to gacq_search64
        xh128 = xh.astype(np.complex128)
        for _ in range(3):
            eng.search_all(sig, xh128, items, ds, ms)
        t_all128 = median_call(lambda: eng.search_all(sig, xh128, items, ds, ms))
        latency = {"search_all_32prn_us": t_all * 1e6, "search_1prn_us": t_one * 1e6, "search_all_32prn_complex128_in_us": t_all128 * 1e6,
                   "cells_per_s_single_epoch_pcie_inclusive": P * D * N / t_all}
        eng.use_torch_stream(dev)
        # host-resident batches streamed through pinned double buffers (H2D of batch i+1 under the kernels of batch i)
        from gnss_dsp_tools_amd import stream
        xs = job["xs"]
        st = stream.EpochStreamer(eng, sig, items, dop, B, E_total, xs.shape[1], depth=3, device=dev)
        nb = 24
        for _ in st.run(xs for _ in range(3)):
            pass
        t1 = time.perf_counter()
        for _ in st.run(xs for _ in range(nb)):
            pass
        t_stream = (time.perf_counter() - t1) / nb
        latency["streamed_batches_pcie_inclusive"] = {"cells_per_s": E_total * P * D * N / t_stream, "ms_per_batch": t_stream * 1e3,
                                                      "epochs_per_batch": E_total, "h2d_bytes_per_batch": int(xs.nbytes)}

    if rank == 0:
        names = "+".join(j["label"] for j in jobs)
        if world == 1 and not exchanged:
            sharding = "none: single rank, whole Doppler grid, no exchange"
        else:
            sharding = ("doppler-slice x%d (every signal's grid cut into contiguous slices, one per rank) + 1 all-gather of 16-byte peak "
                        "records per step (async, overlapped with the next step's kernels) + device-side tie-exact merge" % world)
        out = {
            "metric": "acquisition cells/s (PRN x Doppler x code-phase), " + ("GPS L1 C/A 1 ms" if args.config == 2 else names),
            "value": cells_step * args.steps / dt,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if args.engine == 5 else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %s; %d epoch(s)/step%s batched, inputs resident in HBM"
                                   % (args.config, cfg["label"], epochs, "/GPU" if args.scaling == "weak" else " in total"),
                       "baseline_config": args.config, "signals": [j["label"] for j in jobs],
                       "items": [j["P"] for j in jobs], "doppler_bins": [len(j["dop"]) for j in jobs],
                       "lags": [j["sig"].nfft for j in jobs], "blocks": [j["B"] for j in jobs], "epochs_per_step": E_total,
                       "cells_per_step": cells_step, "cell_blocks_per_step": cell_blocks_step, "sharding": sharding,
                       "shards_seen_by_every_rank": shards_seen,
                       "engine": {0: "auto", 1: "rocfft", 2: "lds-fft", 3: "split", 4: "split-lds", 5: "complex128"}[args.engine],
                       "steps_in_flight": len(lanes), "one_step_in_flight": one_lane, "lane_queues": lane_queue_note},
            "preroll": preroll,
            "sustained": sustained,
            "roofline": roofline,
            "host_call_latency": latency,
            "pipeline": {"a_min_bytes_per_step": sum(a_min_bytes(j["sig"].nfft, j["P"], j["B"], j["xs"].shape[1]) for j in jobs) * E_total,
                         "us_per_search": dt / args.steps / E_total * 1e6, "cell_blocks_per_s": cell_blocks_step * args.steps / dt,
                         "per_signal": per_job},
        }
        if latency:
            # SURVEY 8(d)'s primary figure: wall time of ONE search_all call from host memory (H2D of x + kernels + D2H of the results),
            # median of 60 calls after warm-up
            out["config"]["single_call_us"] = latency["search_all_32prn_us"]
            out["config"]["single_call_cells_per_s"] = latency["cells_per_s_single_epoch_pcie_inclusive"]
            out["config"]["single_call_complex128_in_us"] = latency["search_all_32prn_complex128_in_us"]
        if plain_ratio is not None:
            out["config"]["torchrun_of_one_over_plain_run"] = {"value_ratio": plain_ratio, "within_2_percent": bool(abs(plain_ratio - 1.0) <= 0.02)}
        if use_dist:
            out["config"]["collective_backend"] = dist.get_backend()
            out["config"]["ranks_in_the_process_group"] = dist.get_world_size()
            assert dist.get_world_size() == world and (args.gpus == world or world == 1), ("launched with --gpus %d, the process group has %d ranks" % (args.gpus, dist.get_world_size()))
        # tie-safe peak locations run inside every timed step (tagging in the row reductions, the ambiguity test in the Doppler scan, the
        # re-evaluation launches); the counters say how many pairs of this run needed the complex128 re-evaluation -- the bench's
        # epochs carry strong injected satellites, so usually none (tools/tie_census.py is the noise-only census)
        try:
            eng_t = lanes[0][1]
            before = eng_t.tie_stats()
            run_steps(1)                                           # one more (untimed) step on lane 0: the counters of exactly one step
            torch.cuda.synchronize(dev)
            after = eng_t.tie_stats()
            out["tie_safe"] = dict(after, enabled=bool(eng_t.get_option("tie_safe")), eps=eng_t.get_option("tie_eps_ppb") * 1e-9,
                                   distinct_epochs_per_step=int(sum(len(j["host"]) for j in jobs)),
                                   per_step={k: after[k] - before[k] for k in before})
            if after["kept_fp32"]:
                out["tie_safe"]["warning"] = ("%d ambiguous pairs kept their fp32 answer (re-evaluation list full or unsupported length): raise "
                                              "GACQ_OPT_TIE_CAP" % after["kept_fp32"])
            if out["tie_safe"]["enabled"] and world == 1:
                # what the tie-safe machinery costs: the same steps with it switched off, same process, right after the timed region
                for _, e_l, _ in lanes:
                    e_l.set_option("tie_safe", 0)
                try:
                    _, dt_off = timed(args.steps)
                finally:
                    for _, e_l, _ in lanes:
                        e_l.set_option("tie_safe", 1)
                out["tie_safe"]["ms_per_step_with_tie_safe_off"] = dt_off / args.steps * 1e3
        except Exception as exc:
            out["tie_safe"] = {"error": repr(exc)[:200]}
        if args.config == 2:            # keep the flat stage table of the single-signal line (profiles/)
            out["pipeline"]["stages"] = per_job[0]["stages"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(jobs)
            out["cpu_baseline"]["host"] = host_info()
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if args.config == 2:
                try:
                    out["cpu_baseline_pool"] = cpu_baseline_pool(jobs[0])
                except Exception as exc:                       # the pool leg is informative only
                    out["cpu_baseline_pool"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
    else:
        out = None
    torch.cuda.synchronize(dev)
    for _, e2, _ in lanes:
        e2.close()
    eng.close()
    for q in own_queues:
        q.close()
    return out


if __name__ == "__main__":
    main()
