"""CPU, world_size 2 and 3 over gloo: the multi-GPU path's exchange + merge logic.

The per-rank compute is stood in by the oracle (tests may use it); what is under test is the Doppler
slicing, the single all-gather of peak records, the shard merge with the reference's tie rule and the
final conversion (gacq_finalize, which needs no GPU)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,name,items,ds,ms,mode", [
    (2, "gps-l1", [3, 4, 28], [-2000.0, 2500.0, 500.0], 1, ""),
    (3, "gps-l1", [3, 9], [1000.0, 2050.0, 150.0], 2, ""),
    (2, "glonass-l1", [-2, 5], [1200.0, 1900.0, 100.0], 1, ""),
    (2, "gps-l1", [3, 9], [1000.0, 2050.0, 150.0], 1, "GLOO_ASYNC"),
    (2, "gps-l1", [3, 9, 4], [1400.0, 1700.0, 100.0], 1, ""),            # 3 bins < 4 per rank -> item split, ragged slices
    (3, "glonass-l1", [-2, 5, 1, 0], [1300.0, 1700.0, 200.0], 1, "GLOO_ASYNC"),
])
def test_sharded_search_equals_unsharded_oracle(tmp_path, world, name, items, ds, ms, mode):
    from gnss_dsp_tools_amd import signals, synth
    from oracle import acq_oracle
    out = tmp_path / "res.json"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if mode:
            env[mode] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out), name,
                                       ",".join(map(str, items)), ",".join(map(str, ds)), str(ms)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, log.decode()[-2000:]
    res = json.load(open(out))
    sig = signals.get(name)
    xs = synth.make_epochs(sig, sig.blocks(ms), 5150, [(items[0], 0.4, 1537.0, 1201)], 2)
    for e in range(xs.shape[0]):
        for it, got in zip(items, res[e]):
            want = acq_oracle.search_script(name, xs[e].astype(np.complex128), it, ds, ms)
            assert got[2] == float(want[2]) and got[1] == float(want[1])
            assert got[0] == pytest.approx(float(want[0]), rel=1e-12)


def test_multi_job_search_uses_one_exchange_and_matches_oracle(tmp_path):
    """search_jobs: two searches sharded over 2 ranks, packed into one all-gather, merged per job."""
    from gnss_dsp_tools_amd import acquire, signals, synth
    from oracle import acq_oracle
    name, items, ds, ms, world = "gps-l1", [3, 9], [1000.0, 2050.0, 150.0], 1, 2
    out = tmp_path / "res.json"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_JOBS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out), name,
                                       ",".join(map(str, items)), ",".join(map(str, ds)), str(ms)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=240)
        assert p.returncode == 0, log.decode()[-2000:]
    res = json.load(open(out))
    sig = signals.get(name)
    xs = synth.make_epochs(sig, sig.blocks(ms), 5150, [(items[0], 0.4, 1537.0, 1201)], 2)
    ds2 = [ds[0] + 37.0, ds[1], ds[2] * 2]
    for e in range(xs.shape[0]):
        x = xs[e].astype(np.complex128)
        want = [acq_oracle.search_script(name, x, it, ds, ms) for it in items] + [acq_oracle.search_script(name, x, items[0], ds2, ms)]
        for got, w in zip(res[e], want):
            assert got[2] == float(w[2]) and got[1] == float(w[1]) and got[0] == pytest.approx(float(w[0]), rel=1e-12)


def test_config5_signal_mix_world2_one_exchange(tmp_path):
    """bench.py --config 5's job mix (GPS L1 + Galileo E1B + BeiDou B1I + GLONASS L1: four FFT lengths, normalised and raw
    metrics, padded / BOC / FDMA variants) over 2 ranks: each grid sliced, one all-gather for all four signals, per-signal
    tie-exact merge == the unsharded oracle."""
    from gnss_dsp_tools_amd import signals, synth
    from oracle import acq_oracle
    spec = [["gps-l1", [3, 28], [-1000.0, 2000.0, 500.0], 2], ["galileo-e1b", [5], [1000.0, 2000.0, 250.0], 10],
            ["beidou-b1i", [6, 33], [1000.0, 2000.0, 250.0], 2], ["glonass-l1", [-7, 3], [1000.0, 2000.0, 250.0], 1]]
    out = tmp_path / "res.json"
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_MIX="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out), json.dumps(spec)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=400)
        assert p.returncode == 0, log.decode()[-2000:]
    res = json.load(open(out))
    assert len(res) == len(spec)
    for (name, items, ds, ms), got in zip(spec, res):
        sig = signals.get(name)
        B = sig.blocks(ms)
        x = synth.make_epochs(sig, B, 6060, [(items[0], 0.4, 1537.0, 1201)], 1, nsamp=sig.samples_needed(B))[0].astype(np.complex128)
        for it, g in zip(items, got):
            w = acq_oracle.search_script(name, x, it, ds, ms)
            assert g[2] == float(w[2]) and g[1] == float(w[1]) and g[0] == pytest.approx(float(w[0]), rel=1e-12), (name, it)


def test_doppler_bounds_cover_grid():
    from gnss_dsp_tools_amd import sharded
    for nd in (0, 1, 5, 40, 70, 200):
        for w in (1, 2, 3, 4, 8):
            b = sharded.doppler_bounds(nd, w)
            assert b[0] == 0 and b[-1] == nd and all(b[i] <= b[i + 1] for i in range(w))
            assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1


def test_job_lanes_need_an_engine_and_cost_order_puts_the_longest_search_first():
    """ShardedSearch.enable_job_lanes is a GPU feature (contexts on streams of their own): the CPU stand-in path refuses it loudly and
    keeps running its jobs serially; the queueing order is by cells x blocks, the cheapest job closing the step."""
    import numpy as np
    import pytest
    from gnss_dsp_tools_amd import sharded, signals
    sh = sharded.ShardedSearch(local_fn=lambda *a: None)
    with pytest.raises(RuntimeError):
        sh.enable_job_lanes(2)
    assert sh._lanes == []
    x1 = np.zeros((1, 8), dtype=np.complex64)
    jobs = [{"name": "gps-l1", "x": x1, "items": list(range(32)), "dopplers": np.arange(200.0), "blocks": 10},
            {"name": "galileo-e1b", "x": x1, "items": list(range(50)), "dopplers": np.arange(200.0), "blocks": 1},
            {"name": "beidou-b1i", "x": x1, "items": list(range(63)), "dopplers": np.arange(200.0), "blocks": 10},
            {"name": "glonass-l1", "x": x1, "items": list(range(15)), "dopplers": np.arange(200.0), "blocks": 10}]
    cost = [sharded.ShardedSearch._job_cost(j) for j in jobs]
    assert sorted(range(4), key=lambda i: -cost[i]) == [2, 1, 3, 0]                 # B1I, E1B, GLONASS, GPS L1
    assert cost[0] == 32 * 200 * signals.get("gps-l1").nfft * 10


def test_merge_tie_rule_lowest_doppler_wins():
    """Equal metrics in two shards: the earlier shard (lower Doppler) must win, like the strict '>' scan."""
    from gnss_dsp_tools_amd import acquire, sharded
    g = np.zeros((3, 1, 2), dtype=acquire.PEAK_DTYPE)
    g["metric"] = [[[5.0, 0.0]], [[5.0, 2.0]], [[7.0, 2.0]]]
    g["idx"] = [[[10, -1]], [[11, 20]], [[12, 21]]]
    g["d_index"] = [[[1, -1]], [[0, 3]], [[2, 0]]]
    merged = sharded.merge_peaks_host(g.view(np.float64).reshape(3, 1, 2, 2), [0, 4, 8]).numpy().view(acquire.PEAK_DTYPE).reshape(2)
    assert (merged[0]["metric"], merged[0]["idx"], merged[0]["d_index"]) == (7.0, 12, 10)
    assert (merged[1]["metric"], merged[1]["idx"], merged[1]["d_index"]) == (2.0, 20, 7)
    # and the native host finalize applies the same rule
    res = acquire.finalize("gps-l1", [1, 2], g.reshape(3, 2), np.arange(12) * 100.0, shard_d0=[0, 4, 8])
    assert res[0] == (7.0, 1023 * (12 / 4096), 1000.0) and res[1] == (2.0, 1023 * (20 / 4096), 700.0)
    none = np.zeros((2, 2), dtype=acquire.PEAK_DTYPE)
    none["d_index"] = -1
    assert acquire.finalize("gps-l1", [1, 2], none, np.arange(12) * 100.0, shard_d0=[0, 6]) == [(0, 0, 0), (0, 0, 0)]


@pytest.mark.parametrize("world,config,extra", [(2, 2, []), (3, 2, ["--scaling", "strong", "--epochs", "2"]), (2, 5, []),
                                                (8, 2, []), (8, 3, []), (8, 4, ["--dry-bins", "5"]), (8, 5, ["--scaling", "strong"])])
def test_bench_driver_style_launch_over_gloo(world, config, extra):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` exactly as the driver launches the scaling runs,
    but with --dry-run-cpu: CPU tensors over gloo, the per-rank compute stood in by the oracle (tests/_gloo_worker.py:oracle_local).
    What runs is bench.py's own N > 1 path -- RANK / LOCAL_RANK / WORLD_SIZE from the environment, Doppler slices per rank, ONE
    all-gather with two steps in flight, merge, shard census, barrier-bracketed MAX-reduced timing, one JSON line from rank 0 --
    so the 8-GPU scaling run does not meet that code for the first time on hardware.  Config 5 = four signals in one exchange; the
    world-8 rows are the driver's own N = 8 launch for every BASELINE configuration: config 3's family as one job per member signal,
    config 4 with fewer Doppler bins (5) than ranks, so that three ranks hold no slice at all and contribute "nothing found" records."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", str(config), "--steps", "2",
           "--warmup", "1", "--dry-run-cpu", os.path.join(ROOT, "tests", "_gloo_worker.py") + ":oracle_local"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["value"] is None and "NOT A MEASUREMENT" in j["note"]
    assert j["n_gpus"] == world and j["steps"] == 2 and j["warmup"] == 1 and j["ms_per_step"] > 0 and j["higher_is_better"] is True
    cfg = j["config"]
    assert cfg["shards_seen_by_every_rank"] == world and cfg["merged_equals_single_rank_scan"] is True
    bins = int(extra[extra.index("--dry-bins") + 1]) if "--dry-bins" in extra else 8
    assert all(sum(b) == bins and len(b) == world for b in cfg["doppler_bins_per_rank"])
    assert len(cfg["signals"]) == {2: 1, 3: 2, 4: 2, 5: 4}[config]
    assert cfg["epochs_per_step"] == ((2 if "--epochs" in extra else 1) if "strong" in extra else world)


def test_deferred_merge_refuses_samples_refilled_in_place():
    """the deferred tie-safe merge re-reads the sample tensor at wait() time, so a caller that refills it in place for
    the next step must get an error, not step-i peaks resolved against step-(i+1) samples.  torch's per-tensor version counter catches it."""
    import torch
    from gnss_dsp_tools_amd import sharded
    x = torch.zeros(4, 8, dtype=torch.complex64)
    stamp = sharded._stamp(x)
    sharded._check_unmodified(stamp)                       # untouched: fine
    pend = sharded.PendingSearch(local=None, gathered=torch.zeros(1), work=None, finish=lambda g: g, stamp=stamp)
    assert pend.wait() is not None
    x.copy_(torch.ones_like(x))                            # the in-place refill
    with pytest.raises(RuntimeError, match="modified in place"):
        sharded.PendingSearch(local=None, gathered=torch.zeros(1), work=None, finish=lambda g: g, stamp=stamp).wait()
    assert sharded._stamp(np.zeros(3)) is None             # non-torch inputs of the CPU tests carry no counter
