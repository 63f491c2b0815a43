"""CLI shim + front-end against the reference scripts' own stdout.  tools/make_goldens_cli.py (committed) writes the
synthetic int8 recordings tests/golden/cli_*_int8.iq and runs /root/reference/acquire-<name>.py on them as subprocesses in the
build container; re-running it leaves the fixtures byte-identical."""
import hashlib
import io
import json
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    g = json.load(open(os.path.join(GOLD, "cli_gps_l1.json")))
    path = os.path.join(GOLD, g["file"])
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == g["sha256"]
    g["path"] = path
    return g


def test_frontend_oracle_matches_reference_conditioning(gold):
    from gnss_dsp_tools_amd import frontend, signals
    from oracle import frontend_oracle
    sig = signals.get("gps-l1")
    ms_pad = 2 + 5
    n = int(gold["fs"] * 0.001 * ms_pad)
    with open(gold["path"], "rb") as fp:
        x = frontend_oracle.iq_to_complex(frontend.read_iq_int8(fp, n))
        assert frontend.read_iq_int8(fp, n) is None           # short read -> None (gnsstools/io.py:5-6)
    y = frontend_oracle.condition(x, gold["fs"], gold["coffset"], sig, ms_pad)
    c = gold["conditioned"]
    assert len(y) == c["len"]
    np.testing.assert_allclose(np.c_[y[:16].real, y[:16].imag], c["head"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.c_[y[-16:].real, y[-16:].imag], c["tail"], rtol=1e-9, atol=1e-9)
    assert np.sum(np.abs(y)) == pytest.approx(c["sum_abs"], rel=1e-10)


def test_cli_parser_surface():
    from gnss_dsp_tools_amd import cli, signals
    a = cli.build_parser(signals.get("gps-l1")).parse_args(["f.iq", "69984000", "-9334875"])
    assert (a.items, a.doppler_search, a.time) == ("1-32", "-7000,7000,200", 80)
    g = cli.build_parser(signals.get("glonass-l1")).parse_args(
        cli._join_option_values(["--channel", "-3:3", "--doppler-search", "-7000,7000,500", "f.iq", "16384000", "-250000"]))
    assert (g.items, g.doppler_search, g.carrier_offset) == ("-3:3", "-7000,7000,500", -250000.0)


_LINE = re.compile(r"prn\s+(-?\d+) doppler\s+(-?[\d.]+) metric\s+(-?[\d.]+) code_offset\s+(-?[\d.]+)")


@pytest.mark.gpu
def test_cli_lines_match_reference_stdout(gold):
    from gnss_dsp_tools_amd import cli
    buf = io.StringIO()
    lines = cli.run("gps-l1", gold["argv"] + [gold["path"], str(int(gold["fs"])), str(int(gold["coffset"]))], out=buf)
    assert len(lines) == len(gold["stdout_lines"])
    for mine, ref in zip(lines, gold["stdout_lines"]):
        a, b = _LINE.match(mine).groups(), _LINE.match(ref).groups()
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (mine, ref)          # prn, doppler, code offset: identical text
        assert abs(float(a[2]) - float(b[2])) <= 0.011, (mine, ref)                  # metric printed with 2 decimals
    assert sum(m == r for m, r in zip(lines, gold["stdout_lines"])) >= len(lines) - 1


def test_cli_as_a_program_loads_without_torch():
    """Run as a program (one process per recording, like the reference's scripts) the FFT-search commands need no device tensor
    (gacq_acquire_int8), so the package is loaded without torch -- its import alone would be most of the run's wall time -- and
    refuses to let it in later; the long-code commands keep it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = ("import sys, runpy\nsys.argv = ['cli'] + %r\n"
             "try:\n    runpy.run_module('gnss_dsp_tools_amd.cli', run_name='__main__', alter_sys=True)\nexcept SystemExit:\n    pass\n"
             "from gnss_dsp_tools_amd import _native as nat\nprint('NO_TORCH', nat.NO_TORCH, 'torch' in sys.modules)\n"
             "try:\n    import torch\n    print('torch came in')\nexcept ImportError as e:\n    print('refused:', str(e)[:40])\n")
    out = subprocess.run([sys.executable, "-c", probe % (["--help"],)], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "NO_TORCH True False" in out.stdout and "refused: this process loaded libgacq.so without" in out.stdout, out.stdout[-500:]
    out = subprocess.run([sys.executable, "-c", probe % (["gps-l2cl", "--help"],)], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "NO_TORCH False True" in out.stdout and "torch came in" in out.stdout, out.stdout[-500:]


def test_native_firwin_matches_scipy():
    import scipy.signal
    from gnss_dsp_tools_amd import acquire
    for ntaps, cut in [(161, 1.5e6 / (69.984e6 / 2)), (161, 12e6 / (69.984e6 / 2)), (161, 4e6 / (8.184e6 / 2) * 0.5), (33, 0.3)]:
        np.testing.assert_allclose(acquire.firwin_hann(ntaps, cut), scipy.signal.firwin(ntaps, cut, window='hann'), rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
def test_gpu_frontend_matches_host_frontend(gold):
    """GPU front-end (fp32) vs the numpy front-end oracle that is pinned to the reference: 1e-5 of the signal RMS."""
    import torch
    from gnss_dsp_tools_amd import acquire, frontend, signals
    from oracle import frontend_oracle
    sig = signals.get("gps-l1")
    ms_pad = 2 + 5
    n = int(gold["fs"] * 0.001 * ms_pad)
    raw = open(gold["path"], "rb").read(2 * n)
    with open(gold["path"], "rb") as fp:
        want = frontend_oracle.condition(frontend_oracle.iq_to_complex(frontend.read_iq_int8(fp, n)), gold["fs"], gold["coffset"], sig, ms_pad)
    eng = acquire.Engine(0)
    try:
        got = eng.frontend_dev(sig, np.frombuffer(raw, dtype=np.int8), gold["fs"], gold["coffset"], ms_pad)
        torch.cuda.synchronize()
        got = got.cpu().numpy().astype(np.complex128)
    finally:
        eng.close()
    assert got.shape == want.shape
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    assert np.max(np.abs(got - want)) / rms < 1e-5
    # other rates / cutoffs: 30.69 MS/s family from a 69.984 MS/s-like ratio on random int8 data
    rng = np.random.default_rng(5)
    fs_in, coff = 40.0e6, -1.25e6
    sig2 = signals.get("gps-l5i")
    n2 = int(fs_in * 0.001 * 4)
    iq = rng.integers(-100, 100, size=(n2, 2), dtype=np.int8)
    x = np.empty(n2, dtype=np.complex64)
    x.real, x.imag = iq[:, 0], iq[:, 1]
    want2 = frontend_oracle.condition(x, fs_in, coff, sig2, 4)
    eng = acquire.Engine(0)
    try:
        got2 = eng.frontend_dev(sig2, iq, fs_in, coff, 4)
        torch.cuda.synchronize()
        got2 = got2.cpu().numpy().astype(np.complex128)
    finally:
        eng.close()
    assert np.max(np.abs(got2 - want2)) / np.sqrt(np.mean(np.abs(want2) ** 2)) < 1e-5


_ANY = re.compile(r"(prn|chan)\s+(-?\d+) doppler\s+(-?[\d.]+) metric\s+(-?[\d.]+) code_offset\s+(-?[\d.]+)")


@pytest.mark.gpu
@pytest.mark.parametrize("gname", ["cli_glonass_l1.json", "cli_galileo_e1b.json", "cli_beidou_b1i.json", "cli_gps_l5i.json", "cli_gps_l1cd.json",
                                   "cli_gps_l2cm.json", "cli_xona_x1.json", "cli_glonass_l2.json", "cli_galileo_e6b.json"])
def test_cli_other_signals_match_reference_stdout(gname):
    """--channel list syntax + FDMA bias (acquire-glonass-l1.py:69,28), the BOC/padded 4 ms variant, the padded raw-metric B1I
    search (N = 16384 LDS kernels), the 10.23 Mcps family (N = 61380 radix-31 engine, 30.69 MS/s from a 40 MS/s file), the BOC Weil code of
    L1Cd (N = 81920, ms//10 blocks), the padded 20 ms L2CM search (N = 163840, ms//20 - 1 blocks), Xona X1 (default item list '0', normalised
    metric), GLONASS L2 (437.5 kHz channel spacing) and Galileo E6-B (N = 30690, M = 990 Stockham kernel) through the
    whole device-resident chain: GPU front-end at four different rates / cutoffs, then the engine auto picks."""
    from gnss_dsp_tools_amd import cli
    g = json.load(open(os.path.join(GOLD, gname)))
    path = os.path.join(GOLD, g["file"])
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == g["sha256"]
    lines = cli.run(g["signal"], g["argv"] + [path, str(int(g["fs"])), str(int(g["coffset"]))], out=io.StringIO())
    assert len(lines) == len(g["stdout_lines"])
    for mine, ref in zip(lines, g["stdout_lines"]):
        a, b = _ANY.match(mine).groups(), _ANY.match(ref).groups()
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[4] == b[4], (mine, ref)
        assert abs(float(a[3]) - float(b[3])) <= max(0.11, 2e-5 * float(b[3])), (mine, ref)     # printed with one decimal


@pytest.mark.gpu
@pytest.mark.parametrize("gname", ["cli_gps_l2cl.json", "cli_glonass_l1_p.json", "cli_glonass_l2_p.json"])
def test_cli_longcode_scripts_match_reference_stdout(gname):
    """acquire-gps-l2cl.py / acquire-glonass-l1-p.py / -l2-p.py surface: FILE FS COFFSET ITEM DOPPLER CODE_PHASE -> '%f %f'."""
    from gnss_dsp_tools_amd import cli
    g = json.load(open(os.path.join(GOLD, gname)))
    path = os.path.join(GOLD, g["file"])
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == g["sha256"]
    lines = cli.run_longcode(g["signal"], g["argv"] + [path, str(int(g["fs"])), str(int(g["coffset"]))] + g["tail"], out=io.StringIO())
    a, b = lines[0].split(), g["stdout_lines"][0].split()
    assert a[0] == b[0]                                                  # code phase: identical text
    assert float(a[1]) == pytest.approx(float(b[1]), rel=1e-5)           # metric
