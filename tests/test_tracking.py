"""Batched tracking correlators: oracle (sequential restatement) vs the reference's own correlate() outputs (CPU); HIP path
vs both (GPU).  Tolerance 1e-5 of sum|x| for the fp32-input device path (see gacq_tracking.hip on the closed-form phases)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "tracking_cases.json")))["cases"]
IDS = ["%s-%d" % (c["code"], c["prn"]) for c in CASES]


def _x(case):
    rng = np.random.Generator(np.random.PCG64(case["seed"]))
    return (rng.standard_normal(case["n"]) + 1j * rng.standard_normal(case["n"])).astype(np.complex64)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_matches_reference(case):
    from oracle import tracking_oracle
    p = tracking_oracle.correlate(case["code"], _x(case).astype(np.complex128), case["prn"], case["chips"], case["frac"], case["incr"])
    assert p.real == pytest.approx(case["re"], rel=1e-12, abs=1e-10) and p.imag == pytest.approx(case["im"], rel=1e-12, abs=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_gpu_matches_reference(engine, case):
    from gnss_dsp_tools_amd import tracking
    x = _x(case)
    p = tracking.correlate(case["code"], x, case["prn"], case["chips"], case["frac"], case["incr"], engine=engine)
    scale = float(np.sum(np.abs(x)))
    assert abs(p - complex(case["re"], case["im"])) <= 1e-5 * scale / np.sqrt(len(x)) + 1e-9 * scale


@pytest.mark.gpu
def test_gpu_early_prompt_late_batch_equals_oracle(engine):
    """E/P/L of three satellites in one launch (track-gps-l1.py:48-50 shape) vs the sequential oracle."""
    from gnss_dsp_tools_amd import tracking
    from oracle import tracking_oracle
    rng = np.random.default_rng(11)
    n = 4092
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    prns, code_p, cf = [3, 17, 30], [12.3, 1000.9, 511.0], 1.023e6 / 4.092e6
    got = tracking.early_prompt_late("gps.ca", x, prns, code_p, cf, 0.05, engine=engine)
    for a, prn in enumerate(prns):
        for b, off in enumerate((-0.05, 0.0, 0.05)):
            want = tracking_oracle.correlate("gps.ca", x.astype(np.complex128), prn, 0, code_p[a] + off, cf)
            assert abs(got[a, b] - want) < 1e-4
    # a real signal: prompt dominates, early ~ late
    from gnss_dsp_tools_amd import codes
    c = 1.0 - 2.0 * codes.chips("gps.ca", 9)[np.floor((100.25 + cf * np.arange(n)) % 1023).astype(int)]
    y = (0.5 * c + 0.1 * rng.standard_normal(n)).astype(np.complex64)
    e, p, l = tracking.early_prompt_late("gps.ca", y, [9], [100.25], cf, 0.5, engine=engine)[0]
    assert abs(p) > 1.5 * abs(e) and abs(abs(e) - abs(l)) < 0.1 * abs(p)


@pytest.mark.gpu
def test_gpu_device_resident_correlators_equal_the_host_buffer_call(engine):
    """gacq_correlate_batch_dev: the block is already on the GPU (the front-end's output / the samples an acquisition searched), only
    the K correlator specs travel -- written through the PCIe BAR, results watched for in pinned memory.  Byte-identical to
    gacq_correlate_batch for every subcarrier kind, for a 1 ms block (one launch) and a block long enough to need the second kernel,
    repeated calls included (sentinel / buffer reuse)."""
    import torch
    from gnss_dsp_tools_amd import tracking
    rng = np.random.default_rng(23)
    for code, n, prns in (("gps.ca", 4092, [3, 17, 30, 5]), ("galileo.e1b", 8184, [11, 12]), ("gps.l1cp", 20000, [7]), ("gps.l2cm", 5000, [9, 2])):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        xd = torch.from_numpy(x).cuda()
        code_p = rng.uniform(0.0, 1000.0, len(prns))
        for rep in range(3):
            want = tracking.early_prompt_late(code, x, prns, code_p, 0.25, 0.05 + 0.1 * rep, engine=engine)
            got = tracking.early_prompt_late(code, xd, prns, code_p, 0.25, 0.05 + 0.1 * rep, engine=engine)
            assert got.tobytes() == want.tobytes(), (code, rep)
        # the per-block plan of a tracking loop (arrays laid out once, phases written in place): same bits again
        plan = tracking.EplPlan(code, prns, 0.25, engine=engine)
        assert plan(xd, code_p, 0.25).tobytes() == want.tobytes() and plan(x, code_p, np.full(len(prns), 0.25)).tobytes() == want.tobytes()
    with pytest.raises(ValueError):
        tracking.correlate_batch("gps.ca", xd.to(torch.complex128), [1], 0.0, 0.0, 0.25, engine=engine)
