"""CPU tests of bench.py's host logic: the live-traffic probe (rocprofv3 --pmc child runs) against a stand-in profiler, and the
stage model / byte formulas the roofline block is computed from."""
import importlib.util
import os
import stat
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


FAKE = textwrap.dedent('''\
    #!%s
    # stand-in for rocprofv3: writes a counter_collection.csv like the real tool (one row per counter, dispatch and XCD instance)
    import os, sys
    a = sys.argv[1:]
    cut = a.index("--")
    opts = a[:cut]
    counters = []
    i = opts.index("--pmc") + 1
    while i < len(opts) and not opts[i].startswith("-"):
        counters.append(opts[i]); i += 1
    d = opts[opts.index("-d") + 1]
    if os.environ.get("FAKE_ROCPROF_FAIL"):
        sys.stderr.write("boom\\n"); sys.exit(3)
    assert os.environ.get("GACQ_BENCH_PMC_CHILD") == "1" and "RANK" not in os.environ
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    with open(os.path.join(d, "host", "p_counter_collection.csv"), "w") as f:
        f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
        for disp in (1, 2, 3):
            for c in counters:
                for xcd in range(8):                      # per-XCD instances of one dispatch are summed
                    f.write('%%d,"void (anonymous namespace)::lds_fused4k_kernel<4, true>(float2 const*)",%%s,%%d\\n' %% (disp, c, 100 * disp))
                f.write('%%d,"(anonymous namespace)::best_doppler_kernel(RowRec const*)",%%s,7\\n' %% (disp, c))
    ''') % sys.executable


def test_pmc_passes_reads_one_counter_group_per_child_run(bench, tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr("shutil.which", lambda name: str(exe) if name == "rocprofv3" else None)
    monkeypatch.setenv("RANK", "0")                       # a torchrun parent's rendezvous variables must not reach the child
    got = bench.pmc_passes(["--steps", "3"], "lds_fused4k_kernel", [["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU"]])
    # dispatches 1, 2, 3 -> 8 instances x 100 x id each; the average per launch is 8 * 100 * 2
    assert got["FETCH_SIZE"] == 1600.0 and got["WRITE_SIZE"] == 1600.0 and got["SQ_ACTIVE_INST_VALU"] == 1600.0
    assert got["launches"] == 3
    with pytest.raises(RuntimeError, match="no FETCH_SIZE rows"):
        bench.pmc_passes([], "some_other_kernel", [["FETCH_SIZE"]])


def test_pmc_passes_reports_a_failing_profiler(bench, tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr("shutil.which", lambda name: str(exe) if name == "rocprofv3" else None)
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "1")
    with pytest.raises(RuntimeError, match="rc 3"):
        bench.pmc_passes([], "lds_fused4k_kernel", [["FETCH_SIZE"]])


def test_stage_boundary_byte_model_matches_survey_8d(bench):
    # config 2 of SURVEY.md 8d: P = 32, D = 40, B = 1, N = 4096 -> A_pipe = 216.0 MB; config 1: 5.9 MB
    assert bench.a_pipe_bytes(4096, 32, 40, 1) == 8 * 4096 * (4 * 40 + 32 + 5 * 32 * 40)
    assert round(bench.a_pipe_bytes(4096, 32, 40, 1) / 1e6, 1) == 216.0
    assert round(bench.a_pipe_bytes(4096, 1, 20, 1) / 1e6, 1) == 5.9
    # B > 1 adds the fp32 q read + write per extra block; GLONASS has one forward set per channel
    assert bench.a_pipe_bytes(16384, 15, 200, 10, F=15) - bench.a_pipe_bytes(16384, 15, 200, 10, F=1) == 8 * 16384 * 4 * 200 * 10 * 14
    assert bench.engine_kind(4096) == "lds" and bench.engine_kind(61380) == "split31" and bench.engine_kind(65536) == "split_lds"
