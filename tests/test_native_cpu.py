"""CPU: host-side parts of libgacq.so -- symbol export, PRN generators (bit-exact), replica sampler,
error paths that need no GPU -- and the host logic of the Python mirror."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from gnss_dsp_tools_amd import _native as nat
from gnss_dsp_tools_amd import acquire, codes, signals
from oracle import acq_oracle, codes_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/gacq.h is exported by the built .so and bound in _native.py."""
    hdr = open(os.path.join(ROOT, "include", "gacq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gacq_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(nat.SYMBOLS), declared ^ set(nat.SYMBOLS)
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    assert ctypes.sizeof(nat.Peak) == 16
    assert ctypes.sizeof(nat.Result) == 32
    assert ctypes.sizeof(nat.SigDesc) == 32
    assert acquire.PEAK_DTYPE.itemsize == 16


def test_native_chips_bit_exact_all_prns(golden_chips):
    """SHA-256 of every PRN of every code family equals the reference's."""
    assert sorted(golden_chips) == sorted(codes.names())
    total = 0
    for code, fam in golden_chips.items():
        assert codes.code_length(code) == fam["code_length"]
        assert codes.chip_rate(code) == fam["chip_rate"]
        assert sorted(map(str, codes.prns(code)), key=int) == sorted(fam["prns"], key=int)
        for prn, g in fam["prns"].items():
            c = codes.chips(code, int(prn))
            assert hashlib.sha256(c.tobytes()).hexdigest() == g["sha256"], (code, prn)
            assert "".join(map(str, c[-24:])) == g["tail"]
            total += 1
    assert total == 2239 + 115 + 1        # + L2CL PRNs + the GLONASS P code


def test_native_chips_equal_oracle_chips():
    for code, prn in [("gps.ca", 17), ("gps.l5i", 101), ("galileo.e1b", 50), ("beidou.b1i", 40), ("beidou.b2ad", 63),
                      ("glonass.ca", 0), ("gps.l1cd", 210), ("beidou.b1cp", 1), ("glonass.l3ocp", 5), ("xona.x5p", 0)]:
        np.testing.assert_array_equal(codes.chips(code, prn), codes_oracle.chips(code, prn))


@pytest.mark.parametrize("code,n,boc", [("gps.ca", 4096, False), ("galileo.e1b", 32768, True), ("gps.l5i", 30690, False),
                                         ("beidou.b1i", 8192, False), ("glonass.ca", 16384, False), ("gps.l1cd", 81920, True),
                                         ("galileo.e6b", 15345, False), ("gps.l2cm", 81920, False)])
def test_replica_sampler_matches_oracle(code, n, boc):
    prn = codes.prns(code)[3 % len(codes.prns(code))]
    chips = codes_oracle.chips(code, prn)
    want = acq_oracle.sample_code(chips, 0, 0, float(len(chips)) / n, n)
    if boc:
        want = want * acq_oracle.boc11(0, 0, float(len(chips)) / n, n)
    got = codes.replica(code, prn, n, boc)
    np.testing.assert_array_equal(got.astype(np.float64), want)


def test_code_errors():
    assert nat.lib.gacq_code_length(b"gps.nope") == -2
    buf = (ctypes.c_uint8 * 2048)()
    assert nat.lib.gacq_code_chips(b"gps.ca", 999, buf, 2048) == -3
    assert nat.lib.gacq_code_chips(b"gps.ca", 1, buf, 10) == -1
    with pytest.raises(nat.GacqError):
        codes.chips("galileo.e1b", 51)


def test_create_without_gpu_fails_loudly():
    if nat.lib.gacq_device_count() > 0:
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert nat.lib.gacq_create(0, ctypes.byref(h)) == -7
    assert b"no HIP device" in nat.lib.gacq_last_error(None)
    with pytest.raises(nat.GacqError):
        acquire.Engine(0)


def test_signal_table_agrees_with_oracle_variants():
    assert set(signals.SIGNALS) == set(acq_oracle.VARIANTS)
    for name, s in signals.SIGNALS.items():
        code, fs, n, pad, boc, norm, fold, blocks, bias = acq_oracle.VARIANTS[name]
        assert (s.code, s.fs, s.n, s.pad, s.boc, s.normalised, s.fold, s.bias_hz) == (code, fs, n, pad, boc, norm, fold, bias), name
        for ms in (0, 1, 4, 8, 10, 20, 39, 40, 80):
            assert s.blocks(ms) == blocks(ms), (name, ms)


def test_result_line_formats(golden_cases):
    for case in golden_cases.values():
        for item, res, line in zip(case["items"], case["results"], case["lines"]):
            m, c, d = res
            assert acquire.format_result(case["script"], item, (m, c, d)) == line


def test_option_parsers():
    assert acquire.parse_list_ranges("1,3,7-9") == [1, 3, 7, 8, 9]
    assert acquire.parse_list_ranges("-7:7", sep=":") == list(range(-7, 8))
    assert acquire.parse_list_floats("-7000,7000,200") == [-7000.0, 7000.0, 200.0]
    np.testing.assert_array_equal(acquire.doppler_grid([-5000, 5000, 500]), np.arange(-5000, 5000, 500))
    assert len(acquire.doppler_grid([-5000.0, 5000.0, 250.0])) == 40      # half-open: max excluded


def test_finalize_merges_shards_in_doppler_order_with_strict_greater():
    """gacq_finalize needs no GPU: shard merge + (metric, code, doppler) conversion."""
    # a gacq_sig cannot be built without a GPU, so this pins the merge RULE the native function
    # implements (tests/test_gpu_parity.py::test_finalize_shard_merge calls the native one)
    peaks = np.zeros((3, 2), dtype=acquire.PEAK_DTYPE)
    peaks["metric"] = [[5.0, 0.0], [5.0, 2.0], [7.0, 2.0]]
    peaks["idx"] = [[10, -1], [11, 20], [12, 21]]
    peaks["d_index"] = [[1, -1], [0, 3], [2, 0]]
    d0 = [0, 4, 8]
    best = []
    for p in range(2):
        m, i, d = 0.0, -1, -1
        for s in range(3):
            if peaks[s, p]["d_index"] >= 0 and peaks[s, p]["metric"] > m:
                m, i, d = peaks[s, p]["metric"], peaks[s, p]["idx"], peaks[s, p]["d_index"] + d0[s]
        best.append((m, i, d))
    assert best == [(7.0, 12, 10), (2.0, 20, 7)]       # ties keep the earlier (lower-Doppler) shard


def test_header_compiles_as_strict_c99_and_a_c_program_links(tmp_path):
    """include/gacq.h is the drop-in boundary for C callers too: a C99 translation unit (-pedantic, no C++) includes it,
    links libgacq.so and gets the ICD's first ten C/A chips of PRN 1 (octal 1440) and a loud error without a GPU."""
    import subprocess
    src = tmp_path / "cabi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdint.h>
#include "gacq.h"
int main(void) {
  uint8_t chips[1023];
  gacq_ctx* ctx = 0;
  int i, rc;
  if (gacq_code_chips("gps.ca", 1, chips, 1023) < 0) return 1;
  for (i = 0; i < 10; i++) putchar('0' + chips[i]);
  rc = gacq_create(1 << 20, &ctx);                 /* no such device anywhere: must fail, never crash */
  printf(" %d %d\n", rc, ctx == 0);
  return 0;
}
''')
    exe = tmp_path / "cabi"
    libdir = os.path.join(ROOT, "gnss-dsp-tools_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lgacq", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-2000:]
    chips, rc, null = run.stdout.split()
    assert chips == "1100100000" and int(rc) < 0 and null == "1"


def test_new_entry_points_fail_loudly_without_crashing():
    """Round-2 entry points: NULL handles come back as GACQ_ERR_BAD_ARG, a device group cannot be created without a GPU
    (GACQ_ERR_NO_DEVICE, like gacq_create), option numbers are checked -- all without touching a device."""
    lib = nat.lib
    res = (nat.Result * 2)()
    x = np.zeros(8, dtype=np.complex64)
    items = np.zeros(2, dtype=np.int32)
    dop = np.array([0.0, 250.0])
    args = (x.ctypes.data_as(nat.c_float_p), 4, 1, items.ctypes.data_as(nat.c_int_p), 2, dop.ctypes.data_as(nat.c_double_p), 2, None, 1, res)
    assert lib.gacq_search_batch(None, *args) == -1
    assert lib.gacq_group_search_batch(None, *args) == -1
    assert lib.gacq_group_size(None) == -1 and lib.gacq_group_member(None, 0) is None
    assert lib.gacq_group_set_exchange(None, 1) == -1            # round 5: RCCL exchange of a device group, complex128 samples in
    x128 = np.zeros(8, dtype=np.complex128)
    assert lib.gacq_search64(None, x128.ctypes.data_as(ctypes.c_void_p), 4, items.ctypes.data_as(nat.c_int_p), 2, dop.ctypes.data_as(ctypes.c_void_p), 2, None, 1, res) == -1
    assert lib.gacq_search_batch_dev64(None, None, 4, 1, items.ctypes.data_as(nat.c_int_p), 2, dop.ctypes.data_as(nat.c_double_p), 2, None, 1, None) == -1
    assert lib.gacq_merge_peaks_tiesafe_dev64(None, None, 4, 1, items.ctypes.data_as(nat.c_int_p), 2, dop.ctypes.data_as(nat.c_double_p), 2, None, 1, None, 1,
                                              items.ctypes.data_as(nat.c_int_p), None) == -1
    h = ctypes.c_void_p()
    assert lib.gacq_group_create(None, 1, ctypes.byref(h)) == -1
    ids = (ctypes.c_int * 2)(0, 1)
    if lib.gacq_device_count() == 0:
        assert lib.gacq_group_create(ids, 2, ctypes.byref(h)) == -7 and not h.value
        with pytest.raises(nat.GacqError):
            acquire.DeviceGroup([0])
    assert lib.gacq_set_option(None, 0, 1) == -1 and lib.gacq_get_option(None, 0, None) == -1
    assert lib.gacq_debug_nco_indices(None, 1, 0.0, 0.0, None) == -1
    assert lib.gacq_longcode_search_int8(None, None, 0, 1.0, 0.0, b"gps.l2cl", 1, 0.0, None, 1, 1, 1, None) == -1
    assert lib.gacq_debug_fft_plans(None) == -1
    assert lib.gacq_acquire_int8(None, None, 0, 1.0, 0.0, None, 0, 0, None, 0, None, 0, None, 0, None) == -1
    assert set(nat.OPTIONS.values()) == set(range(len(nat.OPTIONS)))          # GACQ_OPT_* numbering is dense
    # round 6: streams with a hardware queue of their own / a CU mask
    sh, mask = ctypes.c_void_p(), (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
    assert lib.gacq_stream_create_cu_mask(0, mask, 8, None) == -1
    assert lib.gacq_stream_create_cu_mask(-1, mask, 8, ctypes.byref(sh)) == -1 and not sh.value
    assert lib.gacq_stream_create_cu_mask(0, None, 8, ctypes.byref(sh)) == -1 and lib.gacq_stream_create_cu_mask(0, mask, 0, ctypes.byref(sh)) == -1
    if lib.gacq_device_count() > 0:
        assert lib.gacq_stream_create_cu_mask(0, (ctypes.c_uint32 * 8)(), 8, ctypes.byref(sh)) == -1 and not sh.value      # empty mask
    assert lib.gacq_stream_destroy(0, None) == 0
    assert lib.gacq_cu_census(None, 16, None) == -1
    hdr = open(os.path.join(ROOT, "include", "gacq.h")).read()
    for name, num in nat.OPTIONS.items():
        assert re.search(r"#define GACQ_OPT_%s %d\b" % (name.upper(), num), hdr), name
    assert re.search(r"#define GACQ_NOPTS %d\b" % len(nat.OPTIONS), hdr)
    # the one positive return code (a warning, not an error) and its Python mirror
    assert re.search(r"#define GACQ_WARN_TIE_LIST_FULL %d\b" % nat.WARN_TIE_LIST_FULL, hdr) and nat.WARN_TIE_LIST_FULL > 0
    assert nat.check_search(0) == 0
    with pytest.warns(nat.TieListFull):
        assert nat.check_search(nat.WARN_TIE_LIST_FULL) == nat.WARN_TIE_LIST_FULL
    with pytest.raises(nat.GacqError):
        nat.check_search(-1)
