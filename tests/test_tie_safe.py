"""Tie-safe peak locations (gacq_tiesafe.hip, option tie_safe -- on by default).

The reference takes np.argmax and the strict-'>' Doppler scan on fp64 values (acquire-gps-l1.py:34-39).  The fp32 engines tag every
row / (epoch, item) whose runner-up comes within eps of the winner and have the candidate rows re-evaluated in complex128 on the
device; these tests hold (1) the complex128 row kernel itself to the reference's goldens at 1e-10 for every FFT length family,
(2) the resolved locations to the complex128 engine EXACTLY on noise (where every search is a near-tie lottery), (3) the
bookkeeping (list overflow, statistics, Doppler-sliced and multi-device merges)."""
import numpy as np
import pytest

from conftest import case_iq

pytestmark = pytest.mark.gpu

# one case per FFT length / radix set of the complex128 row kernel: 4096 (B = 1, 3), 16384 (B = 10; FDMA biases), 61380 = 4*9*5*11*31
# (B = 1 and the B = 80 of beidou-b2ad), 30690, 65536, 81920 and 163840 (= 2^k * 5)
ROW_KERNEL_CASES = ["cfg1_gps_l1_prn1", "gps_l1_ms3", "cfg5_b1i_ms10", "cfg5_glonass_l1", "cfg4_l5i_subset", "cfg4_b2ad_b80", "gal_e6b",
                    "cfg3_e1b_subset", "gps_l1cd", "gps_l2cm", "xona_x5p", "edge_fractional_grid"]


@pytest.fixture()
def fresh():
    from gnss_dsp_tools_amd import acquire
    eng = acquire.Engine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize("cid", ROW_KERNEL_CASES)
def test_complex128_row_kernel_matches_reference_to_1e_10(fresh, golden_cases, cid):
    """tie_eps_ppb = 10^9 makes every row 'ambiguous': the whole answer then comes from tie_recheck_kernel / tie_resolve_kernel
    (mixed-radix Stockham transforms in complex128) and must agree with the reference's own outputs like engine 5 does -- same
    location, metric within 1e-10 -- with every (item, Doppler) row accounted for in the statistics."""
    from gnss_dsp_tools_amd import acquire
    case = golden_cases[cid]
    x = case_iq(case)
    D = len(acquire.doppler_grid(case["doppler_search"]))
    fresh.set_option("tie_eps_ppb", 1000000000)
    fresh.set_option("tie_cap", max(64, len(case["items"]) * D))
    got = fresh.search_all(case["script"], x, case["items"], case["doppler_search"], case["ms"])
    for g, w, item in zip(got, case["results"], case["items"]):
        assert float(g[2]) == w[2] and float(g[1]) == pytest.approx(w[1], rel=1e-12, abs=1e-9), (cid, item, g, w)
        assert float(g[0]) == pytest.approx(w[0], rel=1e-10), (cid, item, g, w)
    st = fresh.tie_stats()
    assert st["ambiguous_pairs"] == len(case["items"]) and st["rows_reevaluated"] == len(case["items"]) * D and st["kept_fp32"] == 0, st


def _peaks(t):
    from gnss_dsp_tools_amd import acquire
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().view(acquire.PEAK_DTYPE).reshape(t.shape[0], t.shape[1]).copy()


# (signal, items, Doppler search, blocks, epochs, eps of the second pass in ppb)
NOISE_JOBS = [("gps-l1", list(range(1, 33)), [-5000.0, 5000.0, 250.0], 1, 768, 200000),        # the headline shape: fused 4096 kernel
              ("gps-l1", [1, 2, 3, 4, 5, 6], [-2000.0, 2000.0, 250.0], 3, 64, 2000000),        # two-kernel LDS path, B > 1
              ("beidou-b1i", list(range(1, 17)), [-2000.0, 2000.0, 250.0], 2, 8, 20000000),    # N = 16384 correlate kernel
              ("glonass-l1", list(range(-7, 8)), [-2000.0, 2000.0, 250.0], 2, 8, 20000000),    # N = 16384 fused kernel, FDMA biases
              ("gps-l5i", list(range(1, 9)), [-1000.0, 1000.0, 200.0], 1, 4, 20000000),        # engine 3 (Stockham + DFT-31)
              ("galileo-e1b", list(range(1, 9)), [-1000.0, 1000.0, 125.0], 1, 4, 20000000),    # engine 4
              ("galileo-e6b", list(range(1, 5)), [-1000.0, 1000.0, 200.0], 2, 4, 20000000)]    # engine 3, M = 990, B = 2


@pytest.mark.parametrize("name,items,ds,B,E,eps2", NOISE_JOBS, ids=["%s-B%d" % (j[0], j[3]) for j in NOISE_JOBS])
def test_noise_only_locations_equal_the_complex128_engine_exactly(fresh, name, items, ds, B, E, eps2):
    """Noise-only epochs: every search is decided between near-equal candidates, which is where fp32 used to flip a handful of
    locations per 10^5 searches.  With tie-safe locations every peak record must carry the lag and the bin the complex128 engine
    picks -- no tolerance.  A second pass raises eps (2e-4 ... 2e-2) so that a test-sized batch contains dozens of ambiguous pairs
    with several candidate bins each (at the default 8e-6 a batch this small often has none)."""
    import torch
    from gnss_dsp_tools_amd import acquire, signals, synth
    sig = signals.get(name)
    dop = acquire.doppler_grid(ds)
    xs = synth.make_epochs(sig, B, 97531, [], E, nsamp=sig.samples_needed(B))
    xd = torch.from_numpy(xs).cuda()
    fresh.use_torch_stream()
    fresh.set_engine(5)
    ref = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
    fresh.set_engine(0)
    fresh.set_option("tie_cap", 8192)
    for ppb in (8000, eps2):
        fresh.set_option("tie_eps_ppb", ppb)
        before = fresh.tie_stats()
        got = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
        st = fresh.tie_stats()
        np.testing.assert_array_equal(got["idx"], ref["idx"], err_msg="%s eps %d ppb" % (name, ppb))
        np.testing.assert_array_equal(got["d_index"], ref["d_index"])
        np.testing.assert_allclose(got["metric"], ref["metric"], rtol=2e-6)
        assert st["kept_fp32"] == 0, st
        if ppb == eps2:
            assert st["ambiguous_pairs"] > before["ambiguous_pairs"], (st, before)
    # the statistics say how many pairs there were; switched off, the same batch goes through untouched
    fresh.set_option("tie_safe", 0)
    off = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
    assert fresh.tie_stats() == st
    np.testing.assert_allclose(off["metric"], ref["metric"], rtol=2e-6)


def test_full_list_keeps_the_fp32_answer_and_counts_it(fresh):
    """A re-evaluation list that is too small for the launch is dropped whole (which pairs would have found room depends on the arrival
    order of the listing atomics; the result must not): every ambiguous pair keeps its fp32 record -- bit for bit the tie_safe = 0
    answer, run after run -- all of them are counted in kept_fp32, and the next launch starts from an empty list again."""
    import torch
    from gnss_dsp_tools_amd import acquire, signals, synth
    sig = signals.get("gps-l1")
    items = list(range(1, 33))
    dop = acquire.doppler_grid([-5000.0, 5000.0, 250.0])
    xs = synth.make_epochs(sig, 1, 2468, [], 16, nsamp=4096)
    xd = torch.from_numpy(xs).cuda()
    fresh.use_torch_stream()
    fresh.set_option("tie_safe", 0)
    plain = _peaks(fresh.search_batch_dev(sig, xd, items, dop, 1))
    fresh.set_option("tie_safe", 1)
    fresh.set_option("tie_eps_ppb", 1000000000)       # every pair ambiguous, 40 rows each
    fresh.set_option("tie_cap", 100)                  # room for two pairs of 512
    for launch in (1, 2, 3):
        got = _peaks(fresh.search_batch_dev(sig, xd, items, dop, 1))
        st = fresh.tie_stats()
        assert st["rows_reevaluated"] == 0 and st["locations_changed"] == 0 and st["kept_fp32"] == launch * 16 * 32, st
        for k in ("idx", "d_index", "metric"):
            assert (got[k] == plain[k]).all(), k
    # the host-buffer call reports it: GACQ_WARN_TIE_LIST_FULL -> a TieListFull warning, results still the fp32 ones
    from gnss_dsp_tools_amd import _native as nat
    with pytest.warns(nat.TieListFull):
        host = fresh.search_blocks(sig, xs[0], items, dop, 1)
    assert len(host) == len(items)
    assert fresh.tie_stats()["kept_fp32"] == 3 * 16 * 32 + 32
    import warnings
    fresh.set_option("tie_eps_ppb", 8000)
    with warnings.catch_warnings():
        warnings.simplefilter("error", nat.TieListFull)      # an ordinary call afterwards carries no stale warning
        fresh.search_blocks(sig, xs[0], items, dop, 1)
    fresh.set_option("tie_eps_ppb", 1000000000)
    # ... and a list that fits is used again right away
    rows_before = fresh.tie_stats()["rows_reevaluated"]
    fresh.set_option("tie_cap", 16 * 32 * 40)
    _peaks(fresh.search_batch_dev(sig, xd, items, dop, 1))
    st = fresh.tie_stats()
    assert st["rows_reevaluated"] - rows_before == 16 * 32 * 40 and st["kept_fp32"] == 3 * 16 * 32 + 32, st


@pytest.mark.parametrize("name,items,ds,B,E", [("gps-l1", list(range(1, 13)), [-3000.0, 3000.0, 250.0], 2, 6),
                                               ("beidou-b1i", [1, 2, 3, 4, 5], [-2000.0, 2000.0, 250.0], 2, 3),
                                               ("gps-l5i", [1, 2, 3], [-1000.0, 1000.0, 200.0], 1, 2)])
def test_doppler_slices_and_device_groups_merge_to_the_unsliced_records(fresh, name, items, ds, B, E):
    """Near-ties ACROSS slices of the Doppler grid: a workspace limit that cuts one epoch's grid into slices (merged on the device)
    and a three-member device group (merged on the host, re-evaluated on a member) must give the records of the unsliced tie-safe
    search byte for byte.  eps = 2e-2 makes nearly every pair of slice winners a 'near-tie'."""
    import torch
    from gnss_dsp_tools_amd import acquire, signals, synth
    sig = signals.get(name)
    dop = acquire.doppler_grid(ds)
    xs = synth.make_epochs(sig, B, 1122, [], E, nsamp=sig.samples_needed(B))
    xd = torch.from_numpy(xs).cuda()
    fresh.use_torch_stream()
    fresh.set_option("tie_eps_ppb", 20000000)
    fresh.set_option("tie_cap", 8192)
    whole = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
    assert fresh.tie_stats()["ambiguous_pairs"] > 0
    sliced = acquire.Engine(0, workspace_bytes=max(1 << 20, 5 * 8 * sig.nfft * B))      # ~5 bins of forward spectra per slice
    try:
        sliced.use_torch_stream()
        sliced.set_option("tie_eps_ppb", 20000000)
        sliced.set_option("tie_cap", 8192)
        sliced.set_option("fused_4k", 0)               # the fused kernels have no forward-spectra buffer to slice
        sliced.set_option("fused_16k", 0)
        got = _peaks(sliced.search_batch_dev(sig, xd, items, dop, B))
        assert got.tobytes() == whole.tobytes()
    finally:
        sliced.close()
    grp = acquire.DeviceGroup([0, 0, 0])
    try:
        for k in range(3):
            from gnss_dsp_tools_amd import _native as nat
            import ctypes
            nat.check(nat.lib.gacq_set_option(ctypes.c_void_p(nat.lib.gacq_group_member(grp._g, k)), nat.OPTIONS["tie_eps_ppb"], 20000000))
            nat.check(nat.lib.gacq_set_option(ctypes.c_void_p(nat.lib.gacq_group_member(grp._g, k)), nat.OPTIONS["tie_cap"], 8192))
        res = grp.search_batch_host(sig, xs, items, dop, B)
    finally:
        grp.close()
    want = [acquire.finalize(sig, items, whole[e], dop) for e in range(E)]
    assert res == want


def test_reevaluation_buffers_survive_changing_capacities(fresh):
    """One context, searches whose re-evaluation lists have different capacities and block counts in turn (the scratch rows, per-block
    magnitude rows and arrival counters are laid out per call): every row re-evaluated (eps = 1) on the split-LDS kernel (N = 16384, 65536),
    the LDS kernel (4096, B > 1) and the generic kernel (61380), small call, larger call, small call again -- always engine 5's answer."""
    import torch
    from gnss_dsp_tools_amd import acquire, signals, synth
    fresh.use_torch_stream()
    fresh.set_option("tie_eps_ppb", 1000000000)
    plan = [("beidou-b1i", [1, 2], [-500.0, 500.0, 250.0], 2, 1), ("beidou-b1i", list(range(1, 9)), [-1000.0, 1000.0, 250.0], 3, 2),
            ("galileo-e1b", [4, 5, 6], [0.0, 500.0, 125.0], 1, 1), ("beidou-b1i", [1, 2], [-500.0, 500.0, 250.0], 2, 1),
            ("gps-l1", [1, 2, 3], [-1000.0, 1000.0, 250.0], 3, 2), ("gps-l1", list(range(1, 17)), [-1000.0, 1000.0, 250.0], 2, 3),
            ("gps-l5i", [1, 2], [0.0, 400.0, 200.0], 1, 1), ("glonass-l1", [-1, 0, 3], [-500.0, 500.0, 250.0], 2, 2)]
    for name, items, ds, B, E in plan:
        sig = signals.get(name)
        dop = acquire.doppler_grid(ds)
        xs = synth.make_epochs(sig, B, 606, synth.default_sats(items), E, nsamp=sig.samples_needed(B))
        xd = torch.from_numpy(xs).cuda()
        fresh.set_option("tie_cap", E * len(items) * len(dop))
        fresh.set_engine(0)
        got = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
        fresh.set_engine(5)
        ref = _peaks(fresh.search_batch_dev(sig, xd, items, dop, B))
        np.testing.assert_array_equal(got["idx"], ref["idx"], err_msg=name)
        np.testing.assert_array_equal(got["d_index"], ref["d_index"])
        np.testing.assert_allclose(got["metric"], ref["metric"], rtol=1e-10)
    assert fresh.tie_stats()["kept_fp32"] == 0


def test_option_ranges_are_validated(fresh):
    """gacq_set_option rejects values outside an option's range instead of storing them."""
    from gnss_dsp_tools_amd import _native as nat
    for opt, bad in (("lds_pch", -1), ("lds_pch", 100000), ("lds_ugroup", -3), ("split_teams", 5), ("split_mfma", 2), ("fused_inner", 2), ("tie_eps_ppb", -5), ("fused_4k", 7),
                     ("tie_cap", 1 << 30)):
        with pytest.raises(nat.GacqError):
            fresh.set_option(opt, bad)
    fresh.set_option("lds_pch", 8)
    assert fresh.get_option("lds_pch") == 8


@pytest.mark.parametrize("name,items,ds,ms,seed", [
    ("gps-l1", [2, 9, 13, 30], [-3000.0, 3000.0, 250.0], 1, 801),
    ("beidou-b1i", [1, 20], [-1000.0, 1000.0, 250.0], 2, 802),
    ("gps-l5i", [7, 8], [-400.0, 400.0, 200.0], 1, 803),
    ("galileo-e1b", [3], [1000.0, 1500.0, 125.0], 8, 804),
])
def test_complex128_samples_are_searched_unrounded(engine, name, items, ds, ms, seed):
    """The reference's search() is handed complex128 samples (np.interp's output, acquire-gps-l1.py:94-96).  gacq_search64: engine 5
    reads them as given -- 1e-10 against the numpy oracle on the UNROUNDED samples, and measurably different from what it makes of
    their complex64 rounding; the fp32 engines locate like the oracle on the unrounded samples (their near-ties are re-evaluated from
    them) with metrics inside the 1e-5 bar."""
    from gnss_dsp_tools_amd import signals, synth
    from oracle import acq_oracle
    sig = signals.get(name)
    x128 = synth.make_iq(sig, sig.blocks(ms), seed, [(items[0], 0.3, ds[0] + 0.4 * (ds[1] - ds[0]), 1234)], dtype=np.complex128)
    assert x128.dtype == np.complex128 and np.any(x128 != x128.astype(np.complex64))
    want = [acq_oracle.search_script(name, x128, it, ds, ms) for it in items]
    engine.set_engine(5)
    try:
        got5 = engine.search_all(sig, x128, items, ds, ms)
        got5_rounded = engine.search_all(sig, x128.astype(np.complex64), items, ds, ms)
    finally:
        engine.set_engine(0)
    for g, w in zip(got5, want):
        assert float(g[1]) == float(w[1]) and float(g[2]) == float(w[2])
        assert float(g[0]) == pytest.approx(float(w[0]), rel=1e-10)
    assert max(abs(float(a[0]) - float(b[0])) / float(b[0]) for a, b in zip(got5_rounded, want)) > 1e-9      # the rounding is visible at this level
    got = engine.search_all(sig, x128, items, ds, ms)
    for g, w in zip(got, want):
        assert float(g[1]) == float(w[1]) and float(g[2]) == float(w[2])
        assert float(g[0]) == pytest.approx(float(w[0]), rel=1e-5)


def test_complex128_device_batches_locate_like_the_complex128_engine_on_noise(engine):
    """Noise-only epochs (every search a near-tie lottery) as complex128 device tensors: the default engine's locations must be
    those of engine 5 on the same unrounded samples, batch entry point and shard merge included."""
    import torch
    from gnss_dsp_tools_amd import acquire, signals, synth
    sig = signals.get("gps-l1")
    E, items = 96, list(range(1, 33))
    dop = acquire.doppler_grid([-5000.0, 5000.0, 250.0])
    xs = np.stack([synth.make_iq(sig, 1, 9000 + e, [], nsamp=sig.n, dtype=np.complex128) for e in range(E)])
    xd = torch.from_numpy(xs).cuda()
    engine.use_torch_stream()
    got = engine.search_batch_dev(sig, xd, items, dop, 1).cpu().numpy().view(acquire.PEAK_DTYPE).reshape(E, -1)
    engine.set_engine(5)
    try:
        ref = engine.search_batch_dev(sig, xd, items, dop, 1).cpu().numpy().view(acquire.PEAK_DTYPE).reshape(E, -1)
    finally:
        engine.set_engine(0)
    assert np.array_equal(got["idx"], ref["idx"]) and np.array_equal(got["d_index"], ref["d_index"])
    assert np.max(np.abs(got["metric"] - ref["metric"]) / ref["metric"]) < 1e-5
    # two Doppler slices merged with the tie-safe merge on the complex128 samples
    half = len(dop) // 2
    a = engine.search_batch_dev(sig, xd, items, dop[:half], 1)
    b = engine.search_batch_dev(sig, xd, items, dop[half:], 1)
    merged = engine.merge_peaks_tiesafe_dev(sig, xd, items, dop, 1, torch.stack([a, b]), [0, half]).cpu().numpy().view(acquire.PEAK_DTYPE).reshape(E, -1)
    assert np.array_equal(merged["idx"], ref["idx"]) and np.array_equal(merged["d_index"], ref["d_index"])
