"""CPU: the oracle (oracle/) is held to outputs of the REFERENCE ITSELF (tests/golden/, produced by
tools/make_goldens.py importing /root/reference in the build container).  fp64 vs fp64: 1e-12."""
import hashlib
import os

import numpy as np
import pytest

from conftest import case_iq
from oracle import acq_oracle, codes_oracle

RTOL = 1e-12

FAST_CASES = ["cfg1_gps_l1_prn1", "cfg2_gps_l1_all32", "gps_l1_ms3", "gps_l1_default_grid", "e1b_ms12",
              "l5q_subset", "cfg5_b1i_ms10", "b2i_ms2", "cfg5_glonass_l1", "glonass_l2", "gps_l2cm", "gal_e6b",
              "gal_e5bq", "bds_b3i", "bds_b2bi", "glo_l3ocd", "xona_x1", "xona_x5p", "edge_empty_grid",
              "edge_zero_blocks_l1", "edge_zero_blocks_e1b", "edge_fractional_grid", "cfg3_e1c_subset",
              "bds_b2ap", "bds_b2bq", "gal_e5ai", "gal_e5aq", "gal_e5bi", "gal_e6c", "glo_l3ocp"]
SLOW_CASES = ["cfg3_e1b_subset", "cfg4_l5i_subset", "cfg4_b2ad_b80", "gps_l1cd", "bds_b1cp", "bds_b1cd", "gps_l1cp"]


def _check_case(case):
    x = case_iq(case).astype(np.complex128)
    for item, want in zip(case["items"], case["results"]):
        got = acq_oracle.search_script(case["script"], x, item, case["doppler_search"], case["ms"])
        assert got[2] == want[2], (case["id"], item, "doppler", got, want)
        assert got[1] == pytest.approx(want[1], rel=RTOL, abs=1e-9), (case["id"], item, "code", got, want)
        assert got[0] == pytest.approx(want[0], rel=RTOL), (case["id"], item, "metric", got, want)


@pytest.mark.parametrize("cid", FAST_CASES)
def test_oracle_search_matches_reference(golden_cases, cid):
    _check_case(golden_cases[cid])


@pytest.mark.parametrize("cid", SLOW_CASES)
def test_oracle_search_matches_reference_large(golden_cases, cid):
    _check_case(golden_cases[cid])


def test_every_golden_case_is_covered(golden_cases):
    assert set(golden_cases) == set(FAST_CASES + SLOW_CASES)


def test_oracle_rows_match_reference(golden_cases, golden_rows):
    for key, want in golden_rows.items():
        cid, item, dop = key.split("|")
        case = golden_cases[cid]
        code, fs, n, pad, boc, _norm, _fold, blocks, bias = acq_oracle.VARIANTS[case["script"]]
        x = case_iq(case).astype(np.complex128)
        chips = codes_oracle.chips(code, 0 if bias else int(item))
        q = acq_oracle.search_row(x, chips, float(dop), blocks(case["ms"]), fs=fs, n=n, pad=pad, boc=boc,
                                  bias_hz=bias * int(item))
        np.testing.assert_allclose(q, want, rtol=1e-11, atol=1e-13)


def test_result_tuple_types_match_reference(golden_cases):
    """No winner -> the untouched initial ints (0,0,0) (acquire-gps-l1.py:25,40)."""
    case = golden_cases["edge_empty_grid"]
    x = case_iq(case).astype(np.complex128)
    r = acq_oracle.search_script("gps-l1", x, 1, case["doppler_search"], 1)
    assert r == (0, 0, 0) and all(isinstance(v, int) for v in r)


def test_oracle_chips_match_reference_hashes(golden_chips):
    """Every PRN of every family: SHA-256 of the {0,1} chips equals the reference's."""
    for code, fam in golden_chips.items():
        prns = sorted(fam["prns"], key=int)
        # the pure-Python oracle is slow for the 10230-chip registers: check a spread of PRNs per family
        pick = prns if len(prns) <= 8 else [prns[i] for i in sorted({0, 1, len(prns) // 3, len(prns) // 2, len(prns) - 2, len(prns) - 1})]
        for p in pick:
            c = codes_oracle.chips(code, int(p))
            assert len(c) == fam["code_length"]
            assert hashlib.sha256(c.tobytes()).hexdigest() == fam["prns"][p]["sha256"], (code, p)
            assert "".join(map(str, c[:24])) == fam["prns"][p]["head"]


def test_icd_known_answers(icd_kat):
    """The known-answer vectors the reference carries: L2CM end states (gps/l2cm.py:95-126),
    L5 XB start states (gps/l5i.py:140-143, l5q.py:138-141)."""
    for prn, want in list(icd_kat["l2cm_end_state"].items())[::6]:
        assert codes_oracle.l2cm_end_state(int(prn)) == want, prn
    for prn, want in icd_kat["l5i_xb_start_state"].items():
        assert codes_oracle.l5_xb_start_state("gps_l5i", int(prn)) == want
    for prn, want in icd_kat["l5q_xb_start_state"].items():
        assert codes_oracle.l5_xb_start_state("gps_l5q", int(prn)) == want


def test_gps_ca_first_ten_chips_octal():
    """IS-GPS-200 table 3-I: PRN 1 -> 1440, PRN 2 -> 1620 (gps/ca.py:135-149 prints these to be eyeballed)."""
    def octal10(c):
        return "%o" % int("".join(map(str, c[:10])), 2)
    assert octal10(codes_oracle.chips("gps.ca", 1)) == "1440"
    assert octal10(codes_oracle.chips("gps.ca", 2)) == "1620"
    assert octal10(codes_oracle.chips("gps.ca", 32)) == "1712"


def test_nco_index_vectors(golden_nco):
    for v in golden_nco["vectors"]:
        bias = v.get("bias", 0.0)
        f = -(bias + v["doppler"]) / v["fs"] if bias else -v["doppler"] / v["fs"]
        idx = acq_oracle.nco_indices(f, 0, v["n"])
        assert [int(i) for i in idx[:16]] == v["head"]
        assert hashlib.sha256(idx.astype(np.int32).tobytes()).hexdigest() == v["sha256"]
    t = golden_nco["table"]
    assert hashlib.sha256(acq_oracle.PHASOR_TABLE.tobytes()).hexdigest() == t["sha256_c128"]
    b = golden_nco["boc11"]
    assert hashlib.sha256(acq_oracle.boc11(0, 0, 4092.0 / 32768, 32768).astype(np.int8).tobytes()).hexdigest() == b["n32768_L4092"]
    assert hashlib.sha256(acq_oracle.boc11(0, 0, 10230.0 / 81920, 81920).astype(np.int8).tobytes()).hexdigest() == b["n81920_L10230"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/gnsstools"), reason="reference only exists in the build container")
def test_oracle_against_live_reference():
    """Where the reference is present, run its search() head-to-head on fresh random input."""
    import sys
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    src = open("/root/reference/acquire-beidou-b1i.py").read().split("#\n# main program\n#")[0]
    ns = {}
    exec(compile(src, "acquire-beidou-b1i.py", "exec"), ns)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(4 * 8192) + 1j * rng.standard_normal(4 * 8192)
    want = ns["search"](x, 9, [-600.0, 600.0, 300.0], 2)
    got = acq_oracle.search_script("beidou-b1i", x, 9, [-600.0, 600.0, 300.0], 2)
    assert got[2] == want[2] and got[1] == want[1] and got[0] == pytest.approx(want[0], rel=1e-13)
