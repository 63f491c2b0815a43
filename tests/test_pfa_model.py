"""Host-side check of the prime-factor engine's index algebra (CPU, no GPU): tools/model_pfa.py replays the four kernels of
gacq_pfa.hip -- rotated gather + DFT-31, the in-place 11 x Nb x 9 passes with their own LDS address expressions and the pass-b lane
table, the inverse DFT-31 with its lag labels -- in numpy.  The chain must equal ifft(fft(c) conj(fft(x))) in natural lag order for both
shapes (N = 61380, 30690), and no LDS access may put two lanes of a 16-lane store group / 32-lane load group on one bank slot."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pfa_index_maps_compute_the_correlation_and_are_bank_conflict_free():
    spec = importlib.util.spec_from_file_location("model_pfa", os.path.join(ROOT, "tools", "model_pfa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    errs, conf = m.run()
    assert set(errs) == {1980, 990}
    assert all(e < 1e-12 for e in errs.values()), errs
    for M, c in conf.items():
        assert len(c) == 7 and all(v == 0 for v in c.values()), (M, c)
