"""Host-side check of the N = 16384 engine's index algebra (CPU, no GPU): tools/model_fft16k.py replays the data movement of
fft16k_fwd / ifft16k_* of gacq_ldsfft.hip -- every register/lane/LDS address of the three passes, the two wave-private transposes and
the cross-wave exchange -- in numpy.  The forward model must equal numpy.fft in the digit-permuted order the kernels store spectra in,
the inverse model must undo it, and no LDS access may put two lanes of a 16-lane store group / 32-lane load group on one bank slot."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft16k_layouts_compute_the_dft_and_are_bank_conflict_free():
    spec = importlib.util.spec_from_file_location("model_fft16k", os.path.join(ROOT, "tools", "model_fft16k.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ferr, ierr, conflicts = m.run()
    assert ferr < 1e-12 and ierr < 1e-12, (ferr, ierr)
    assert len(conflicts) == 12 and all(v == 0 for v in conflicts.values()), conflicts


def test_radix32_fft16k_layouts_compute_the_dft_and_are_bank_conflict_free():
    """The radix-32 form (gacq_lds16k.hip, GACQ_OPT_LDS_VARIANT = 32): 512 threads x 32 points, 32 x 32 x 16, two exchanges."""
    spec = importlib.util.spec_from_file_location("model_fft16k_r32", os.path.join(ROOT, "tools", "model_fft16k_r32.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ferr, ierr, conflicts = m.run()
    assert ferr < 1e-12 and ierr < 1e-12, (ferr, ierr)
    assert len(conflicts) == 8 and all(v == 0 for v in conflicts.values()), conflicts
    assert m.staged_row_conflicts() <= 32 * m.S * 8           # the LDS-DMA landing area of the last wave ends inside its own regions
