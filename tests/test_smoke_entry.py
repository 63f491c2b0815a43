"""CPU: __graft_entry__.build() produces the library; GPU: smoke() passes."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_products_exist():
    assert os.path.exists(os.path.join(ROOT, "gnss-dsp-tools_amd", "lib", "libgacq.so"))


@pytest.mark.gpu
def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
