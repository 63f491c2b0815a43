#!/usr/bin/env python3
"""pytest as a script, so that tools/variant.py can run the test-suite against a variant build:
tools/variant.py <variant> tools/run_pytest.py tests/test_tie_safe.py -x -q"""
import sys

import pytest

sys.exit(pytest.main(sys.argv[1:]))
