#!/usr/bin/env python3
"""Run a tools/ script against a variant build of the library: tools/variant.py <variant-name|path/to/libgacq.so> <script.py> [args...]
(variants are built by tools/build_variant.sh into gnss-dsp-tools_amd/build/variants/<name>/; the product lib/libgacq.so is never touched)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1]
if not lib.endswith(".so"):
    lib = os.path.join(ROOT, "gnss-dsp-tools_amd", "build", "variants", lib, "libgacq.so")
if not os.path.exists(lib):
    sys.exit("variant library %s not found (tools/build_variant.sh)" % lib)
GACQ_TUNING_LIB = lib          # read by gnss-dsp-tools_amd/_native.py from __main__
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, init_globals={"GACQ_TUNING_LIB": lib}, run_name="__main__")
