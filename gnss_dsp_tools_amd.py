"""Import alias: the package directory is named ``gnss-dsp-tools_amd`` (not a valid Python
identifier), so ``import gnss_dsp_tools_amd`` resolves here and this stub replaces itself in
``sys.modules`` with the real package loaded from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gnss-dsp-tools_amd")
_spec = importlib.util.spec_from_file_location(
    "gnss_dsp_tools_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["gnss_dsp_tools_amd"] = _pkg
_spec.loader.exec_module(_pkg)
