"""ctypes binding of libgacq.so (the C ABI declared in include/gacq.h).

The library is built in-tree by __graft_entry__.build() / `make -C gnss-dsp-tools_amd/csrc`.
There is NO Python/CPU fallback for the engine: if the shared object is missing this module
raises at import, and the engine entry points raise when no GPU is visible.
"""
import ctypes
import os

import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgacq.so")
# Tuning tools only (tools/variant.py): a script that defines GACQ_TUNING_LIB = "<path>" in its __main__ module before
# importing the package loads that build (an A/B or diagnostic variant under build/variants/) instead of the product library.
# lib/libgacq.so itself is never replaced, and nothing reads the environment.
_tuning = getattr(sys.modules.get("__main__"), "GACQ_TUNING_LIB", None)
if _tuning:
    LIB_PATH = os.path.abspath(_tuning)

# One HIP runtime per process: when torch is (or will be) in the process its bundled
# libamdhip64/librocfft (same SONAMEs as /opt/rocm's) must be the ones that get bound, so it
# is imported before libgacq.so is dlopen'ed.  Device buffers are torch tensors anyway.
# A program that never touches a device tensor -- the command-line shim, a stub around search() on numpy samples -- can skip that
# (importing torch is ~0.8 s, most of such a run): it defines GACQ_NO_TORCH = True in its __main__ module before importing the package.
# libgacq.so then binds /opt/rocm's runtime through its rpath, and the entry points that take torch tensors refuse to run (require_torch).
NO_TORCH = bool(getattr(sys.modules.get("__main__"), "GACQ_NO_TORCH", False)) and "torch" not in sys.modules
torch = None
if not NO_TORCH:
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is part of the image; C-only users do not need it
        torch = None


if NO_TORCH:
    # ... and nothing else in the process may bring torch (a second HIP runtime) in afterwards
    import importlib.abc

    class _NoTorchLater(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if name == "torch" or name.startswith("torch."):
                raise ImportError("this process loaded libgacq.so without torch (GACQ_NO_TORCH in __main__): torch and the device-tensor "
                                  "entry points of the package are not available in it")
            return None

    sys.meta_path.insert(0, _NoTorchLater())


def require_torch():
    """For the device-tensor entry points (raises in a GACQ_NO_TORCH process, see above)."""
    import torch as _t
    return _t

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libgacq.so not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C gnss-dsp-tools_amd/csrc` (needs hipcc). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

c_int_p = ctypes.POINTER(ctypes.c_int)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class SigDesc(ctypes.Structure):
    _fields_ = [("code_length", ctypes.c_int), ("n", ctypes.c_int), ("pad", ctypes.c_int),
                ("boc", ctypes.c_int), ("metric_mode", ctypes.c_int), ("fold_code", ctypes.c_int),
                ("fs", ctypes.c_double)]


class Result(ctypes.Structure):
    _fields_ = [("metric", ctypes.c_double), ("code_chips", ctypes.c_double),
                ("doppler_hz", ctypes.c_double), ("idx", ctypes.c_int), ("d_index", ctypes.c_int)]


class Peak(ctypes.Structure):
    _fields_ = [("metric", ctypes.c_double), ("idx", ctypes.c_int), ("d_index", ctypes.c_int)]


ERRORS = {0: "GACQ_OK", -1: "GACQ_ERR_BAD_ARG", -2: "GACQ_ERR_UNKNOWN_CODE", -3: "GACQ_ERR_BAD_PRN",
          -4: "GACQ_ERR_HIP", -5: "GACQ_ERR_ROCFFT", -6: "GACQ_ERR_SHORT_INPUT", -7: "GACQ_ERR_NO_DEVICE",
          -8: "GACQ_ERR_INTERNAL", -9: "GACQ_ERR_UNSUPPORTED"}

# GACQ_OPT_* of include/gacq.h
OPTIONS = {"fused_inner": 0, "fused_16k": 1, "lds_variant": 2, "lds_pch": 3, "split_pch": 4, "split_teams": 5, "fused_4k": 6, "split_dt": 7, "fe_generic": 8, "lds_ugroup": 9, "search1": 10, "bar_upload": 11, "watch_results": 12,
           "tie_safe": 13, "tie_eps_ppb": 14, "tie_cap": 15, "fused_c128": 16, "split_mfma": 17}

# name -> (restype, argtypes): every symbol include/gacq.h declares
SYMBOLS = {
    "gacq_code_count": (ctypes.c_int, []),
    "gacq_code_name": (ctypes.c_char_p, [ctypes.c_int]),
    "gacq_code_length": (ctypes.c_int, [ctypes.c_char_p]),
    "gacq_code_chip_rate": (ctypes.c_double, [ctypes.c_char_p]),
    "gacq_code_prns": (ctypes.c_int, [ctypes.c_char_p, c_int_p, ctypes.c_int]),
    "gacq_code_chips": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, c_uint8_p, ctypes.c_int]),
    "gacq_code_replica": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p]),
    "gacq_device_count": (ctypes.c_int, []),
    "gacq_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_destroy": (None, [ctypes.c_void_p]),
    "gacq_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "gacq_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gacq_use_null_stream": (ctypes.c_int, [ctypes.c_void_p]),
    "gacq_set_engine": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gacq_set_workspace_limit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "gacq_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_long]),
    "gacq_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_long)]),
    "gacq_signal_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SigDesc), ctypes.c_char_p,
                                          c_int_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_signal_create_chips": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SigDesc), c_uint8_p,
                                                ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_signal_destroy": (None, [ctypes.c_void_p]),
    "gacq_signal_fft_length": (ctypes.c_int, [ctypes.c_void_p]),
    "gacq_signal_spectrum": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p]),
    # the sample and Doppler pointers are declared void*: the latency path passes plain addresses (no ctypes.cast per call); typed
    # pointer objects are accepted for a void* parameter as well
    "gacq_search": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_int_p, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.POINTER(Result)]),
    "gacq_search_batch_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                             c_int_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p,
                                             ctypes.c_int, ctypes.c_void_p]),
    # complex128 samples (what the reference's search() is handed): same signatures, sample pointer to interleaved doubles
    "gacq_search64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_int_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.POINTER(Result)]),
    "gacq_search_batch_dev64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                               c_int_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p,
                                               ctypes.c_int, ctypes.c_void_p]),
    "gacq_search_batch": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_int, c_int_p, ctypes.c_int,
                                         c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.POINTER(Result)]),
    "gacq_group_create": (ctypes.c_int, [c_int_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_group_destroy": (None, [ctypes.c_void_p]),
    "gacq_group_size": (ctypes.c_int, [ctypes.c_void_p]),
    "gacq_group_set_exchange": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gacq_group_member": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "gacq_group_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "gacq_group_signal_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SigDesc), ctypes.c_char_p, c_int_p, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_group_signal_destroy": (None, [ctypes.c_void_p]),
    "gacq_group_search_batch": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_int, c_int_p, ctypes.c_int,
                                               c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.POINTER(Result)]),
    "gacq_get_tie_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]),
    "gacq_merge_peaks_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_int_p, ctypes.c_long,
                                            ctypes.c_void_p]),
    "gacq_merge_peaks_tiesafe_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_int_p, ctypes.c_int,
                                                    c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_int_p,
                                                    ctypes.c_void_p]),
    "gacq_merge_peaks_tiesafe_dev64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_int_p, ctypes.c_int,
                                                      c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_int_p,
                                                      ctypes.c_void_p]),
    "gacq_finalize": (ctypes.c_int, [ctypes.POINTER(SigDesc), ctypes.POINTER(Peak), ctypes.c_int, c_int_p, ctypes.c_int,
                                     c_double_p, ctypes.c_int, ctypes.POINTER(Result)]),
    "gacq_firwin_hann": (ctypes.c_int, [ctypes.c_int, ctypes.c_double, c_double_p]),
    "gacq_frontend_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
                                         c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p]),
    "gacq_longcode_search": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_char_p,
                                            ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            c_double_p]),
    "gacq_longcode_search_int8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
                                                 ctypes.c_char_p, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, c_double_p]),
    "gacq_correlate_batch": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, c_int_p,
                                            c_double_p, c_double_p, c_double_p, ctypes.c_int, c_double_p]),
    "gacq_longcode_search_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_char_p,
                                                ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                c_double_p]),
    "gacq_correlate_batch_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, c_int_p,
                                                c_double_p, c_double_p, c_double_p, ctypes.c_int, c_double_p]),
    "gacq_mix_int8_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "gacq_stream_probe": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, c_double_p]),
    "gacq_stream_create_cu_mask": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gacq_stream_destroy": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "gacq_cu_census": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]),
    "gacq_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gacq_get_stage_time": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.POINTER(ctypes.c_long)]),
    "gacq_reset_stage_times": (ctypes.c_int, [ctypes.c_void_p]),
    "gacq_stage_name": (ctypes.c_char_p, [ctypes.c_int]),
    "gacq_debug_nco_indices": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, c_int_p]),
    "gacq_debug_fft_plans": (ctypes.c_int, [ctypes.c_void_p]),
    "gacq_acquire_int8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, c_double_p, ctypes.c_int,
                                         ctypes.c_size_t, c_int_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_void_p]),
    "gacq_debug_row": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int, c_float_p]),
}

for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)          # AttributeError here == the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class GacqError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "GACQ_ERR_?"), code, message))
        self.code = code


WARN_TIE_LIST_FULL = 1


class TieListFull(UserWarning):
    """gacq.h GACQ_WARN_TIE_LIST_FULL: the results are valid, but the call's near-ties were NOT re-evaluated in complex128."""


def check(rc, ctx=None):
    if rc < 0:
        msg = lib.gacq_last_error(ctx)
        raise GacqError(rc, msg.decode() if msg else "")
    return rc


def check_search(rc, ctx=None):
    """check() for gacq_search / gacq_search64, whose positive return is a warning (other entry points return counts)."""
    if rc == WARN_TIE_LIST_FULL:
        import warnings
        warnings.warn("tie-safe re-evaluation list full (option tie_cap / workspace limit): every near-tied (epoch, item) of this call kept "
                      "its fp32 location", TieListFull, stacklevel=3)
        return rc
    return check(rc, ctx)
