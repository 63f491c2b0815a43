"""Host-side mirror of the reference's acquisition entry point.

The reference defines, inline in every acquire-*.py,

    search(x, prn, doppler_search, ms) -> (m_metric, m_code, m_doppler)      acquire-gps-l1.py:18-40

and maps it over PRNs with multiprocessing.Pool (acquire-gps-l1.py:100-108).  Here the same call
surface is kept (``make_search(name)`` returns a function with exactly that signature and return
tuple) but every PRN / Doppler bin / block of the search runs on the GPU through libgacq.so.
``search_all`` replaces the Pool.map: all items in one launch sequence, forward FFTs shared.

No CPU fallback: constructing an Engine without a visible GPU raises.
"""
import ctypes
import weakref

import numpy as np

from . import _native as nat
from . import codes
from . import signals as _signals

PEAK_DTYPE = np.dtype([("metric", "<f8"), ("idx", "<i4"), ("d_index", "<i4")])
RESULT_DTYPE = np.dtype([("metric", "<f8"), ("code_chips", "<f8"), ("doppler_hz", "<f8"), ("idx", "<i4"), ("d_index", "<i4")])      # gacq_result


def parse_list_ranges(s, sep='-'):
    """'1,3,7-14' -> [1,3,7,...,14]   (option syntax of gnsstools/util.py:1-10)."""
    out = []
    for tok in s.split(','):
        ends = tok.split(sep)
        if len(ends) == 1:
            out.append(int(ends[0]))
        else:
            out.extend(range(int(ends[0]), int(ends[1]) + 1))
    return out


def parse_list_floats(s):
    """'-7000,7000,200' -> [-7000.0, 7000.0, 200.0]   (gnsstools/util.py:12-14)."""
    return [float(v) for v in s.split(',')]


def doppler_grid(doppler_search):
    """The half-open grid the reference iterates: np.arange(min,max,incr) (acquire-gps-l1.py:26)."""
    lo, hi, step = doppler_search
    return np.ascontiguousarray(np.arange(lo, hi, step), dtype=np.float64)


class AcqSignal:
    """Code spectra of one signal for a fixed item list, resident on the device."""

    def __init__(self, engine, sig, prns, chips=None):
        """chips: optional uint8 array [len(prns), code_length] of {0,1} chips supplied by the caller
        (gacq_signal_create_chips) instead of the built-in generators."""
        self.engine = engine
        self.sig = sig
        self.prns = list(prns)
        L = nat.check(nat.lib.gacq_code_length(sig.code.encode()))
        self.code_length = L
        self.desc = nat.SigDesc(L, sig.n, int(sig.pad), int(sig.boc), int(sig.normalised), int(sig.fold), sig.fs)
        h = ctypes.c_void_p()
        if chips is None:
            arr = (ctypes.c_int * len(self.prns))(*self.prns)
            nat.check(nat.lib.gacq_signal_create(engine._ctx, ctypes.byref(self.desc), sig.code.encode(), arr,
                                                 len(self.prns), ctypes.byref(h)), engine._ctx)
        else:
            c = np.ascontiguousarray(chips, dtype=np.uint8)
            if c.shape != (len(self.prns), L):
                raise ValueError("chips must have shape (%d, %d)" % (len(self.prns), L))
            nat.check(nat.lib.gacq_signal_create_chips(engine._ctx, ctypes.byref(self.desc), c.ctypes.data_as(nat.c_uint8_p),
                                                       len(self.prns), ctypes.byref(h)), engine._ctx)
        self._h = h
        self._index = {p: i for i, p in enumerate(self.prns)}
        engine._live.add(self)              # Engine.close() destroys every signal created on it before the context

    def close(self):
        """Free the device-side code spectra.  A signal never outlives its context: Engine.close() closes it first."""
        if self._h:
            nat.lib.gacq_signal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nco_indices(self, kernel, doppler, bias_hz=0.0):
        """Test hook (gacq_debug_nco_indices): the table-NCO index vector of one (doppler, bias) row as the forward kernel
        `kernel` (1 mix_nco, 2 LDS forward, 3 split outer forward, 4 fused 16K, 5 prime-factor outer forward) computes it on the device; int32 [N]."""
        out = np.empty(self.sig.nfft, dtype=np.int32)
        nat.check(nat.lib.gacq_debug_nco_indices(self._h, int(kernel), float(doppler), float(bias_hz), out.ctypes.data_as(nat.c_int_p)),
                  self.engine._ctx)
        return out

    def spectrum(self, prn):
        out = np.empty(self.sig.nfft, dtype=np.complex64)
        nat.check(nat.lib.gacq_signal_spectrum(self._h, self._index[prn], out.ctypes.data_as(nat.c_float_p)), self.engine._ctx)
        return out


class MaskedStream:
    """A HIP stream with a hardware queue of its own (gacq_stream_create_cu_mask), optionally limited to `cus` of the device's compute
    units: the low `cus` bits of the mask, which the driver deals round-robin over the XCDs (cus / 8 CUs of every XCD); cus = None
    selects every CU.  Why not a plain stream: the HIP runtime multiplexes plain streams over a few hardware queues (four by default),
    and two streams that land on the same queue run their kernels one after the other however independent they are; a stream created
    with a CU mask always gets its own queue, so work on two of them really runs side by side (tools/exp_lanes_debug.py).
    `.handle` is the hipStream_t for Engine.set_stream; `.torch_stream` wraps it as a torch.cuda.ExternalStream, so torch's allocator
    and wait_stream() see it."""

    def __init__(self, device, cus=None, total_cus=None):
        if total_cus is None:
            total_cus = nat.require_torch().cuda.get_device_properties(int(device)).multi_processor_count       # 256 on MI355X
        cus = int(total_cus if cus is None else cus)
        if not 0 < cus <= total_cus:
            raise ValueError("cus must be in 1..%d" % total_cus)
        nwords = (total_cus + 31) // 32
        words = (ctypes.c_uint32 * nwords)(*[((1 << max(0, min(32, cus - 32 * i))) - 1) & 0xffffffff for i in range(nwords)])
        h = ctypes.c_void_p()
        nat.check(nat.lib.gacq_stream_create_cu_mask(int(device), words, nwords, ctypes.byref(h)))
        self.device, self.cus, self.handle = int(device), cus, h.value
        self._torch_stream = None

    @property
    def torch_stream(self):
        if self._torch_stream is None:
            torch = nat.require_torch()
            self._torch_stream = torch.cuda.ExternalStream(self.handle, device=torch.device("cuda", self.device))
        return self._torch_stream

    def close(self):
        if self.handle:
            nat.lib.gacq_stream_destroy(self.device, ctypes.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One acquisition context bound to one GPU (one process per GPU).  Like the gacq_ctx it wraps, an Engine is single-threaded:
    the plan cache, the reusable result buffer and the library's workspaces are per context -- use one Engine per thread."""

    def __init__(self, device=0, engine=0, workspace_bytes=None):
        h = ctypes.c_void_p()
        nat.check(nat.lib.gacq_create(int(device), ctypes.byref(h)))
        self._ctx = h
        self.device = int(device)
        self._signals = {}
        self._families = {}
        self._live = weakref.WeakSet()      # every AcqSignal created on this context (also user-held ones)
        if engine:
            self.set_engine(engine)
        if workspace_bytes:
            nat.check(nat.lib.gacq_set_workspace_limit(self._ctx, int(workspace_bytes)), self._ctx)

    # -- configuration -----------------------------------------------------------------------
    def set_engine(self, engine):
        """0 auto, 1 rocFFT pipeline, 2 LDS-resident FFT kernels, 3 / 4 split engines, 5 complex128 verification pipeline."""
        nat.check(nat.lib.gacq_set_engine(self._ctx, int(engine)), self._ctx)

    def set_stream(self, stream_handle):
        """Launch on the caller's HIP stream, e.g. torch.cuda.current_stream().cuda_stream.  Handle 0 is the legacy
        default stream (torch's current stream outside any stream context); None returns to the engine's own stream."""
        if stream_handle is None:
            nat.check(nat.lib.gacq_set_stream(self._ctx, None), self._ctx)
        elif int(stream_handle) == 0:
            nat.check(nat.lib.gacq_use_null_stream(self._ctx), self._ctx)
        else:
            nat.check(nat.lib.gacq_set_stream(self._ctx, ctypes.c_void_p(int(stream_handle))), self._ctx)

    def use_torch_stream(self, device=None):
        """Order the engine's launches with torch work (incl. RCCL collectives) on torch's current stream."""
        torch = nat.require_torch()
        self.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def set_option(self, option, value):
        """gacq_set_option: tuning switches of the launch path (include/gacq.h GACQ_OPT_*), by number or by name."""
        if isinstance(option, str):
            option = nat.OPTIONS[option.lower()]
        nat.check(nat.lib.gacq_set_option(self._ctx, int(option), int(value)), self._ctx)

    def get_option(self, option):
        if isinstance(option, str):
            option = nat.OPTIONS[option.lower()]
        v = ctypes.c_long()
        nat.check(nat.lib.gacq_get_option(self._ctx, int(option), ctypes.byref(v)), self._ctx)
        return v.value

    def tie_stats(self):
        """gacq_get_tie_stats: {ambiguous (epoch, item) pairs found, rows re-evaluated in complex128, pairs that kept their fp32 answer
        (list full / unsupported length), pairs whose location the re-evaluation changed} since the context was created."""
        v = (ctypes.c_longlong * 4)()
        nat.check(nat.lib.gacq_get_tie_stats(self._ctx, v), self._ctx)
        return {"ambiguous_pairs": v[0], "rows_reevaluated": v[1], "kept_fp32": v[2], "locations_changed": v[3]}

    def fft_plans(self):
        """gacq_debug_fft_plans: rocFFT plans this context has created (0 on the default engines' path, code spectra included)."""
        return nat.check(nat.lib.gacq_debug_fft_plans(self._ctx), self._ctx)

    def stream_probe(self, kind, nbytes=2 << 30, reps=10):
        """gacq_stream_probe: GB/s a tuned streaming kernel reaches on this device; kind "fill", "read" or "copy" (read + write bytes)."""
        v = ctypes.c_double()
        nat.check(nat.lib.gacq_stream_probe(self._ctx, {"fill": 0, "read": 1, "copy": 2, "fill_read": 3, "fill_plain": 4, "read_plain": 5, "fill_read_plain": 6}[kind], int(nbytes), int(reps), ctypes.byref(v)), self._ctx)
        return v.value

    def cu_census(self, nworkgroups=4096):
        """gacq_cu_census: the distinct (xcc, se, sh, cu) places the workgroups of one launch on the context's current stream ran on,
        as a sorted list of (xcc, se, sh, cu) tuples -- what a CU-masked stream really selects."""
        w = np.zeros(int(nworkgroups), dtype=np.uint32)
        nat.check(nat.lib.gacq_cu_census(self._ctx, int(nworkgroups), w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))), self._ctx)
        return sorted({(int(v) >> 8, (int(v) >> 5) & 7, (int(v) >> 4) & 1, int(v) & 15) for v in w})

    def set_profiling(self, on):
        nat.check(nat.lib.gacq_set_profiling(self._ctx, int(bool(on))), self._ctx)

    def reset_stage_times(self):
        nat.check(nat.lib.gacq_reset_stage_times(self._ctx), self._ctx)

    def stage_times(self):
        """{stage name: (total_ms, launches)} from HIP events on the launch stream."""
        out = {}
        for s in range(7):
            ms, n = ctypes.c_double(), ctypes.c_long()
            nat.check(nat.lib.gacq_get_stage_time(self._ctx, s, ctypes.byref(ms), ctypes.byref(n)), self._ctx)
            out[nat.lib.gacq_stage_name(s).decode()] = (ms.value, n.value)
        return out

    def close(self):
        self._plan_key = None
        for s in list(self._live):
            s.close()
        self._signals.clear()
        self._families.clear()
        if self._ctx:
            nat.lib.gacq_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- signals -----------------------------------------------------------------------------
    def signal(self, name, prns, chips=None):
        """Device-resident code spectra for `prns` (cached).  With `chips` the caller's own {0,1} chip rows are used and the
        entry replaces any cached one for the same (signal, prns)."""
        sig = _signals.get(name) if isinstance(name, str) else name
        key = (sig.name, tuple(prns))
        if chips is not None:
            self._plan_key = None                      # a cached plan may point at the signal replaced here
            if key in self._signals:
                self._signals.pop(key).close()
            self._signals[key] = AcqSignal(self, sig, prns, chips)
        elif key not in self._signals:
            self._signals[key] = AcqSignal(self, sig, prns)
        return self._signals[key]

    def _plan(self, name, items):
        """Map reference 'items' (PRNs, or GLONASS channels) to (AcqSignal, item indices, per-item bias).  The last plan is kept: a
        scan calls search_all with the same item list for every block of samples."""
        sig = _signals.get(name) if isinstance(name, str) else name
        key = (sig.name, tuple(items))
        # identity, not id(): a caller-made descriptor may reuse the name of a built-in one, and the cache entry holds a reference to its
        # descriptor so that a collected one cannot be impersonated by a new object at the same address
        if getattr(self, "_plan_key", None) == key and getattr(self, "_plan_sig", None) is sig:
            return self._plan_val
        val = self._plan_uncached(sig, items)
        self._plan_key, self._plan_sig, self._plan_val = key, sig, val
        # ctypes views of the index / bias arrays and a reusable result buffer: numpy's .ctypes accessor costs ~1 us per use, which shows
        # on the single-search latency path
        _, idx, bias = val
        res = (nat.Result * len(idx))()
        self._plan_c = (idx.ctypes.data_as(nat.c_int_p), bias.ctypes.data_as(nat.c_double_p) if bias is not None else None,
                        res, np.frombuffer(res, dtype=RESULT_DTYPE))
        return val

    def _plan_uncached(self, sig, items):
        items = [int(i) for i in items]
        if sig.bias_hz:
            s = self.signal(sig, [0])                                  # one shared code (glonass/ca.py:27)
            idx = np.zeros(len(items), dtype=np.int32)
            bias = np.array([sig.bias_hz * c for c in items], dtype=np.float64)   # 562500*chan
            return s, idx, bias
        uniq = sorted(set(items))
        s = self.signal(sig, uniq)
        idx = np.array([s._index[p] for p in items], dtype=np.int32)
        return s, idx, None

    # -- the reference call surface ------------------------------------------------------------
    def search_all(self, name, x, items, doppler_search, ms):
        """[search(x, item, doppler_search, ms) for item in items] of acquire-<name>.py, one GPU pass."""
        sig = _signals.get(name) if isinstance(name, str) else name
        return self.search_blocks(sig, x, items, doppler_grid(doppler_search), sig.blocks(int(ms)))

    def search(self, name, x, item, doppler_search, ms):
        """search(x, prn, doppler_search, ms) -> (metric, code, doppler)   acquire-gps-l1.py:18-40."""
        return self.search_all(name, x, [item], doppler_search, ms)[0]

    def search_blocks(self, name, x, items, dopplers, blocks):
        """Engine-level form: explicit Doppler values and block count B."""
        sig = _signals.get(name) if isinstance(name, str) else name
        if len(items) == 0:
            return []                                   # the reference maps worker over an empty list
        s, idx, bias = self._plan(sig, items)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        blocks = max(int(blocks), 0)                    # range(negative) is empty in the reference
        # an empty Doppler grid: the reference's loop body never runs, x is never looked at (acquire-gps-l1.py:25-26,40)
        need = sig.samples_needed(blocks) if len(dopplers) else 0
        x = np.asarray(x)
        if x.ndim != 1:
            raise ValueError("x must be a 1-D complex array")
        if len(x) < need:
            # the reference fails here with numpy's broadcast ValueError (short last block * w)
            raise ValueError("operands could not be broadcast together: search needs %d samples "
                             "(%d block(s) of n=%d%s), x has %d" % (need, blocks, sig.n, ", padded" if sig.pad else "", len(x)))
        # complex128 samples (what the reference's search() gets from np.interp, acquire-gps-l1.py:94-96) go down unrounded
        # (gacq_search64: engine 5 and the tie-safe re-evaluation read them as given); everything else is searched as complex64
        wide = x.dtype == np.complex128
        xc = np.ascontiguousarray(x[:need], dtype=np.complex128 if wide else np.complex64) if need else np.zeros(1, dtype=np.complex64)
        idx_p, bias_p, res, view = self._plan_c
        fn = nat.lib.gacq_search64 if (wide and need and len(dopplers)) else nat.lib.gacq_search
        rc = fn(s._h, xc.__array_interface__["data"][0], len(xc), idx_p, len(idx),
                dopplers.__array_interface__["data"][0] if len(dopplers) else None, len(dopplers), bias_p, blocks, res)
        if rc:
            nat.check_search(rc, self._ctx)
        return _as_tuples(view)

    def _family(self, names, items, ms=None):
        """(base descriptor, stacked AcqSignal or None, per-signal item lists) for signals that differ only in their code
        tables; the stacked signal holds one row per (signal, item) in order and is cached."""
        sigs = [_signals.get(n) if isinstance(n, str) else n for n in names]
        if len(sigs) != len(items) or not sigs:
            raise ValueError("search_family: one item list per signal name")
        base = sigs[0]
        shape = lambda g: (g.fs, g.n, g.pad, g.boc, g.normalised, g.fold, g.bias_hz, None if ms is None else g.blocks(int(ms)),
                           nat.check(nat.lib.gacq_code_length(g.code.encode())))
        for g in sigs[1:]:
            if shape(g) != shape(base):
                raise ValueError("search_family: %s and %s differ in more than their code tables" % (base.name, g.name))
        if base.bias_hz:
            raise ValueError("search_family: FDMA signals share one code; use search_all")
        lists = [[int(i) for i in it] for it in items]
        key = (tuple(g.name for g in sigs), tuple(tuple(it) for it in lists))
        if key not in self._families:
            rows = [codes.chips(g.code, p) for g, it in zip(sigs, lists) for p in it]
            self._families[key] = AcqSignal(self, base, list(range(len(rows))), np.stack(rows).astype(np.uint8)) if rows else None
        return base, self._families[key], lists

    def search_family(self, names, x, items, doppler_search, ms):
        """Signals that differ only in their code tables -- E1B + E1C, L5I + L5Q, E5aI + E5aQ, L1Cd + L1Cp ... -- searched
        in ONE pass over x: the mix and the forward transforms are shared by all of them (BASELINE config 3 counts them
        once), only the correlation rows multiply.  `items` is a list of item lists, one per name.  Returns one result list
        per name, each identical to search_all(name, x, items_k, doppler_search, ms)."""
        base, fam, lists = self._family(names, items, ms)
        if fam is None:
            return [[] for _ in lists]
        total = len(fam.prns)
        dopplers = doppler_grid(doppler_search)
        blocks = max(base.blocks(int(ms)), 0)
        need = base.samples_needed(blocks)
        x = np.asarray(x)
        if x.ndim != 1 or len(x) < need:
            raise ValueError("operands could not be broadcast together: search needs %d samples, x has shape %r" % (need, x.shape))
        wide = x.dtype == np.complex128 and need > 0 and len(dopplers) > 0          # see search_blocks
        xc = np.ascontiguousarray(x[:need], dtype=np.complex128 if wide else np.complex64) if need else np.zeros(1, dtype=np.complex64)
        idx = np.arange(total, dtype=np.int32)
        res = (nat.Result * total)()
        fn = nat.lib.gacq_search64 if wide else nat.lib.gacq_search
        nat.check_search(fn(fam._h, xc.ctypes.data_as(ctypes.c_void_p), len(xc), idx.ctypes.data_as(nat.c_int_p), total,
                            dopplers.ctypes.data_as(nat.c_double_p), len(dopplers), None, blocks, res), self._ctx)
        flat = _as_tuples(res)
        out, at = [], 0
        for it in lists:
            out.append(flat[at:at + len(it)])
            at += len(it)
        return out

    def search_family_batch_dev(self, names, x_dev, items, dopplers, blocks, out=None):
        """Device-resident, batched form of search_family: x_dev [nepoch, nsamp] complex64 on the GPU -> peaks tensor
        [nepoch, sum(len(items_k)), 2] (gacq_peak records, signals concatenated in order).  Asynchronous."""
        torch = nat.require_torch()
        base, fam, lists = self._family(names, items)
        if fam is None:
            return torch.empty((x_dev.shape[0], 0, 2), dtype=torch.float64, device=x_dev.device)
        return self.search_batch_dev(fam.sig, x_dev, fam.prns, dopplers, blocks, out=out, _signal=fam)

    def debug_row(self, name, x, item, doppler, blocks):
        """Accumulated magnitude row q[0:N] of one (item, doppler) via the rocFFT pipeline."""
        sig = _signals.get(name) if isinstance(name, str) else name
        s, idx, bias = self._plan(sig, [item])
        need = sig.samples_needed(blocks)
        xc = np.ascontiguousarray(np.asarray(x)[:need], dtype=np.complex64)
        if len(xc) < need:
            raise ValueError("x too short: %d < %d" % (len(xc), need))
        q = np.empty(sig.nfft, dtype=np.float32)
        nat.check(nat.lib.gacq_debug_row(s._h, xc.ctypes.data_as(nat.c_float_p), len(xc), int(idx[0]), float(doppler),
                                         float(bias[0]) if bias is not None else 0.0, int(blocks),
                                         q.ctypes.data_as(nat.c_float_p)), self._ctx)
        return q

    # -- device-resident batched form (bench / sharded path) --------------------------------------
    def search_batch_dev(self, name, x_dev, items, dopplers, blocks, out=None, _signal=None):
        """x_dev: torch complex64 (or complex128: searched unrounded where it matters, gacq_search_batch_dev64) CUDA tensor
        [nepoch, nsamp]; returns a torch tensor [nepoch, nitems, 2] of float64 whose 16-byte rows are gacq_peak records (view with
        PEAK_DTYPE on the host).  Asynchronous on the engine's stream."""
        torch = nat.require_torch()
        sig = _signals.get(name) if isinstance(name, str) else name
        if not (x_dev.is_cuda and x_dev.dtype in (torch.complex64, torch.complex128) and x_dev.dim() == 2 and x_dev.is_contiguous()):
            raise ValueError("x_dev must be a contiguous 2-D complex64 (or complex128) CUDA tensor")
        if len(items) == 0:
            return torch.empty((x_dev.shape[0], 0, 2), dtype=torch.float64, device=x_dev.device)
        if _signal is not None:                # a stacked family signal: items are its row numbers
            s, idx, bias = _signal, np.array([_signal._index[i] for i in items], dtype=np.int32), None
        else:
            s, idx, bias = self._plan(sig, items)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        nepoch, nsamp = x_dev.shape
        if out is None:
            out = torch.empty((nepoch, len(idx), 2), dtype=torch.float64, device=x_dev.device)
        elif not (torch.is_tensor(out) and out.dtype == torch.float64 and tuple(out.shape) == (nepoch, len(idx), 2) and out.is_contiguous()
                  and (out.is_cuda or out.is_pinned())):
            # the last kernel writes 16-byte gacq_peak records into it: a wrong buffer would be overwritten silently
            raise ValueError("out must be a contiguous float64 tensor of shape (%d, %d, 2) on the device (or pinned host memory)" % (nepoch, len(idx)))
        fn = nat.lib.gacq_search_batch_dev64 if x_dev.dtype == torch.complex128 else nat.lib.gacq_search_batch_dev
        nat.check(fn(
            s._h, ctypes.c_void_p(x_dev.data_ptr()), nsamp, nepoch, idx.ctypes.data_as(nat.c_int_p), len(idx),
            dopplers.ctypes.data_as(nat.c_double_p), len(dopplers),
            bias.ctypes.data_as(nat.c_double_p) if bias is not None else None, int(blocks),
            ctypes.c_void_p(out.data_ptr())), self._ctx)
        return out

    def search_batch_host(self, name, xs, items, dopplers, blocks, raw=False):
        """Host-buffer batch (gacq_search_batch): xs [nepoch, nsamp] complex64 in host memory -> per epoch the list of the
        reference's (metric, code, doppler) tuples (raw=True: the gacq_result records as a structured array [nepoch, nitems],
        without building Python tuples).  No torch involved: the library stages the epochs through its own pinned ring, H2D
        copies overlapped with the kernels of the previous chunk."""
        sig = _signals.get(name) if isinstance(name, str) else name
        xs = np.ascontiguousarray(xs, dtype=np.complex64)
        if xs.ndim != 2:
            raise ValueError("xs must be [nepoch, nsamp]")
        if len(items) == 0:
            return [[] for _ in range(xs.shape[0])]
        s, idx, bias = self._plan(sig, items)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        blocks = max(int(blocks), 0)
        if len(dopplers) and xs.shape[1] < sig.samples_needed(blocks):      # an empty grid never touches the samples
            raise ValueError("operands could not be broadcast together: search needs %d samples per epoch, xs has %d" % (sig.samples_needed(blocks), xs.shape[1]))
        res = (nat.Result * (xs.shape[0] * len(idx)))()
        nat.check(nat.lib.gacq_search_batch(
            s._h, xs.ctypes.data_as(nat.c_float_p), xs.shape[1], xs.shape[0], idx.ctypes.data_as(nat.c_int_p), len(idx),
            dopplers.ctypes.data_as(nat.c_double_p), len(dopplers),
            bias.ctypes.data_as(nat.c_double_p) if bias is not None else None, blocks, res), self._ctx)
        if raw:
            return np.frombuffer(res, dtype=RESULT_DTYPE).reshape(xs.shape[0], len(idx))
        return [[_as_tuple(res[e * len(idx) + p]) for p in range(len(idx))] for e in range(xs.shape[0])]

    def finalize(self, name, items, peaks, dopplers, shard_d0=None):
        return finalize(name, items, peaks, dopplers, shard_d0)

    # -- the main program of an acquire script in one call, no torch (acquire-gps-l1.py:78-108) --------------
    def acquire_int8(self, name, iq_int8, fs, coffset, ms_pad, items, dopplers, blocks):
        """gacq_acquire_int8: iq_int8 = numpy int8 [n, 2] / flat interleaved block at fs (host memory); front-end to ms_pad ms at the
        signal's rate and search of `items` over `dopplers`, all on the device -> the reference's (metric, code, doppler) tuples."""
        sig = _signals.get(name) if isinstance(name, str) else name
        if len(items) == 0:
            return []
        iq = np.ascontiguousarray(iq_int8, dtype=np.int8).reshape(-1)
        n_in = iq.size // 2
        n_out = int(ms_pad) * int(round(sig.fs * 0.001))
        taps = firwin_hann(161, sig.fir_cutoff / (fs / 2))
        s, idx, bias = self._plan(sig, items)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        res = (nat.Result * len(idx))()
        nat.check_search(nat.lib.gacq_acquire_int8(
            s._h, iq.ctypes.data_as(ctypes.c_void_p), n_in, float(fs), float(coffset), taps.ctypes.data_as(nat.c_double_p), len(taps), n_out,
            idx.ctypes.data_as(nat.c_int_p), len(idx), dopplers.ctypes.data_as(nat.c_double_p) if len(dopplers) else None, len(dopplers),
            bias.ctypes.data_as(nat.c_double_p) if bias is not None else None, max(int(blocks), 0), res), self._ctx)
        return _as_tuples(res)

    # -- front-end on the GPU (acquire-gps-l1.py:87-96) ---------------------------------------------------
    def frontend_dev(self, name, iq_int8, fs, coffset, ms_pad):
        """iq_int8: numpy int8 array [n, 2] (or flat interleaved) or a torch int8 CUDA tensor; returns a torch complex64
        CUDA tensor with ms_pad ms of samples at the signal's internal rate, ready for search_batch_dev."""
        torch = nat.require_torch()
        sig = _signals.get(name) if isinstance(name, str) else name
        if not torch.is_tensor(iq_int8):
            iq_int8 = torch.from_numpy(np.array(iq_int8, dtype=np.int8, copy=True)).to("cuda:%d" % self.device)
        self.use_torch_stream(iq_int8.device)            # see mix_int8_dev: torch-owned temporaries and the context's stream
        iq_int8 = iq_int8.contiguous().view(-1)
        n_in = iq_int8.numel() // 2
        per_ms = int(round(sig.fs * 0.001))
        n_out = int(ms_pad) * per_ms
        taps = firwin_hann(161, sig.fir_cutoff / (fs / 2))
        out = torch.empty(n_out, dtype=torch.complex64, device=iq_int8.device)
        nat.check(nat.lib.gacq_frontend_dev(self._ctx, ctypes.c_void_p(iq_int8.data_ptr()), n_in, float(fs), float(coffset),
                                            taps.ctypes.data_as(nat.c_double_p), len(taps), sig.fs, n_out,
                                            ctypes.c_void_p(out.data_ptr())), self._ctx)
        return out

    def mix_int8_dev(self, iq_int8, fs, coffset):
        """nco.mix(x, -coffset/fs, 0) on the GPU (gacq_mix_int8_dev; gnsstools/nco.py:30-41): int8 I/Q (numpy [n, 2] / flat, or a torch
        int8 CUDA tensor) -> torch complex64 CUDA tensor at the same rate, the input of the device-resident long-code search and
        correlators (longcode.search_*, tracking.correlate_batch accept it as x)."""
        torch = nat.require_torch()
        if not torch.is_tensor(iq_int8):
            iq_int8 = torch.from_numpy(np.array(iq_int8, dtype=np.int8, copy=True)).to("cuda:%d" % self.device)
        # the kernel runs on the context's stream: make that torch's current stream, so that the caching allocator cannot hand the
        # temporary above (dropped on return) or `out` to another torch allocation while the kernel still uses them
        self.use_torch_stream(iq_int8.device)
        iq_int8 = iq_int8.contiguous().view(-1)
        n = iq_int8.numel() // 2
        out = torch.empty(n, dtype=torch.complex64, device=iq_int8.device)
        nat.check(nat.lib.gacq_mix_int8_dev(self._ctx, ctypes.c_void_p(iq_int8.data_ptr()), n, float(fs), float(coffset),
                                            ctypes.c_void_p(out.data_ptr())), self._ctx)
        return out

    def merge_peaks_dev(self, gathered, shard_d0, out=None):
        """gathered: torch float64 CUDA tensor [nshard, ..., 2] of gacq_peak records (all_gather output);
        returns the merged [..., 2] tensor with global Doppler indices (device-side, asynchronous)."""
        torch = nat.require_torch()
        nshard = gathered.shape[0]
        n = int(gathered[0].numel() // 2)
        if out is None:
            out = torch.empty(gathered.shape[1:], dtype=torch.float64, device=gathered.device)
        d0 = np.ascontiguousarray(shard_d0, dtype=np.int32)
        nat.check(nat.lib.gacq_merge_peaks_dev(self._ctx, ctypes.c_void_p(gathered.data_ptr()), nshard,
                                               d0.ctypes.data_as(nat.c_int_p), n, ctypes.c_void_p(out.data_ptr())), self._ctx)
        return out


    def merge_peaks_tiesafe_dev(self, name, x_dev, items, dopplers, blocks, gathered, shard_d0, out=None, _signal=None):
        """gacq_merge_peaks_tiesafe_dev: merge_peaks_dev for searches that run with tie-safe locations (the default) -- shard
        winners within eps of the best one are re-evaluated in complex128 on this GPU before the scan decides.  name / x_dev /
        items / blocks are those of the sharded search, `dopplers` its FULL grid."""
        torch = nat.require_torch()
        sig = _signals.get(name) if isinstance(name, str) else name
        if _signal is not None:
            s, idx, bias = _signal, np.array([_signal._index[i] for i in items], dtype=np.int32), None
        else:
            s, idx, bias = self._plan(sig, items)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        nepoch, nsamp = x_dev.shape
        nshard = gathered.shape[0]
        if tuple(gathered.shape[1:]) != (nepoch, len(idx), 2):
            raise ValueError("gathered must have shape (nshard, %d, %d, 2)" % (nepoch, len(idx)))
        if out is None:
            out = torch.empty(gathered.shape[1:], dtype=torch.float64, device=gathered.device)
        d0 = np.ascontiguousarray(shard_d0, dtype=np.int32)
        fn = nat.lib.gacq_merge_peaks_tiesafe_dev64 if x_dev.dtype == torch.complex128 else nat.lib.gacq_merge_peaks_tiesafe_dev
        nat.check(fn(
            s._h, ctypes.c_void_p(x_dev.data_ptr()), nsamp, nepoch, idx.ctypes.data_as(nat.c_int_p), len(idx),
            dopplers.ctypes.data_as(nat.c_double_p), len(dopplers), bias.ctypes.data_as(nat.c_double_p) if bias is not None else None,
            int(blocks), ctypes.c_void_p(gathered.data_ptr()), nshard, d0.ctypes.data_as(nat.c_int_p), ctypes.c_void_p(out.data_ptr())),
            self._ctx)
        return out


class DeviceGroup:
    """Several GPUs driven from one process through the library alone (gacq_group_*): the Doppler grid (or, for coarse grids,
    the item list) is cut over the devices, every device reads the same samples, the 16-byte peak records are merged on the
    host with the reference's tie rule (or, set_exchange("rccl"), gathered over xGMI by RCCL and merged on a device).  `devices` may
    repeat an index (several contexts on one GPU; host exchange only).
    One-process-per-GPU deployments use sharded.ShardedSearch over torch.distributed / RCCL instead."""

    def __init__(self, devices):
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        nat.check(nat.lib.gacq_group_create(ids, len(devices), ctypes.byref(h)))
        self._g = h
        self.devices = [int(d) for d in devices]
        self._signals = {}

    def _check(self, rc):
        if rc < 0:
            msg = nat.lib.gacq_group_last_error(self._g)
            raise nat.GacqError(rc, msg.decode() if msg else "")

    def set_engine(self, engine):
        for k in range(len(self.devices)):
            nat.check(nat.lib.gacq_set_engine(ctypes.c_void_p(nat.lib.gacq_group_member(self._g, k)), int(engine)))

    def set_exchange(self, mode):
        """gacq_group_set_exchange: "host" (records through pinned memory, merged on the host; default) or "rccl" (records stay on
        the devices, one ncclAllGather per chunk over xGMI, tie-safe merge on a member; needs distinct devices)."""
        self._check(nat.lib.gacq_group_set_exchange(self._g, {"host": 0, "rccl": 1}[mode] if isinstance(mode, str) else int(mode)))

    def _signal(self, sig, prns):
        key = (sig.name, tuple(prns))
        if key not in self._signals:
            L = nat.check(nat.lib.gacq_code_length(sig.code.encode()))
            desc = nat.SigDesc(L, sig.n, int(sig.pad), int(sig.boc), int(sig.normalised), int(sig.fold), sig.fs)
            arr = (ctypes.c_int * len(prns))(*prns)
            h = ctypes.c_void_p()
            self._check(nat.lib.gacq_group_signal_create(self._g, ctypes.byref(desc), sig.code.encode(), arr, len(prns), ctypes.byref(h)))
            self._signals[key] = h
        return self._signals[key]

    def search_batch_host(self, name, xs, items, dopplers, blocks):
        """Same contract as Engine.search_batch_host, over all devices of the group."""
        sig = _signals.get(name) if isinstance(name, str) else name
        xs = np.ascontiguousarray(xs, dtype=np.complex64)
        items = [int(i) for i in items]
        if xs.ndim != 2:
            raise ValueError("xs must be [nepoch, nsamp]")
        if not items:
            return [[] for _ in range(xs.shape[0])]
        if sig.bias_hz:
            prns, idx = [0], np.zeros(len(items), dtype=np.int32)
            bias = np.array([sig.bias_hz * c for c in items], dtype=np.float64)
        else:
            prns = sorted(set(items))
            idx = np.array([prns.index(p) for p in items], dtype=np.int32)
            bias = None
        h = self._signal(sig, prns)
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        blocks = max(int(blocks), 0)
        if len(dopplers) and xs.shape[1] < sig.samples_needed(blocks):      # an empty grid never touches the samples
            raise ValueError("operands could not be broadcast together: search needs %d samples per epoch, xs has %d" % (sig.samples_needed(blocks), xs.shape[1]))
        res = (nat.Result * (xs.shape[0] * len(idx)))()
        self._check(nat.lib.gacq_group_search_batch(
            h, xs.ctypes.data_as(nat.c_float_p), xs.shape[1], xs.shape[0], idx.ctypes.data_as(nat.c_int_p), len(idx),
            dopplers.ctypes.data_as(nat.c_double_p), len(dopplers),
            bias.ctypes.data_as(nat.c_double_p) if bias is not None else None, blocks, res))
        return [[_as_tuple(res[e * len(idx) + p]) for p in range(len(idx))] for e in range(xs.shape[0])]

    def close(self):
        for h in self._signals.values():
            nat.lib.gacq_group_signal_destroy(h)
        self._signals.clear()
        if self._g:
            nat.lib.gacq_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def firwin_hann(ntaps, cutoff_norm):
    """scipy.signal.firwin(ntaps, cutoff_norm, window='hann') from the native library (host)."""
    taps = np.empty(int(ntaps), dtype=np.float64)
    nat.check(nat.lib.gacq_firwin_hann(int(ntaps), float(cutoff_norm), taps.ctypes.data_as(nat.c_double_p)))
    return taps


def descriptor(name):
    sig = _signals.get(name) if isinstance(name, str) else name
    L = nat.check(nat.lib.gacq_code_length(sig.code.encode()))
    return nat.SigDesc(L, sig.n, int(sig.pad), int(sig.boc), int(sig.normalised), int(sig.fold), sig.fs)


def finalize(name, items, peaks, dopplers, shard_d0=None):
    """Host-side last step (no GPU needed): peaks [nshard, nitems] (PEAK_DTYPE) -> [(metric, code, doppler)].
    Shards are merged in Doppler order with strict '>' (acquire-gps-l1.py:36-39), then converted
    to the reference's return tuple (acquire-gps-l1.py:38-40)."""
    nitems = len(items)
    peaks = np.ascontiguousarray(peaks).view(PEAK_DTYPE).reshape(-1, nitems)
    nshard = peaks.shape[0]
    dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
    d0 = np.ascontiguousarray(shard_d0 if shard_d0 is not None else np.zeros(nshard), dtype=np.int32)
    desc = descriptor(name)
    res = (nat.Result * nitems)()
    nat.check(nat.lib.gacq_finalize(ctypes.byref(desc), peaks.ctypes.data_as(ctypes.POINTER(nat.Peak)), nshard,
                                    d0.ctypes.data_as(nat.c_int_p), nitems,
                                    dopplers.ctypes.data_as(nat.c_double_p), len(dopplers), res))
    return _as_tuples(res)


def _as_tuple(r):
    if r.d_index < 0:
        return 0, 0, 0                  # the reference's untouched initial values (acquire-gps-l1.py:25)
    return np.float64(r.metric), float(r.code_chips), np.float64(r.doppler_hz)


def _as_tuples(res):
    """[_as_tuple(r) for r in res] for a ctypes array of gacq_result, through one structured-array view: iterating a float64 column
    yields np.float64 scalars (the reference's metric / doppler types), .tolist() plain floats (its code phase) -- a third of the time of
    32 per-record attribute reads on the single-search latency path."""
    a = res if isinstance(res, np.ndarray) else np.frombuffer(res, dtype=RESULT_DTYPE)
    out = list(zip(a["metric"], a["code_chips"].tolist(), a["doppler_hz"]))
    if a["d_index"].min() < 0:
        for k in np.flatnonzero(a["d_index"] < 0):
            out[k] = (0, 0, 0)          # the reference's untouched initial values (acquire-gps-l1.py:25)
    return out


def format_result(name, item, result):
    """The worker() output line of acquire-<name>.py (e.g. acquire-gps-l1.py:103)."""
    sig = _signals.get(name) if isinstance(name, str) else name
    metric, code, doppler = result
    return sig.fmt % (item, doppler, metric, code)


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def make_search(name, engine=None):
    """Return ``search(x, prn, doppler_search, ms)`` with the reference's signature and return tuple,
    bound to acquire-<name>.py's hard-coded fs/n/variant."""
    sig = _signals.get(name)

    def search(x, prn, doppler_search, ms):
        return (engine or default_engine()).search(sig, x, prn, doppler_search, ms)

    search.__doc__ = "GPU search() of acquire-%s.py: (metric, code, doppler)" % sig.name
    return search


def search_north_star(prn, samples, fs, doppler_range, name="gps-l1", ms=1, engine=None):
    """BASELINE.json's argument order: search(prn, samples, fs, doppler_range) -> (code_phase, doppler, metric).
    `fs` must equal the signal's internal rate (the reference hard-codes it, acquire-gps-l1.py:19)."""
    sig = _signals.get(name)
    if float(fs) != sig.fs:
        raise ValueError("%s search runs at fs=%.0f Hz (hard-coded in the reference), got %r" % (sig.name, sig.fs, fs))
    metric, code, doppler = (engine or default_engine()).search(sig, samples, prn, doppler_range, ms)
    return code, doppler, metric
