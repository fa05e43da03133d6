"""ctypes binding of the C ABI (include/termdaw_amd.h -> termdaw_amd/lib/libtermdaw_amd.so).

The classes keep the reference's names and method meanings (SampleBank sample.rs:187-348, FlowwBank
floww.rs:6-141, Graph graph.rs:12-238, State state.rs:27-578) so that parity tests read like tests of
the reference.  Everything renders on the GPU through the HIP library; there is no CPU fallback --
a missing library or missing device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtermdaw_amd.so")
_LIB_OVERRIDE = os.environ.get("TD_LIB")   # A/B experiments only (tools/ab_lib.py)


class TermdawError(RuntimeError):
    pass


class td_event(C.Structure):
    _fields_ = [("t_sec", C.c_float), ("note", C.c_float), ("vel", C.c_float)]


_lib = None

# name -> (restype, argtypes); also used by the symbol-export test
_f32, _sz, _vp, _cp, _i32, _lng = C.c_float, C.c_size_t, C.c_void_p, C.c_char_p, C.c_int, C.c_long
_fp = C.POINTER(C.c_float)
SIGNATURES = {
    "td_last_error": (_cp, []),
    "td_device_count": (_i32, []),
    "td_set_device": (_i32, [_i32]),
    "td_samplebank_new": (_vp, [_sz]),
    "td_samplebank_free": (None, [_vp]),
    "td_samplebank_add_file": (_i32, [_vp, _cp, _cp, _cp]),
    "td_samplebank_add_decoded": (_i32, [_vp, _cp, _fp, _sz, _i32, _sz, _sz, _cp]),
    "td_samplebank_get_index": (_lng, [_vp, _cp]),
    "td_samplebank_sample_len": (_sz, [_vp, _sz]),
    "td_samplebank_read": (_i32, [_vp, _sz, _fp, _fp]),
    "td_samplebank_get_max_sr_bd": (None, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "td_flowwbank_new": (_vp, [_sz, _sz]),
    "td_flowwbank_free": (None, [_vp]),
    "td_flowwbank_reset": (None, [_vp]),
    "td_flowwbank_add_events": (_lng, [_vp, _cp, C.POINTER(td_event), _sz]),
    "td_flowwbank_add_midi": (_lng, [_vp, _cp, _cp]),
    "td_flowwbank_declare_stream": (_lng, [_vp, _cp]),
    "td_flowwbank_append_stream": (_lng, [_vp, _cp, C.POINTER(td_event), _sz]),
    "td_flowwbank_trim_streams": (None, [_vp]),
    "td_flowwbank_get_events": (_sz, [_vp, _sz, C.POINTER(td_event), _sz]),
    "td_flowwbank_get_index": (_lng, [_vp, _cp]),
    "td_flowwbank_set_time": (None, [_vp, _sz]),
    "td_flowwbank_set_time_to_next_block": (None, [_vp]),
    "td_graph_new": (_vp, [_sz, _sz]),
    "td_graph_free": (None, [_vp]),
    "td_graph_reset": (None, [_vp]),
    "td_graph_add_sum": (_i32, [_vp, _cp, _f32, _f32]),
    "td_graph_add_normalize": (_i32, [_vp, _cp, _f32, _f32]),
    "td_graph_add_sampleloop": (_i32, [_vp, _cp, _f32, _f32, _sz]),
    "td_graph_add_sample_multi": (_i32, [_vp, _cp, _f32, _f32, _sz, _sz, _i32]),
    "td_graph_add_sample_lerp": (_i32, [_vp, _cp, _f32, _f32, _sz, _sz, _i32, _i32]),
    "td_graph_add_debug_sine": (_i32, [_vp, _cp, _f32, _f32, _sz]),
    "td_graph_add_synth": (_i32, [_vp, _cp, _f32, _f32, _sz, _f32, _f32, _fp, _i32, _f32, _f32, _fp, _i32, _f32, _fp, _i32]),
    "td_graph_add_adsr": (_i32, [_vp, _cp, _f32, _f32, _f32, _sz, _i32, _i32, _i32, _fp, _i32]),
    "td_graph_add_sampsyn": (_i32, [_vp, _cp, _f32, _f32, _sz, _fp, _i32, _cp, _sz]),
    "td_graph_add_bandpass": (_i32, [_vp, _cp, _f32, _f32, _f32, _f32, _f32, _i32]),
    "td_graph_connect": (_i32, [_vp, _cp, _cp]),
    "td_graph_set_output": (_i32, [_vp, _cp]),
    "td_graph_check": (_i32, [_vp]),
    "td_graph_set_time": (None, [_vp, _sz]),
    "td_graph_change_time": (_sz, [_vp, _sz, _i32]),
    "td_graph_get_time": (_sz, [_vp]),
    "td_graph_reset_normalize_vertices": (None, [_vp]),
    "td_graph_get_normalization_value": (_f32, [_vp, _cp]),
    "td_graph_vertex_count": (_sz, [_vp]),
    "td_graph_render_block": (_i32, [_vp, _vp, _vp, _fp, _fp]),
    "td_graph_normalize_scan": (_i32, [_vp, _vp, _vp, _sz]),
    "td_graph_render_all": (_sz, [_vp, _vp, _vp, _sz, _i32]),
    "td_graph_render_all_resampled": (_sz, [_vp, _vp, _vp, _sz, _i32, _sz, _sz]),
    "td_graph_output_pcm_device": (_vp, [_vp]),
    "td_graph_output_f32_device": (_vp, [_vp]),
    "td_graph_read_pcm": (_i32, [_vp, _vp, _sz]),
    "td_graph_read_f32": (_i32, [_vp, _fp, _sz]),
    "td_graph_output_peak": (_f32, [_vp]),
    "td_graph_render_all_async": (_sz, [_vp, _vp, _vp, _sz, _i32]),
    "td_graph_sync": (_i32, [_vp]),
    "td_graph_norm_fix_runs": (_sz, [_vp]),
    "td_graph_set_profiling": (None, [_vp, _i32]),
    "td_graph_host_times": (_sz, [_vp, C.POINTER(C.c_double), _i32]),
    "td_graph_last_kernel_times": (_sz, [_vp, C.POINTER(_cp), _fp, C.POINTER(_sz), _sz]),
    "td_graph_device_bytes": (_sz, [_vp]),
    "td_trim_memory": (None, []),
    "td_device_sinf": (_i32, [_fp, _fp, _sz, _i32]),
    "td_cached_memory_bytes": (_sz, []),
    "td_graph_set_option": (_i32, [_vp, _cp, _lng]),
    "td_graph_get_option": (_i32, [_vp, _cp, C.POINTER(_lng)]),
    "td_graph_option_key": (_cp, [_sz]),
    "td_graph_band_stats": (_i32, [_vp, C.POINTER(C.c_uint32)]),
    "td_graph_band_guard_stats": (_i32, [_vp, C.POINTER(C.c_double)]),
    "td_batch_new": (_vp, []),
    "td_batch_free": (None, [_vp]),
    "td_batch_add": (_lng, [_vp, _vp, _vp, _vp]),
    "td_batch_size": (_sz, [_vp]),
    "td_batch_rewind": (None, [_vp]),
    "td_batch_render_all": (_sz, [_vp, _sz, _i32]),
    "td_batch_render_all_async": (_sz, [_vp, _sz, _i32]),
    "td_batch_sync": (_i32, [_vp]),
    "td_batch_normalize_scan": (_i32, [_vp, _sz]),
    "td_batch_render_to_files": (_i32, [_vp, _sz, _i32, _sz, C.POINTER(_cp), _i32, _i32, C.POINTER(C.c_double)]),
    "td_batch_host_pcm": (_vp, [_vp, _sz, C.POINTER(_sz)]),
    "td_batch_peaks": (_i32, [_vp, _fp]),
    "td_batch_peak_table_device": (_i32, [_vp, _vp, _sz, _sz, _sz]),
    "td_comm_unique_id": (_i32, [_vp, _sz]),
    "td_comm_init": (_vp, [_vp, _sz, _i32, _i32]),
    "td_comm_init_host": (_vp, [_vp, _vp, _i32, _i32]),
    "td_comm_free": (None, [_vp]),
    "td_comm_backend": (_cp, [_vp]),
    "td_comm_library": (_cp, []),
    "td_batch_exchange_peaks": (_i32, [_vp, _vp, _sz]),
    "td_batch_peak_table": (_vp, [_vp, C.POINTER(_sz)]),
    "td_batch_read_peak_table": (_i32, [_vp, _fp, _sz]),
    "td_batch_set_profiling": (None, [_vp, _i32]),
    "td_batch_last_kernel_times": (_sz, [_vp, C.POINTER(_cp), _fp, C.POINTER(_sz), _sz]),
    "td_batch_host_times": (_sz, [_vp, C.POINTER(C.c_double), _i32]),
    "td_batch_mark": (_i32, [_vp, _i32]),
    "td_batch_marked_ms": (C.c_double, [_vp]),
    "td_state_new": (_vp, [_cp, _sz, _sz]),
    "td_state_open": (_vp, [_cp]),
    "td_state_free": (None, [_vp]),
    "td_state_set_option": (_i32, [_vp, _cp, _lng]),
    "td_state_refresh_source": (_i32, [_vp, _cp]),
    "td_state_refresh": (_i32, [_vp]),
    "td_state_scan_exact": (_i32, [_vp]),
    "td_state_render": (_i32, [_vp, _cp]),
    "td_state_render_to_memory": (_sz, [_vp, _vp, _sz]),
    "td_state_render_view": (_vp, [_vp, C.POINTER(_sz)]),
    "td_state_chunk_count": (_sz, [_vp]),
    "td_state_render_samplerate": (_sz, [_vp]),
    "td_state_bitdepth": (_sz, [_vp]),
    "td_state_output_file": (_cp, [_vp]),
    "td_state_buffer_length": (_sz, [_vp]),
    "td_state_project_samplerate": (_sz, [_vp]),
    "td_state_graph": (_vp, [_vp]),
    "td_state_samplebank": (_vp, [_vp]),
    "td_state_flowwbank": (_vp, [_vp]),
    "td_state_dump_calls": (_cp, [_vp]),
}


def build(force=False):
    """hipcc build of the in-tree HIP library (termdaw_amd/Makefile); rebuilds when a source is newer."""
    import glob
    import subprocess
    srcs = glob.glob(os.path.join(_HERE, "csrc", "*")) + [os.path.join(_HERE, "..", "include", "termdaw_amd.h"),
                                                         os.path.join(_HERE, "Makefile")]
    def stale():
        return (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH)
                                                     for s in srcs if os.path.exists(s))
    if force or stale():
        # one builder at a time: the ranks of a multi-GPU launch all come through here at once
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if force or stale():
                    subprocess.check_call(["make", "-C", _HERE, "-j8", "-s"])
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def lib():
    """Loads the HIP library, building it first if the in-tree .so is missing or older than its sources.
    There is no CPU fallback: if hipcc cannot build it, this raises."""
    global _lib
    if _lib is None:
        # A stale or missing library whose rebuild fails is an error: tests and bench must never run a binary
        # that is older than the sources they claim to measure.  Only an explicit opt-in skips the build:
        # TD_LIB=<path> (A/B experiments) or TD_NO_BUILD=1 (a box without hipcc, prebuilt .so shipped along).
        if not (_LIB_OVERRIDE or os.environ.get("TD_NO_BUILD") == "1"):
            try:
                build()
            except Exception as e:   # noqa: BLE001
                raise TermdawError("`make -C termdaw_amd` failed (%s) and %s is missing or older than its sources; "
                                   "there is no CPU fallback (TD_NO_BUILD=1 loads a prebuilt library as it is)"
                                   % (e, LIB_PATH))
        if not os.path.exists(_LIB_OVERRIDE or LIB_PATH):
            raise TermdawError("%s is missing; there is no CPU fallback" % (_LIB_OVERRIDE or LIB_PATH))
        L = C.CDLL(_LIB_OVERRIDE or LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().td_last_error().decode(errors="replace")


def _check(ok):
    if not ok:
        raise TermdawError(last_error())
    return ok


def _fa(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def device_count():
    return lib().td_device_count()


def set_device(i):
    _check(lib().td_set_device(i))


class SampleBank:
    def __init__(self, sample_rate, _handle=None):
        self.h = _handle or lib().td_samplebank_new(sample_rate)
        self._own = _handle is None
        self.sample_rate = sample_rate

    def __del__(self):
        if getattr(self, "h", None) and self._own:
            lib().td_samplebank_free(self.h)
            self.h = None

    def add_decoded(self, name, values, channels, sr, bits, method=""):
        a, p = _fa(values)
        _check(lib().td_samplebank_add_decoded(self.h, name.encode(), p, a.size, channels, sr, bits, method.encode()))

    def add(self, name, file, method=""):
        _check(lib().td_samplebank_add_file(self.h, name.encode(), file.encode(), method.encode()))

    def get_index(self, name):
        i = lib().td_samplebank_get_index(self.h, name.encode())
        return None if i < 0 else i

    def get_sample(self, index):
        n = lib().td_samplebank_sample_len(self.h, index)
        l = np.empty(n, np.float32)
        r = np.empty(n, np.float32)
        _check(lib().td_samplebank_read(self.h, index, l.ctypes.data_as(_fp), r.ctypes.data_as(_fp)))
        return l, r

    def get_max_sr_bd(self):
        a, b = _sz(), _sz()
        lib().td_samplebank_get_max_sr_bd(self.h, C.byref(a), C.byref(b))
        return a.value, b.value


class FlowwBank:
    def __init__(self, sr, bl, _handle=None):
        self.h = _handle or lib().td_flowwbank_new(sr, bl)
        self._own = _handle is None

    def __del__(self):
        if getattr(self, "h", None) and self._own:
            lib().td_flowwbank_free(self.h)
            self.h = None

    def add_events(self, name, events):
        ev = np.ascontiguousarray(np.asarray(events, dtype=np.float32).reshape(-1, 3))
        return lib().td_flowwbank_add_events(self.h, name.encode(), ev.ctypes.data_as(C.POINTER(td_event)), ev.shape[0])

    def add_midi(self, name, path):
        """FlowwBank::add_floww (floww.rs:40-48); raises TermdawError like the reference's Err."""
        i = lib().td_flowwbank_add_midi(self.h, name.encode(), str(path).encode())
        if i < 0:
            raise TermdawError(last_error())
        return i

    def declare_stream(self, name):
        return lib().td_flowwbank_declare_stream(self.h, name.encode())

    def append_stream(self, name, events):
        ev = np.ascontiguousarray(np.asarray(events, dtype=np.float32).reshape(-1, 3))
        return lib().td_flowwbank_append_stream(self.h, name.encode(), ev.ctypes.data_as(C.POINTER(td_event)), ev.shape[0])

    def trim_streams(self):
        lib().td_flowwbank_trim_streams(self.h)

    def get_events(self, index):
        n = lib().td_flowwbank_get_events(self.h, index, None, 0)
        buf = np.zeros((max(n, 1), 3), np.float32)
        lib().td_flowwbank_get_events(self.h, index, buf.ctypes.data_as(C.POINTER(td_event)), n)
        return buf[:n]

    def get_index(self, name):
        i = lib().td_flowwbank_get_index(self.h, name.encode())
        return None if i < 0 else i

    def set_time(self, t):
        lib().td_flowwbank_set_time(self.h, t)

    def set_time_to_next_block(self):
        lib().td_flowwbank_set_time_to_next_block(self.h)


class Graph:
    def __init__(self, bl, sr, _handle=None):
        self.h = _handle or lib().td_graph_new(bl, sr)
        self._own = _handle is None
        self.bl = bl
        self.sr = sr

    def __del__(self):
        if getattr(self, "h", None) and self._own:
            lib().td_graph_free(self.h)
            self.h = None

    def add_sum(self, name, gain, angle):
        _check(lib().td_graph_add_sum(self.h, name.encode(), gain, angle))

    def add_normalize(self, name, gain, angle):
        _check(lib().td_graph_add_normalize(self.h, name.encode(), gain, angle))

    def add_sampleloop(self, name, gain, angle, sample_index):
        _check(lib().td_graph_add_sampleloop(self.h, name.encode(), gain, angle, sample_index))

    def add_sample_multi(self, name, gain, angle, sample_index, floww_index, note):
        _check(lib().td_graph_add_sample_multi(self.h, name.encode(), gain, angle, sample_index, floww_index, note))

    def add_sample_lerp(self, name, gain, angle, sample_index, floww_index, note, lerp_len):
        _check(lib().td_graph_add_sample_lerp(self.h, name.encode(), gain, angle, sample_index, floww_index, note, lerp_len))

    def add_debug_sine(self, name, gain, angle, floww_index):
        _check(lib().td_graph_add_debug_sine(self.h, name.encode(), gain, angle, floww_index))

    def add_synth(self, name, gain, angle, floww_index, sq_vel, sq_z, sq_adsr, tf_vel, tf_z, tf_adsr, tr_vel, tr_adsr):
        a1, p1 = _fa(sq_adsr)
        a2, p2 = _fa(tf_adsr)
        a3, p3 = _fa(tr_adsr)
        _check(lib().td_graph_add_synth(self.h, name.encode(), gain, angle, floww_index, sq_vel, sq_z, p1, a1.size,
                                        tf_vel, tf_z, p2, a2.size, tr_vel, p3, a3.size))

    def add_sampsyn(self, name, gain, angle, floww_index, adsr, table_bytes):
        a, p = _fa(adsr)
        tb = bytes(table_bytes) if table_bytes is not None else None
        _check(lib().td_graph_add_sampsyn(self.h, name.encode(), gain, angle, floww_index, p, a.size, tb, len(tb) if tb else 0))

    def add_adsr(self, name, gain, angle, wet, floww_index, use_off, use_max, note, adsr):
        a, p = _fa(adsr)
        _check(lib().td_graph_add_adsr(self.h, name.encode(), gain, angle, wet, floww_index, int(use_off), int(use_max),
                                       note, p, a.size))

    def add_bandpass(self, name, gain, angle, wet, lo_hz, hi_hz, pass_):
        _check(lib().td_graph_add_bandpass(self.h, name.encode(), gain, angle, wet, lo_hz, hi_hz, int(pass_)))

    def connect(self, a, b):
        return bool(lib().td_graph_connect(self.h, a.encode(), b.encode()))

    def set_output(self, name):
        return bool(lib().td_graph_set_output(self.h, name.encode()))

    def check_graph(self):
        return bool(lib().td_graph_check(self.h))

    def set_time(self, t):
        lib().td_graph_set_time(self.h, t)

    def change_time(self, delta, plus):
        return lib().td_graph_change_time(self.h, delta, int(plus))

    def get_time(self):
        return lib().td_graph_get_time(self.h)

    def reset_normalize_vertices(self):
        lib().td_graph_reset_normalize_vertices(self.h)

    def get_normalization_value(self, name):
        return lib().td_graph_get_normalization_value(self.h, name.encode())

    def render(self, sb, fb):
        """Graph::render: one block at the playhead -> (l, r) or None."""
        l = np.empty(self.bl, np.float32)
        r = np.empty(self.bl, np.float32)
        ok = lib().td_graph_render_block(self.h, sb.h, fb.h, l.ctypes.data_as(_fp), r.ctypes.data_as(_fp))
        if ok < 0:
            raise TermdawError(last_error())
        if not ok:
            return None
        return l, r

    def true_normalize_scan(self, sb, fb, chunks):
        _check(lib().td_graph_normalize_scan(self.h, sb.h, fb.h, chunks))

    def render_all(self, sb, fb, cs, bd=16, want_f32=True, want_pcm=True):
        """State::render loop on the GPU. Returns (pcm[frames,2], f32[frames,2])."""
        n = lib().td_graph_render_all(self.h, sb.h, fb.h, cs, bd)
        if cs and not n:
            raise TermdawError(last_error())
        frames = cs * self.bl
        pcm = f = None
        if want_pcm:
            pcm = np.zeros((frames, 2), np.int32 if bd > 16 else np.int16)
            if frames:
                _check(lib().td_graph_read_pcm(self.h, pcm.ctypes.data_as(_vp), pcm.nbytes))
        if want_f32:
            f = np.zeros((frames, 2), np.float32)
            if frames:
                _check(lib().td_graph_read_f32(self.h, f.ctypes.data_as(_fp), f.size))
        return pcm, f

    def render_all_resampled(self, sb, fb, cs, bd, psr, render_sr):
        """State::render, psr > render_sr arm (build-defined resampler). Returns (pcm, f32)."""
        n = lib().td_graph_render_all_resampled(self.h, sb.h, fb.h, cs, bd, psr, render_sr)
        if cs and not n:
            raise TermdawError(last_error())
        pcm = np.zeros((n, 2), np.int32 if bd > 16 else np.int16)
        f = np.zeros((n, 2), np.float32)
        if n:
            _check(lib().td_graph_read_pcm(self.h, pcm.ctypes.data_as(_vp), pcm.nbytes))
            _check(lib().td_graph_read_f32(self.h, f.ctypes.data_as(_fp), f.size))
        return pcm, f

    # -- bench hooks --
    def render_all_async(self, sb, fb, cs, bd=16):
        n = lib().td_graph_render_all_async(self.h, sb.h, fb.h, cs, bd)
        if cs and not n:
            raise TermdawError(last_error())
        return n

    def sync(self):
        _check(lib().td_graph_sync(self.h))

    def norm_fix_runs(self):
        """How often a single-pass Normalize launch was redone by the check kernel after the fact (0 in normal operation)."""
        return lib().td_graph_norm_fix_runs(self.h)

    def host_times(self, reset=True):
        """Host ms per phase since the last reset: compile, descriptors, upload, launches; and chunk count."""
        out = (C.c_double * 4)()
        n = lib().td_graph_host_times(self.h, out, int(reset))
        return {"compile": out[0], "descriptors": out[1], "upload": out[2], "launch": out[3], "chunks": n}

    def set_profiling(self, on):
        lib().td_graph_set_profiling(self.h, int(on))

    def kernel_times(self):
        cap = 32
        names = (_cp * cap)()
        ms = (C.c_float * cap)()
        cnt = (_sz * cap)()
        n = lib().td_graph_last_kernel_times(self.h, names, ms, cnt, cap)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}

    def output_peak(self):
        return lib().td_graph_output_peak(self.h)

    def device_bytes(self):
        return lib().td_graph_device_bytes(self.h)

    def band_stats(self):
        out = (C.c_uint32 * 3)()
        _check(lib().td_graph_band_stats(self.h, out))
        return {"mismatched": out[0], "recomputed": out[1], "parked": out[2]}

    def band_guard_stats(self):
        """band_mode 2: renders audited, renders done again with the exact kernels, last / largest estimated RMS deviation."""
        out = (C.c_double * 4)()
        _check(lib().td_graph_band_guard_stats(self.h, out))
        return {"audits": int(out[0]), "redos": int(out[1]), "last_est": out[2], "max_est": out[3]}

    def set_option(self, key, value):
        _check(lib().td_graph_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = _lng(0)
        _check(lib().td_graph_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)


COMM_ID_BYTES = 128
_HOST_ALLREDUCE = C.CFUNCTYPE(_i32, _vp, _fp, _sz)


def comm_unique_id():
    """rank 0: the 128-byte id every rank hands to Comm (ncclGetUniqueId behind td_comm_unique_id)."""
    buf = (C.c_char * COMM_ID_BYTES)()
    _check(lib().td_comm_unique_id(buf, COMM_ID_BYTES))
    return bytes(buf)


class Comm:
    """The ranks of a multi-GPU job (td_comm): `Comm(unique_id, rank, world)` joins over RCCL on the current device;
    `Comm.over_host(fn, rank, world)` runs the same exchange over the host's own all-reduce -- fn(table: np.float32[n]) replaces
    the table in place by its element-wise maximum over the ranks (MPI hosts; two ranks on one GPU in tests)."""

    def __init__(self, unique_id=None, rank=0, world=1, _h=None, _keep=None):
        self._keep = _keep
        if _h is not None:
            self.h = _h
        else:
            if unique_id is None or len(unique_id) < COMM_ID_BYTES:
                raise ValueError("Comm needs the 128-byte id of comm_unique_id()")
            self.h = lib().td_comm_init(bytes(unique_id), len(unique_id), rank, world)
        if not self.h:
            raise TermdawError(last_error())
        self.rank, self.world = rank, world

    @classmethod
    def over_host(cls, fn, rank, world):
        def tramp(_ctx, table, n):
            try:
                fn(np.ctypeslib.as_array(table, shape=(n,)))
                return 1
            except Exception:   # (an exception must not cross the C ABI: the entry reports a failed exchange)
                return 0
        cb = _HOST_ALLREDUCE(tramp)
        h = lib().td_comm_init_host(C.cast(cb, _vp), None, rank, world)
        return cls(rank=rank, world=world, _h=h, _keep=cb)

    def backend(self):
        return lib().td_comm_backend(self.h).decode()

    def __del__(self):
        if getattr(self, "h", None):
            lib().td_comm_free(self.h)
            self.h = None


def comm_library():
    return lib().td_comm_library().decode()


class Batch:
    """Many independent projects rendered together on one GPU (BASELINE config 5): State::render's loop
    (state.rs:563-575) for every project, same-kind launches of different projects merged into one grid."""

    def __init__(self):
        self.h = lib().td_batch_new()
        self.projects = []   # (sb, fb, g) kept alive: the batch only borrows the handles

    def __del__(self):
        if getattr(self, "h", None):
            lib().td_batch_free(self.h)
            self.h = None

    def add(self, sb, fb, g):
        i = lib().td_batch_add(self.h, g.h, sb.h, fb.h)
        if i < 0:
            raise TermdawError(last_error())
        self.projects.append((sb, fb, g))
        return i

    def __len__(self):
        return lib().td_batch_size(self.h)

    def rewind(self):
        lib().td_batch_rewind(self.h)

    def render_all(self, cs, bd=16):
        n = lib().td_batch_render_all(self.h, cs, bd)
        if cs and len(self) and not n:
            raise TermdawError(last_error())
        return n

    def render_all_async(self, cs, bd=16):
        n = lib().td_batch_render_all_async(self.h, cs, bd)
        if cs and len(self) and not n:
            raise TermdawError(last_error())
        return n

    def sync(self):
        _check(lib().td_batch_sync(self.h))

    E2E_KEYS = ("wall_ms", "setup_ms", "gpu_render_span_ms", "copy_span_ms", "copy_busy_ms", "bytes", "write_span_ms", "enqueue_ms")

    def render_to_files(self, cs, bd=16, render_sr=48000, paths=None, group=4, writers=8):
        """State::render end to end for every project: render -> page-locked host PCM -> WAV files, pipelined.  paths None:
        no files (render + D2H).  Returns the timing report (E2E_KEYS)."""
        arr = None
        if paths is not None:
            arr = (_cp * len(paths))(*[p.encode() for p in paths])
        times = (C.c_double * 8)()
        _check(lib().td_batch_render_to_files(self.h, cs, bd, render_sr, arr, group, writers if paths is not None else 0, times))
        return dict(zip(self.E2E_KEYS, [float(x) for x in times]))

    def host_pcm(self, i, bd=16):
        """Project i's PCM of the last render_to_files, copied out of the library's page-locked buffer."""
        n = _sz(0)
        p = lib().td_batch_host_pcm(self.h, i, C.byref(n))
        if not p:
            raise TermdawError("no host PCM")
        raw = C.string_at(p, n.value)
        return np.frombuffer(raw, dtype=np.int32 if bd > 16 else np.int16).reshape(-1, 2).copy()

    def normalize_scan(self, chunks):
        _check(lib().td_batch_normalize_scan(self.h, chunks))

    def peaks(self):
        out = np.zeros(max(len(self), 1), np.float32)
        _check(lib().td_batch_peaks(self.h, out.ctypes.data_as(_fp)))
        return out[:len(self)]

    def exchange_peaks(self, comm, per_rank):
        """The job's one collective (td_batch_exchange_peaks): this rank's peaks into the per_rank x world table, then ONE
        all-reduce(max) on the batch's stream -- enqueued, not waited for (sync() does).  comm None: a job of one rank."""
        _check(lib().td_batch_exchange_peaks(self.h, comm.h if comm is not None else None, per_rank))

    def peak_table(self, n=None):
        """The exchanged table, copied to the host (synchronises)."""
        cnt = _sz(0)
        lib().td_batch_peak_table(self.h, C.byref(cnt))
        n = cnt.value if n is None else n
        out = np.zeros(max(n, 1), np.float32)
        _check(lib().td_batch_read_peak_table(self.h, out.ctypes.data_as(_fp), n))
        return out[:n]

    def peak_table_device(self, device_ptr, n_total, first=0, stride=1):
        """Fills n_total floats at `device_ptr` (e.g. torch_tensor.data_ptr()): own entries at first + i*stride, 0 elsewhere."""
        _check(lib().td_batch_peak_table_device(self.h, C.c_void_p(device_ptr), n_total, first, stride))

    def read_pcm(self, i, cs, bd=16):
        sb, fb, g = self.projects[i]
        pcm = np.zeros((cs * g.bl, 2), np.int32 if bd > 16 else np.int16)
        if pcm.size:
            _check(lib().td_graph_read_pcm(g.h, pcm.ctypes.data_as(_vp), pcm.nbytes))
        return pcm

    def set_profiling(self, on):
        lib().td_batch_set_profiling(self.h, int(on))

    def kernel_times(self):
        cap = 32
        names = (_cp * cap)()
        ms = (C.c_float * cap)()
        cnt = (_sz * cap)()
        n = lib().td_batch_last_kernel_times(self.h, names, ms, cnt, cap)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}

    def mark(self, which):
        """A mark on the batch's stream: 0 before a run of submissions, 1 behind it (marked_ms: device time between them)."""
        _check(lib().td_batch_mark(self.h, int(which)))

    def marked_ms(self):
        return float(lib().td_batch_marked_ms(self.h))

    def host_times(self, reset=True):
        out = (C.c_double * 4)()
        n = lib().td_batch_host_times(self.h, out, int(reset))
        return {"compile": out[0], "descriptors": out[1], "upload": out[2], "launch": out[3], "steps": n}


class State:
    """State (state.rs:27-578): project script -> banks + graph -> render to WAV."""

    def __init__(self, wdir="", project_samplerate=44100, buffer_length=1024, open_dir=None):
        if open_dir is not None:
            self.h = lib().td_state_open(open_dir.encode())
        else:
            self.h = lib().td_state_new(wdir.encode(), project_samplerate, buffer_length)
        if not self.h:
            raise TermdawError(last_error())

    def __del__(self):
        if getattr(self, "h", None):
            lib().td_state_free(self.h)
            self.h = None

    def refresh(self, source=None):
        ok = lib().td_state_refresh(self.h) if source is None else lib().td_state_refresh_source(self.h, source.encode())
        return bool(ok)

    def set_option(self, key, value):
        """Engine option of the State's graph; a State defaults to band_mode 2 (the scan under the guard), set_option("band_mode", 0) = exact."""
        _check(lib().td_state_set_option(self.h, key.encode(), int(value)))

    def scan_exact(self):
        _check(lib().td_state_scan_exact(self.h))

    def render(self, path=None):
        _check(lib().td_state_render(self.h, path.encode() if path else None))

    def render_to_memory(self):
        nbytes = lib().td_state_render_to_memory(self.h, None, 0)
        bd = self.bd
        out = np.zeros(nbytes // (4 if bd > 16 else 2), np.int32 if bd > 16 else np.int16)
        if nbytes:
            if not lib().td_state_render_to_memory(self.h, out.ctypes.data_as(_vp), out.nbytes):
                raise TermdawError(last_error())
        return out.reshape(-1, 2)

    def render_view(self):
        """Render and return the PCM as a read-only numpy view of the library's page-locked read-back buffer
        (frames x 2; valid until the next render of this State)."""
        n = _sz(0)
        p = lib().td_state_render_view(self.h, C.byref(n))
        if not p:
            raise TermdawError(last_error())
        dt = np.int32 if self.bd > 16 else np.int16
        if n.value == 0:
            return np.zeros((0, 2), dt)
        buf = (C.c_uint8 * n.value).from_address(p)
        out = np.frombuffer(buf, dtype=dt).reshape(-1, 2)
        out.flags.writeable = False
        return out

    @property
    def cs(self):
        return lib().td_state_chunk_count(self.h)

    @property
    def render_sr(self):
        return lib().td_state_render_samplerate(self.h)

    @property
    def bd(self):
        return lib().td_state_bitdepth(self.h)

    @property
    def output_file(self):
        return lib().td_state_output_file(self.h).decode()

    def dump_calls(self):
        return lib().td_state_dump_calls(self.h).decode()

    @property
    def g(self):
        return Graph(lib().td_state_buffer_length(self.h), lib().td_state_project_samplerate(self.h),
                     _handle=lib().td_state_graph(self.h))

    @property
    def sb(self):
        return SampleBank(lib().td_state_project_samplerate(self.h), _handle=lib().td_state_samplebank(self.h))

    @property
    def fb(self):
        return FlowwBank(lib().td_state_project_samplerate(self.h), lib().td_state_buffer_length(self.h),
                         _handle=lib().td_state_flowwbank(self.h))
