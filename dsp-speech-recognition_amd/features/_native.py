"""ctypes binding of libdsp_frontend.so (include/dsp_frontend.h) -- the only way this package
computes anything.  There is deliberately NO CPU fallback: if the library or a GPU is missing,
every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('DSP_FRONTEND_LIB', os.path.join(_PKG, 'lib', 'libdsp_frontend.so'))

OK, EINVAL, EHIP, ENODEV = 0, -1, -2, -3
WAVE_F32, WAVE_I16 = 0, 1
OUT_FRAMES, OUT_MAGSPEC, OUT_POWSPEC, OUT_FBANK, OUT_MFCC = 0, 1, 2, 3, 4

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class PlanDesc(C.Structure):
    _fields_ = [
        ('frame_len', c_i32), ('frame_step', c_i32), ('nfft', c_i32), ('nfilt', c_i32),
        ('numcep', c_i32), ('append_energy', c_i32), ('preemph', c_f32),
        ('h_window', c_vp), ('h_mel_start', c_vp), ('h_mel_count', c_vp), ('h_mel_weights', c_vp),
        ('h_dct', c_vp),
    ]


# name -> (restype, argtypes); mirrors include/dsp_frontend.h one to one.
SIGNATURES = {
    'dsp_abi_version': (C.c_int, []),
    'dsp_last_error': (C.c_char_p, []),
    'dsp_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'dsp_set_device': (C.c_int, [C.c_int]),
    'dsp_get_device': (C.c_int, [c_vp]),
    'dsp_malloc': (C.c_int, [C.POINTER(c_vp), C.c_size_t]),
    'dsp_free': (C.c_int, [c_vp]),
    'dsp_memcpy_h2d': (C.c_int, [c_vp, c_vp, C.c_size_t, c_vp]),
    'dsp_memcpy_d2h': (C.c_int, [c_vp, c_vp, C.c_size_t, c_vp]),
    'dsp_memset': (C.c_int, [c_vp, C.c_int, C.c_size_t, c_vp]),
    'dsp_stream_synchronize': (C.c_int, [c_vp]),
    'dsp_frame_count': (C.c_int, [c_i64, c_i32, c_i32, C.POINTER(c_i64)]),
    'dsp_frame_offsets': (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    'dsp_plan_create': (C.c_int, [C.POINTER(PlanDesc), C.POINTER(c_vp)]),
    'dsp_plan_destroy': (C.c_int, [c_vp]),
    'dsp_plan_has_fast_path': (C.c_int, [c_vp]),
    'dsp_debug_force_generic': (C.c_int, [C.c_int]),
    'dsp_debug_pool_stats': (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'dsp_debug_use_mfma512': (C.c_int, [C.c_int]),
    'dsp_debug_host_dry_run': (C.c_int, [C.c_int]),
    'dsp_plan_has_mfma512': (C.c_int, [c_vp]),
    'dsp_preemphasis_batch': (C.c_int, [c_vp, C.c_int, c_vp, c_i32, c_i64, c_f32, c_vp, c_vp]),
    'dsp_features_batch': (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_i32, c_i64, c_i64, C.c_int,
                                     c_vp, c_i64, c_vp, c_vp]),
    'dsp_delta_batch': (C.c_int, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64,
                                  c_vp, c_i64, c_vp]),
    'dsp_mfcc_delta_batch': (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_i32, c_i64, c_i64, c_i32,
                                       c_vp, c_vp]),
    'dsp_scale_columns': (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    'dsp_vad_features_batch': (C.c_int, [c_vp, C.c_int, c_vp, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32,
                                         c_i32, c_vp, c_vp, c_vp]),
    'dsp_segments_workspace_bytes': (C.c_int, [c_vp, c_i32, c_i64, C.POINTER(C.c_size_t)]),
    'dsp_mfcc_delta_segments_batch': (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32,
                                                c_vp, C.c_size_t, c_vp, c_vp]),
    'dsp_endpoint_layout_segments_batch': (C.c_int, [c_vp, c_vp, c_i32, c_f64, c_f64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                     c_i64, c_vp, C.c_size_t, c_vp]),
    'dsp_layout_create': (C.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, C.POINTER(c_vp)]),
    'dsp_layout_destroy': (C.c_int, [c_vp]),
    'dsp_vad_features_layout_batch': (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    'dsp_trim_scale_batch': (C.c_int, [c_vp, C.c_int, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    'dsp_endpoint_rule_batch': (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_f64, c_f64, c_vp, c_vp]),
    'dsp_acr_gate_batch': (C.c_int, [c_vp, C.c_int, c_vp, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_f64, c_vp, c_vp]),
    'dsp_endpoint_rule_acr_batch': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f64, c_f64, c_vp, c_vp]),
    'dsp_endpoint_layout_batch': (C.c_int, [c_vp, c_vp, c_i32, c_f64, c_f64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dsp_model_timefeat_batch': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'dsp_pitch_scores_batch': (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp, c_i32,
                                         c_i32, c_i32, c_vp, c_vp]),
    'dsp_pitch_track_batch': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'dsp_pitch_rows_batch': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'dsp_resample_layout_batch': (C.c_int, [c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'dsp_decimate_batch': (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp]),
    'dsp_model_pitchfeat_batch': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    'dsp_model_finalize_batch': (C.c_int, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'dsp_model_finalize_segments_batch': (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
}

_lib = None
_lock = threading.Lock()


class DspError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``) but link
    to it under the unversioned name, so the loader does not recognise a system runtime that is
    already mapped: a process that loads this library first and imports torch later would end up
    with two HIP runtimes, and the second one finds no GPU ("No HIP GPUs are available").  When
    torch is installed but not imported yet, map ITS runtime first (without importing torch); our
    library's ``libamdhip64.so.7`` dependency then binds to it by SONAME, which is also what
    happens when torch is imported first.  ``DSP_HIP_RUNTIME=system`` keeps the system runtime.
    """
    if 'torch' in sys.modules or os.environ.get('DSP_HIP_RUNTIME', 'auto') == 'system':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass            # fall back to whatever the dynamic loader finds


def load():
    """Load the shared library (no GPU needed for this step) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DspError(
                f'{LIB_PATH} not found: build it with `python __graft_entry__.py` (or make -C '
                f'dsp-speech-recognition_amd/csrc). There is no CPU fallback.')
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.dsp_abi_version() != 1:
            raise DspError(f'ABI version mismatch: {lib.dsp_abi_version()}')
        _lib = lib
    return _lib


def current_device():
    """HIP device current on this thread (plans and scratch buffers are per device)."""
    dev = C.c_int(-1)
    check(load().dsp_get_device(C.byref(dev)))
    return dev.value


def check(rc):
    if rc != OK:
        msg = load().dsp_last_error()
        raise DspError(f'dsp_frontend error {rc}: {msg.decode() if msg else "?"}')


_device_ready = False


def require_device():
    """Fail loudly when no GPU is usable (product path has no CPU route)."""
    global _device_ready
    if _device_ready:
        return
    lib = load()
    n = C.c_int(0)
    rc = lib.dsp_device_count(C.byref(n))
    if rc != OK or n.value < 1:
        msg = lib.dsp_last_error()
        raise DspError('no MI355X/HIP device available; this package has no CPU fallback '
                       f'({msg.decode() if msg else "device count 0"})')
    _device_ready = True


def frame_count(n, L, S):
    out = c_i64(0)
    check(load().dsp_frame_count(int(n), int(L), int(S), C.byref(out)))
    return out.value


def frame_offsets(sample_offsets, L, S):
    so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
    fo = np.empty_like(so)
    check(load().dsp_frame_offsets(so.ctypes.data, len(so) - 1, int(L), int(S), fo.ctypes.data))
    return fo


class DeviceBuffer:
    """Owning handle of hipMalloc'ed memory."""

    def __init__(self, nbytes):
        require_device()
        self.nbytes = int(nbytes)
        p = c_vp(0)
        check(load().dsp_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(load().dsp_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, stream))
        check(load().dsp_stream_synchronize(stream))  # `arr` may be a temporary
        return self

    def download(self, shape, dtype, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        check(load().dsp_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream))
        return out

    def free(self):
        if getattr(self, 'ptr', None):
            try:
                load().dsp_free(self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __del__(self):
        self.free()


class Scratch:
    """Grow-only named device buffers so single-utterance calls do not hipMalloc every time.

    Slots are PER THREAD (and per device): every drop-in call uploads, launches and downloads on its
    own thread's buffers, so concurrent callers never share a slot (ctypes releases the GIL during
    the C calls).  A slot is only ever replaced by the thread that owns it, between its own calls;
    the old buffer's hipFree waits for work still queued on it.  Results that outlive the call
    (``download=False`` paths) are never scratch slots: they are owned DeviceBuffers.
    """

    def __init__(self, alloc=None):
        self._tls = threading.local()
        self._alloc = alloc or DeviceBuffer

    def _bufs(self):
        d = getattr(self._tls, 'bufs', None)
        if d is None:
            d = self._tls.bufs = {}
        return d

    def get(self, name, nbytes, device=None):
        key = (current_device() if device is None else device, name)   # hipMalloc'ed memory belongs to one device
        bufs = self._bufs()
        b = bufs.get(key)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self._alloc(max(int(nbytes * 1.5), 4096))
            bufs[key] = b
        return b


SCRATCH = Scratch()


def device_array(name, arr, stream=None):
    """Upload a host array into this thread's named scratch slot; returns the DeviceBuffer."""
    arr = np.ascontiguousarray(arr)
    return SCRATCH.get(name, arr.nbytes).upload(arr, stream)


def as_wave(sig):
    """Map a host signal to (contiguous array, dtype code): int16 stays int16 (reader.py:80), every
    other numeric type is rounded to fp32 (the arithmetic type of the kernels)."""
    a = np.asarray(sig)
    if a.dtype == np.int16:
        return np.ascontiguousarray(a), WAVE_I16
    return np.ascontiguousarray(a, dtype=np.float32), WAVE_F32
