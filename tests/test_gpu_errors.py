"""Error behaviour of the C ABI (include/dsp_frontend.h): bad arguments come back as negative status
codes with a message -- never a crash, never a silent CPU path -- and the Python mirror raises what
the reference raises (base.py:42,71-72)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from features import _native as nat
    from features.batch import FeaturePlan
    nat.require_device()
    plan = FeaturePlan(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512,
                       winfunc=np.hamming)
    return nat, nat.load(), plan


def _msg(lib):
    return lib.dsp_last_error().decode()


def test_status_codes_and_messages(env):
    nat, lib, plan = env
    x = nat.DeviceBuffer(4 * 16000).upload(np.zeros(16000, dtype=np.float32))
    out = nat.DeviceBuffer(4 * 99 * 39)
    h = plan.plan.handle
    # NULL output / plan
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, nat.OUT_MFCC, None, 0, None, None) == -1
    assert 'NULL' in _msg(lib)
    assert lib.dsp_features_batch(None, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, nat.OUT_MFCC, out.ptr, 0, None, None) == -1
    # frame count that does not match the geometry
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 98, 16000, nat.OUT_MFCC, out.ptr, 0, None, None) == -1
    assert 'n_frames_total' in _msg(lib)
    # unknown output kind, row stride smaller than the row, fbank without the energy buffer
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, 17, out.ptr, 0, None, None) == -1
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, nat.OUT_MFCC, out.ptr, 5, None, None) == -1
    assert 'ld_out' in _msg(lib)
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, nat.OUT_FBANK, out.ptr, 0, None, None) == -1
    # unsupported sample type; ragged batch without offsets
    assert lib.dsp_features_batch(h, x.ptr, 7, None, None, 1, 99, 16000, nat.OUT_MFCC, out.ptr, 0, None, None) == -1
    assert lib.dsp_features_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 0, nat.OUT_MFCC, out.ptr, 0, None, None) == -1
    # the fused MFCC + delta entry point validates before its first launch (a NULL d_out used to reach a kernel)
    assert lib.dsp_mfcc_delta_batch(h, x.ptr, nat.WAVE_F32, None, None, 1, 99, 16000, 2, None, None) == -1
    assert 'NULL' in _msg(lib)
    assert lib.dsp_mfcc_delta_batch(h, None, nat.WAVE_F32, None, None, 1, 99, 16000, 2, out.ptr, None) == -1
    assert lib.dsp_mfcc_delta_batch(h, x.ptr, nat.WAVE_F32, None, None, 0, 99, 16000, 2, out.ptr, None) == -1
    # delta: N < 1 is the reference's ValueError (base.py:71-72)
    assert lib.dsp_delta_batch(out.ptr, 39, None, 1, 99, 99, 13, 0, out.ptr, 39, None, 0, None) == -1
    assert 'N must be an integer >= 1' in _msg(lib)
    assert lib.dsp_model_finalize_batch(out.ptr, 13, None, 1, 13, 3, 200, out.ptr, out.ptr, None) == -1
    # after all that the library still works
    res, _ = plan.mfcc_batch(np.zeros((1, 16000), dtype=np.float32), delta_n=2)
    assert res.shape == (99, 39) and np.isfinite(res).all()


def test_round3_entry_points_reject_bad_arguments(env):
    """dsp_layout_* / dsp_*_segments_batch / dsp_pitch_track_batch: NULL handles, empty batches and too small a work
    buffer come back as DSP_EINVAL with a message -- before any launch."""
    import ctypes as C
    nat, lib, plan = env
    h = plan.plan.handle
    so = np.array([0, 9000, 20000], dtype=np.int64)
    d_so = nat.DeviceBuffer(so.nbytes).upload(so)
    fo = nat.frame_offsets(so, 480, 160)
    d_fo = nat.DeviceBuffer(fo.nbytes).upload(fo)
    out = C.c_void_p(0)
    assert lib.dsp_layout_create(None, 2, int(fo[-1]), 480, 160, None, C.byref(out)) == -1
    assert lib.dsp_layout_create(d_fo.ptr, 0, int(fo[-1]), 480, 160, None, C.byref(out)) == -1
    assert lib.dsp_layout_create(d_fo.ptr, 2, int(fo[-1]), 480, 160, None, C.byref(out)) == 0 and out.value
    x = nat.DeviceBuffer(2 * 20000).upload(np.zeros(20000, dtype=np.int16))
    amp, zcr = nat.DeviceBuffer(8 * int(fo[-1])), nat.DeviceBuffer(4 * int(fo[-1]))
    assert lib.dsp_vad_features_layout_batch(None, x.ptr, nat.WAVE_I16, d_so.ptr, d_fo.ptr, 0, amp.ptr, zcr.ptr, None) == -1
    assert lib.dsp_vad_features_layout_batch(out.value, x.ptr, nat.WAVE_I16, d_so.ptr, d_fo.ptr, 0, None, zcr.ptr, None) == -1
    assert lib.dsp_vad_features_layout_batch(out.value, x.ptr, nat.WAVE_I16, d_so.ptr, d_fo.ptr, 0, amp.ptr, zcr.ptr, None) == 0
    assert lib.dsp_layout_destroy(out.value) == 0 and lib.dsp_layout_destroy(None) == 0
    nbytes = C.c_size_t(0)
    assert lib.dsp_segments_workspace_bytes(None, 2, 200, C.byref(nbytes)) == -1
    assert lib.dsp_segments_workspace_bytes(h, 2, 200, C.byref(nbytes)) == 0 and nbytes.value > 200 * 13 * 4
    seg = nat.DeviceBuffer(32).upload(np.array([[100, 8000], [0, 11000]], dtype=np.int64))
    mfo = nat.DeviceBuffer(24).upload(nat.frame_offsets(np.array([0, 7900, 18900], dtype=np.int64), 400, 160))
    work = nat.DeviceBuffer(int(nbytes.value))
    res = nat.DeviceBuffer(200 * 39 * 4)
    args = (h, x.ptr, nat.WAVE_I16, d_so.ptr, seg.ptr, mfo.ptr, 2, 200, 2, 1)
    assert lib.dsp_mfcc_delta_segments_batch(*args, work.ptr, 64, res.ptr, None) == -1      # work buffer too small
    assert 'work buffer' in _msg(lib)
    assert lib.dsp_mfcc_delta_segments_batch(*args, None, int(nbytes.value), res.ptr, None) == -1
    assert lib.dsp_mfcc_delta_segments_batch(*args[:8], -1, 1, work.ptr, int(nbytes.value), res.ptr, None) == -1   # delta N < 0 (0 = cepstra only, round 4)
    assert lib.dsp_mfcc_delta_segments_batch(*args[:8], 0, 1, work.ptr, int(nbytes.value), res.ptr, None) == 0
    assert lib.dsp_mfcc_delta_segments_batch(*args, work.ptr, int(nbytes.value), res.ptr, None) == 0
    assert lib.dsp_pitch_track_batch(None, d_fo.ptr, 2, 180, 20, 2, amp.ptr, None) == -1
    assert lib.dsp_pitch_track_batch(x.ptr, d_fo.ptr, 2, 300, 20, 2, amp.ptr, None) == -1
    nat.check(lib.dsp_stream_synchronize(None))


def test_plan_creation_rejects_bad_descriptions(env):
    nat, lib, _ = env
    from features import _plan
    with pytest.raises(nat.DspError):          # nfft neither 2^k nor 3 * 2^k
        _plan.mfcc_plan(16000, 0.025, 0.01, 13, 26, 500, 0, None, 0.97, 22, True, np.hamming)
    # numcep > nfilt: dct(...)[:, :numcep] just returns the nfilt columns there are (base.py:13)
    assert _plan.mfcc_plan(16000, 0.025, 0.01, 30, 26, 512, 0, None, 0.97, 22, True, np.hamming).C == 26
    with pytest.raises(AssertionError):        # base.py:42
        _plan.mfcc_plan(16000, 0.025, 0.01, 13, 26, 512, 0, 9000, 0.97, 22, True, np.hamming)
    handle = C.c_void_p(0)
    assert lib.dsp_plan_create(None, C.byref(handle)) == -1


def test_python_surface_raises_like_the_reference(env):
    import features
    with pytest.raises(ValueError, match='N must be an integer >= 1'):
        features.delta(np.zeros((5, 13)), 0)
    with pytest.raises(AssertionError):
        features.get_filterbanks(26, 512, 16000, 0, 9000)
