"""Seeded random configurations: the specialised kernels (NFFT 512 and 1536) against the generic
table-driven kernel on the same ragged and dense batches.  The two are independent implementations
of the reference path (sigproc.py:66-185, base.py:8-79), so agreement on shapes nobody hand-picked
-- 1-filter banks, 1 cepstrum, tiny hops, odd lengths, band limits -- guards the table builders."""
import numpy as np
import pytest

from conftest import normwise

pytestmark = pytest.mark.gpu
TOLD = 1e-4


def _random_cfg(rng, nfft):
    rate = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    if nfft == 512:
        L = int(rng.integers(16, 513))
        S = 2 * int(rng.integers(1, 115))          # even hop, 7 S + 512 fits the wave buffer
        M = int(rng.integers(1, 65))
    else:
        L = int(rng.integers(64, 1537))
        S = int(rng.integers(1, 600))
        M = int(rng.integers(1, 65))
    C = int(rng.integers(1, min(M, 16) + 1))
    low = float(rng.choice([0, 0, 50, 300]))
    high = None if rng.random() < 0.5 else float(rng.uniform(low + 1000, rate / 2))
    return dict(samplerate=rate, winlen=(L + 0.25) / rate, winstep=(S + 0.25) / rate, numcep=C, nfilt=M,
                nfft=nfft, lowfreq=low, highfreq=high, preemph=float(rng.choice([0.0, 0.95, 0.97])),
                ceplifter=int(rng.choice([0, 22])), appendEnergy=bool(rng.integers(0, 2))), L, S


@pytest.mark.parametrize('nfft', [512, 1536])
def test_random_configs_fast_vs_generic(nfft):
    from features.batch import FeaturePlan
    from features import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(1000 + nfft)
    n_fast, worst = 0, 0.0
    for trial in range(40):
        cfg, L, S = _random_cfg(rng, nfft)
        try:
            plan = FeaturePlan(winfunc=np.hamming if trial % 2 else np.hanning, **cfg)
        except AssertionError:
            continue                                 # highfreq > rate / 2 after rounding: reference asserts too
        assert (plan.L, plan.S) == (L, S)
        if not lib.dsp_plan_has_fast_path(plan.plan.handle):
            continue
        n_fast += 1
        dtype = np.int16 if trial % 3 == 0 else np.float32
        lens = [int(x) for x in rng.integers(1, 6 * L + 7 * S, 9)] + [L, L + 1, 1]
        so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        if dtype == np.int16:
            flat = np.clip(np.round(3000 * rng.standard_normal(so[-1])), -32768, 32767).astype(np.int16)
        else:
            flat = (0.25 * rng.standard_normal(so[-1])).astype(np.float32)
        n_dense = (3 * L + 5 * S) // 4 * 4
        dense = flat[:5 * n_dense].reshape(5, n_dense) if len(flat) >= 5 * n_dense else None
        got_r, fo = plan.mfcc_batch(flat, sample_offsets=so, delta_n=1)
        got_d = plan.mfcc_batch(dense, delta_n=1)[0] if dense is not None else None
        try:
            nat.check(lib.dsp_debug_force_generic(1))
            ref_r, _ = plan.mfcc_batch(flat, sample_offsets=so, delta_n=1)
            ref_d = plan.mfcc_batch(dense, delta_n=1)[0] if dense is not None else None
        finally:
            nat.check(lib.dsp_debug_force_generic(0))
        assert np.isfinite(got_r).all(), (trial, cfg)
        # two fp32 pipelines; observed worst case 3.2e-5 (NFFT 512) and 1.3e-6 (NFFT 1536)
        tol = 1e-4
        worst = max(worst, normwise(got_r, ref_r))
        assert normwise(got_r, ref_r) <= tol, (trial, cfg, normwise(got_r, ref_r))
        if dense is not None:
            assert normwise(got_d, ref_d) <= tol, (trial, cfg, 'dense', normwise(got_d, ref_d))
    assert n_fast >= 15, n_fast
    print(f'nfft={nfft}: {n_fast} configurations on the fast path, worst normwise {worst:.2e}')


@pytest.mark.parametrize('dtype', [np.int16, np.float64])
def test_random_clips_endpoints_exact(dtype):
    """300 random clips (length, burst count / position / level, noise floor) through the batched
    device endpointing: every (left, right) pair must equal the oracle's (endpoint.py:34-66)."""
    from features.batch import EndpointPlan
    from oracle import dsp_oracle
    rng = np.random.default_rng(77)
    clips = []
    for _ in range(300):
        n = int(rng.uniform(0.2, 3.0) * 16000)
        x = rng.normal(0, rng.choice([1.0, 30.0, 300.0]), n)
        for _ in range(int(rng.integers(0, 4))):
            blen = int(rng.uniform(0.05, 0.9) * n)
            b0 = int(rng.integers(0, max(1, n - blen)))
            t = np.arange(blen) / 16000.0
            x[b0:b0 + blen] += rng.choice([200.0, 2000.0, 8000.0]) * np.sin(2 * np.pi * rng.uniform(80, 3000) * t) * np.hanning(blen)
        x = np.clip(np.round(x), -32768, 32767)
        clips.append(x.astype(np.int16) if dtype == np.int16 else x / 32768.0)
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    flat = np.concatenate(clips)
    got = EndpointPlan(16000, 0.03, 0.01).detect_batch(flat.astype(np.float32) if dtype != np.int16 else flat,
                                                       sample_offsets=so)
    bad = []
    for b, c in enumerate(clips):
        ref_in = c if dtype == np.int16 else c.astype(np.float32)     # the device sees fp32 samples
        if tuple(got[b]) != dsp_oracle.basic_endpoint_detection(ref_in, 16000):
            bad.append((b, tuple(got[b]), dsp_oracle.basic_endpoint_detection(ref_in, 16000)))
    assert not bad, bad[:5]


@pytest.mark.parametrize('dtype,rate', [(np.int16, 16000), (np.float64, 16000), (np.int16, 44100)])
def test_random_clips_robust_endpoints_exact(dtype, rate):
    """Random clips with voiced (harmonic) and unvoiced (noise) bursts through the batched device form of
    endpoint.robust_endpoint_detection (endpoint.py:68-92: dsp_acr_gate_batch + dsp_endpoint_rule_acr_batch): every
    (left, right) pair must equal the oracle's, and so must the gate bit of every frame."""
    from features.batch import EndpointPlan
    from features import _native as nat
    from oracle import dsp_oracle
    rng = np.random.default_rng(99 + rate)
    clips = []
    for _ in range(200 if rate == 16000 else 60):
        n = int(rng.uniform(0.3, 2.5) * rate)
        x = rng.normal(0, rng.choice([1.0, 30.0, 300.0]), n)
        for _ in range(int(rng.integers(0, 4))):
            blen = int(rng.uniform(0.05, 0.9) * n)
            b0 = int(rng.integers(0, max(1, n - blen)))
            t = np.arange(blen) / float(rate)
            if rng.random() < 0.7:      # voiced: a few harmonics of a pitch in the gate's range, some of it borderline noisy
                f0 = rng.uniform(60, 450)
                burst = sum(np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) / h for h in range(1, int(rng.integers(2, 6))))
                burst = burst + rng.choice([0.0, 0.3, 1.0, 2.0]) * rng.standard_normal(blen)
            else:
                burst = rng.standard_normal(blen)
            x[b0:b0 + blen] += rng.choice([200.0, 2000.0, 8000.0]) * burst * np.hanning(blen)
        x = np.clip(np.round(x), -32768, 32767)
        clips.append(x.astype(np.int16) if dtype == np.int16 else x / 32768.0)
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    flat = np.concatenate(clips)
    plan = EndpointPlan(rate, 0.03, 0.01, robust=True)
    got = plan.detect_batch(flat.astype(np.float32) if dtype != np.int16 else flat, sample_offsets=so)
    voiced = nat.SCRATCH.get('ep_voiced', 1).download((int(plan.layout(flat, so).total_frames),), np.uint8)
    fo = plan.layout(flat, so).frame_offsets
    bad, n_voiced, n_gate_bad = [], 0, 0
    for b, c in enumerate(clips):
        ref_in = c if dtype == np.int16 else c.astype(np.float32)     # the device sees fp32 samples
        want = dsp_oracle.robust_endpoint_detection(ref_in, rate)
        if tuple(got[b]) != want:
            bad.append((b, tuple(got[b]), want))
        if b % 10 == 0:     # the gate itself, frame by frame (sigproc.acr in fp64)
            frames = dsp_oracle.to_frames(ref_in, rate, 0.03, 0.01)
            for t_, fr in enumerate(frames):
                e0 = dsp_oracle.acr(fr, 0)
                with np.errstate(invalid='ignore', divide='ignore'):
                    w = max(dsp_oracle.acr(fr, n_) for n_ in range(rate // 500, rate // 50)) / e0 > 0.55
                n_voiced += int(w)
                n_gate_bad += int(bool(voiced[fo[b] + t_]) != bool(w))
    assert not bad, bad[:5]
    assert n_gate_bad == 0 and n_voiced > 50, (n_gate_bad, n_voiced)


@pytest.mark.parametrize('nfft,rate,winlen', [(512, 16000, 0.025), (1536, 48000, 0.03)])
def test_degenerate_signals_through_the_fused_kernels(nfft, rate, winlen):
    """All-zero clips (every filterbank energy and the frame energy hit the eps substitution of
    base.py:26,30), DC, a full-scale alternating square wave, one impulse, digital silence with a
    speech-like tail: fused kernels vs the oracle, dense and ragged, int16 and fp32."""
    from features.batch import FeaturePlan
    from oracle import dsp_oracle
    cfg = dict(samplerate=rate, winlen=winlen, winstep=0.01, numcep=13, nfilt=26, nfft=nfft, lowfreq=0,
               highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
    plan = FeaturePlan(winfunc=np.hamming, **cfg)
    n = rate // 2
    rng = np.random.default_rng(3)
    sigs = {
        'zeros': np.zeros(n, dtype=np.int16),
        'dc': np.full(n, 1234, dtype=np.int16),
        'square': np.tile(np.array([32767, -32768], dtype=np.int16), n // 2),
        'impulse': np.concatenate([np.zeros(n // 3, dtype=np.int16), [20000], np.zeros(n - n // 3 - 1, dtype=np.int16)]).astype(np.int16),
        'silence_then_noise': np.concatenate([np.zeros(n // 2, dtype=np.int16),
                                              np.round(500 * rng.standard_normal(n - n // 2)).astype(np.int16)]),
    }
    names = list(sigs)
    for dtype in (np.int16, np.float32):
        dense = np.stack([sigs[k] for k in names]).astype(dtype)
        out, fo = plan.mfcc_batch(dense, delta_n=2)
        flat = np.concatenate([sigs[k][:n - 7 * i] for i, k in enumerate(names)]).astype(dtype)
        so = np.concatenate(([0], np.cumsum([n - 7 * i for i in range(len(names))]))).astype(np.int64)
        out_r, fo_r = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
        assert np.isfinite(out).all() and np.isfinite(out_r).all()
        for b, k in enumerate(names):
            ref = dsp_oracle.mfcc_delta(dense[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **cfg)
            # exact-zero spectra: both sides are log(eps) constants; elsewhere the 1e-4 normwise bar.
            # A bin that is exactly 0 in fp64 can be ~1e-30 in fp32 (and vice versa) only for the
            # non-zero signals' leakage floor, where log() differences are bounded by the atol term.
            assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOLD, (dtype, k, normwise(out[fo[b]:fo[b + 1]], ref))
            ref_r = dsp_oracle.mfcc_delta(flat[so[b]:so[b + 1]].astype(np.float64), delta_n=2, winfunc=np.hamming, **cfg)
            assert normwise(out_r[fo_r[b]:fo_r[b + 1]], ref_r) <= TOLD, (dtype, k, "ragged")


def test_pitch_scores_batch_and_tracks_vs_oracle():
    """Row f-4: the per-frame pitch scores (centre clip -> complex band-pass FIR -> |.| -> 180
    autocorrelation lags) for a ragged batch of 10 kHz signals in one launch, and the pitch tracks
    of pitch_detect_sr on raw clips, against the oracle (pitch.py:96-132)."""
    import features
    from features import pitch as gp
    from oracle import dsp_oracle
    rng = np.random.default_rng(61)
    clips = []
    for i in range(10):
        n = int(rng.integers(2000, 9000))
        t = np.arange(n) / 10000.0
        f0 = rng.uniform(80, 400)
        x = sum(np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6)) / h for h in range(1, 6))
        clips.append(np.round(4000 * x * np.hanning(n) + 30 * rng.standard_normal(n)))
    clips.append(np.zeros(700))                       # all zero: median 0, every score 0
    clips.append(-np.abs(clips[0][:900]) - 1.0)       # no non-negative sample: NaN clip level -> zeros
    clips.append(clips[1][:150])                      # shorter than one frame (zero padded)
    so = np.concatenate(([0], np.cumsum([len(c) for c in clips]))).astype(np.int64)
    scores, fo = gp.frame_scores_batch(np.concatenate(clips), so, 300, 100)
    assert scores.shape == (fo[-1], 180)
    # the one-workgroup-per-frame kernel (bitonic sort, scalar loops) is an independent implementation
    from features import _native as nat
    for L in (300, 512):
        try:
            nat.check(nat.load().dsp_debug_force_generic(1))
            slow, _ = gp.frame_scores_batch(np.concatenate(clips), so, L, 100)
        finally:
            nat.check(nat.load().dsp_debug_force_generic(0))
        fast, _ = gp.frame_scores_batch(np.concatenate(clips), so, L, 100)
        assert normwise(fast, slow) <= 1e-5, (L, normwise(fast, slow))
    for b, c in enumerate(clips):
        frames = dsp_oracle.to_frames(c, 10000, 0.03, 0.01)
        assert fo[b + 1] - fo[b] == len(frames)
        ref = np.stack([dsp_oracle.pitch_frame_scores(dsp_oracle.center_clip(fr, False), 10000) for fr in frames])
        got = scores[fo[b]:fo[b + 1]]
        if np.max(np.abs(ref)) == 0:
            assert np.max(np.abs(got)) == 0, b
        else:
            assert normwise(got, ref) <= 1e-4, (b, normwise(got, ref))
    # whole tracks from raw clips at two rates: identical Hz values frame by frame
    for rate in (16000, 44100):
        n = int(0.6 * rate)
        t = np.arange(n) / rate
        sig = np.round(3000 * (np.sin(2 * np.pi * 140 * t) + 0.5 * np.sin(2 * np.pi * 280 * t)) * np.hanning(n)
                       + 40 * rng.standard_normal(n)).astype(np.int16)
        got, frames = features.pitch_detect_sr(sig, rate, winlen=0.03, step=0.01)
        ref, ref_frames = dsp_oracle.pitch_detect_sr(sig, rate, winlen=0.03, step=0.01)
        assert np.array_equal(np.asarray(frames), ref_frames)
        assert len(got) == len(ref)
        same = np.isclose(got, ref, rtol=1e-9, atol=0)
        from conftest import record
        record('pitch_track_mismatch_fraction', 1.0 - same.mean())
        assert same.all(), (rate, same.mean())     # measured: every frame identical (parity_measured.json)


def test_pitch_helpers_on_the_device_equal_the_oracle():
    """features.pitch.smooth / max_pitch / robust_max_pitch are batches of one through dsp_pitch_rows_batch: exact
    against the oracle's restatement of pitch.py:157-206 on random score rows (ties, a one-row input, other degrees)."""
    import features
    from features.pitch import smooth, max_pitch, robust_max_pitch
    from oracle import dsp_oracle
    rng = np.random.default_rng(8)
    for T, n, degree in ((57, 180, 2), (1, 180, 2), (2, 7, 2), (40, 65, 1), (33, 130, 3), (5, 180, 4)):
        g = rng.standard_normal((T, n)) * rng.uniform(0.1, 100)
        g[T // 2, 3] = g[T // 2, 5] = g[T // 2].max() + 1.0          # a tie: the first index wins
        with np.errstate(invalid='ignore'), np.testing.suppress_warnings() as sup:
            sup.filter(RuntimeWarning)
            want_s = np.asarray(dsp_oracle.smooth(g, degree))
        got_s = np.asarray(smooth(g, degree))
        assert got_s.shape == want_s.shape and np.allclose(got_s, want_s, rtol=1e-15, atol=0, equal_nan=True), (T, n, degree)
        assert np.array_equal(np.asarray(max_pitch(g, 20)), np.asarray(dsp_oracle.max_pitch(g, 20)))
        assert np.array_equal(np.asarray(robust_max_pitch(g, 20)), np.asarray(dsp_oracle.robust_max_pitch(g, 20)))
    assert features.smooth is smooth and features.robust_max_pitch is robust_max_pitch


@pytest.mark.parametrize('rate', [11025, 16000, 22050, 44100, 48000])
def test_device_decimation_equals_preprocess_downsampling(rate):
    """dsp_resample_layout_batch + dsp_decimate_batch against preprocess.downsampling (preprocess.py:21-28) on a ragged
    batch: every kept sample, every offset, and the frame offsets of the decimated clips."""
    import torch
    from features import _native as nat
    from oracle import dsp_oracle
    rng = np.random.default_rng(rate)
    lens = [0, 1, 2, 3, 4, 5, 7, 441, 4410, 4411] + [int(x) for x in rng.integers(1, 30000, 30)]
    so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    x = rng.standard_normal(int(so[-1])).astype(np.float32)
    dev = torch.device('cuda', 0)
    d_x, d_so = torch.from_numpy(x).to(dev), torch.from_numpy(so).to(dev)
    B = len(lens)
    d_so10 = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_fo = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_out = torch.full((x.size + 1,), float('nan'), dtype=torch.float32, device=dev)
    lib = nat.load()
    L, S = 300, 100
    nat.check(lib.dsp_resample_layout_batch(d_so.data_ptr(), B, rate, 10000, L, S, d_so10.data_ptr(), d_fo.data_ptr(), None))
    nat.check(lib.dsp_decimate_batch(d_x.data_ptr(), d_so.data_ptr(), d_so10.data_ptr(), B, x.size, rate, 10000, d_out.data_ptr(), None))
    torch.cuda.synchronize()
    so10, fo, out = d_so10.cpu().numpy(), d_fo.cpu().numpy(), d_out.cpu().numpy()
    want = [np.asarray(dsp_oracle.downsampling(x[so[b]:so[b + 1]], rate, 10000), dtype=np.float32) for b in range(B)]
    assert np.array_equal(np.diff(so10), [len(w) for w in want])
    assert np.array_equal(out[:so10[-1]], np.concatenate(want)) and np.isnan(out[so10[-1]:]).all()
    assert np.array_equal(np.diff(fo), [dsp_oracle.frame_geometry(len(w), L, S)[2] for w in want])


def test_long_utterance_takes_the_serial_rule_path():
    """Utterances longer than 2048 VAD frames (20.5 s) do not fit the rule kernel's LDS copy and are
    scanned straight from HBM by one lane: same endpoints as the oracle, mixed in a batch with short
    clips that take the LDS path."""
    from features.batch import EndpointPlan
    from oracle import dsp_oracle
    rng = np.random.default_rng(5)
    clips = []
    for n, b0, blen in ((16000 * 31, 16000 * 7, 16000 * 9), (16000, 3000, 9000), (16000 * 25, 16000 * 20, 16000 * 3)):
        x = rng.normal(0, 30, n)
        t = np.arange(blen) / 16000.0
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * 180 * t) * np.hanning(blen)
        clips.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    so = np.concatenate(([0], np.cumsum([len(c) for c in clips]))).astype(np.int64)
    got = EndpointPlan(16000, 0.03, 0.01).detect_batch(np.concatenate(clips), sample_offsets=so)
    for b, c in enumerate(clips):
        assert tuple(got[b]) == dsp_oracle.basic_endpoint_detection(c, 16000), b


@pytest.mark.parametrize('T,D,N', [(500, 64, 9), (1, 5, 4), (3, 13, 3), (1000, 200, 2), (129, 13, 1)])
def test_delta_shapes_beyond_the_tiled_kernel(T, D, N):
    """base.delta for widths / window sizes whose tile does not fit the LDS budget (per-element
    kernel), for T < N (everything is edge padding) and across the 128-frame tile seam."""
    import features
    from oracle import dsp_oracle
    x = np.random.default_rng(T + D).standard_normal((T, D))
    got = features.delta(x, N)
    ref = dsp_oracle.delta(x.astype(np.float32).astype(np.float64), N)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))


def test_device_pitch_tracker_equals_the_reference_sequence():
    """dsp_pitch_track_batch (in-place smoothing with its end-of-utterance windows, first arg-max, two octave-repair
    sweeps; pitch.py:157-206) against the host restatement of those loops on random score tables -- utterances of
    1, 2, 3, 4, 5 frames (empty / short smoothing windows), ties, a NaN score, and one longer than the kernel's LDS
    buffer.  Pitch values are fp64 and must be IDENTICAL."""
    import ctypes as C
    from features import _native as nat
    from oracle.dsp_oracle import smooth, robust_max_pitch       # the checker (pitch.py:157-206 restated)
    lib = nat.load()
    rng = np.random.default_rng(77)
    Ts = [1, 2, 3, 4, 5, 17, 64, 150, 2100, 1, 33]
    fo = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int64)
    n_lags = 180
    sc = rng.random((int(fo[-1]), n_lags)).astype(np.float32)
    # half-frequency structure so that the repair sweeps fire: a strong peak at lag 2 L next to a weaker one at L
    for b, T in enumerate(Ts):
        for t in range(T):
            lag = int(rng.integers(10, 60))
            sc[fo[b] + t, lag] += 1.5
            if rng.random() < 0.5:
                sc[fo[b] + t, min(2 * lag + 20, n_lags - 1)] += 1.6
    sc[fo[5] + 3, 7] = sc[fo[5] + 3, 9] = 9.0           # a tie: the first maximum wins
    sc[fo[6] + 10, 40] = np.nan                        # numpy's arg-max takes the first NaN
    d_sc = nat.DeviceBuffer(sc.nbytes).upload(sc)
    d_fo = nat.DeviceBuffer(fo.nbytes).upload(fo)
    d_p = nat.DeviceBuffer(int(fo[-1]) * 8)
    nat.check(lib.dsp_pitch_track_batch(d_sc.ptr, d_fo.ptr, len(Ts), n_lags, 20, 2, d_p.ptr, None))
    got = d_p.download((int(fo[-1]),), np.float64)
    import warnings
    for b, T in enumerate(Ts):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')             # the one-frame utterance averages over nothing (NaN), as the reference
            want = np.asarray(robust_max_pitch(smooth(sc[fo[b]:fo[b + 1]].astype(np.float64), 2), bias=20))
        assert np.array_equal(got[fo[b]:fo[b + 1]], want), (b, T, np.argwhere(got[fo[b]:fo[b + 1]] != want)[:5])
    assert lib.dsp_pitch_track_batch(d_sc.ptr, d_fo.ptr, len(Ts), n_lags, 20, 3, d_p.ptr, None) == -1


def test_vad_kernel_at_random_frame_lengths_and_hops():
    """dsp_vad_features_batch on int16 clips at 24 random (frame length, hop) pairs of either parity -- whole vectors, odd
    hops, hops longer than a quarter of the frame, frames a few vectors long -- against sums formed in NumPy on frames cut
    as sigproc.to_frames cuts them (sigproc.py:11-19).  Every amplitude sum and every zero-crossing count exact."""
    from features import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(77)
    for case in range(24):
        L = int(rng.integers(64, 1500)) if case % 3 else 4 * int(rng.integers(16, 360))
        S = int(rng.integers(max(8, L // 6), L + 1)) if case % 4 else 4 * int(rng.integers(4, L // 4 + 1))
        lens = [int(rng.integers(1, 6 * L)) for _ in range(7)] + [L, L + 1, L + S, L - 1, 1]
        clips = [rng.integers(-32768, 32768, n).astype(np.int16) if k % 3 else rng.integers(-3, 4, n).astype(np.int16)
                 for k, n in enumerate(lens)]
        so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        fo = nat.frame_offsets(so, L, S)
        flat = np.concatenate(clips)
        d_x = nat.DeviceBuffer(flat.nbytes).upload(flat)
        d_so = nat.DeviceBuffer(so.nbytes).upload(so)
        d_fo = nat.DeviceBuffer(fo.nbytes).upload(fo)
        d_amp = nat.DeviceBuffer(int(fo[-1]) * 8)
        d_zcr = nat.DeviceBuffer(int(fo[-1]) * 4)
        nat.check(lib.dsp_vad_features_batch(d_x.ptr, nat.WAVE_I16, d_so.ptr, d_fo.ptr, len(lens), int(fo[-1]), 0, L, S, 0,
                                             d_amp.ptr, d_zcr.ptr, None))
        amp = d_amp.download((int(fo[-1]),), np.float64)
        zcr = d_zcr.download((int(fo[-1]),), np.int32)
        for b, c in enumerate(clips):
            n = len(c)
            T = 1 if n <= L else 1 + -(-(n - L) // S)
            assert fo[b + 1] - fo[b] == T, (L, S, n)
            x = np.zeros((T - 1) * S + L, dtype=np.int64)
            x[:n] = c
            fr = np.stack([x[t * S:t * S + L] for t in range(T)])
            want_amp = np.abs(fr).sum(1).astype(np.float64)
            want_zcr = ((fr[:, 1:] * fr[:, :-1]) < 0).sum(1)
            assert np.array_equal(amp[fo[b]:fo[b + 1]], want_amp), (case, L, S, b, n)
            assert np.array_equal(zcr[fo[b]:fo[b + 1]], want_zcr), (case, L, S, b, n)
