"""GPU parity, part 2: batched entry points at BASELINE.json's sizes and the properties that do not
need the oracle at full size.  Everything goes through the C ABI (features.batch -> ctypes)."""
import numpy as np
import pytest

from conftest import normwise
from oracle import dsp_oracle

pytestmark = pytest.mark.gpu

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, lowfreq=0,
           highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
FB_CFG = {k: v for k, v in CFG.items() if k not in ('numcep', 'ceplifter', 'appendEnergy')}
TOL = 1e-4


@pytest.fixture(scope='module')
def plan():
    from features.batch import FeaturePlan
    return FeaturePlan(winfunc=np.hamming, **CFG)


def _batch(seed, B, N=16000, dtype=np.float32):
    rng = np.random.default_rng(seed)
    if dtype == np.int16:
        return np.clip(np.round(3000 * rng.standard_normal((B, N))), -32768, 32767).astype(np.int16)
    return (0.25 * rng.standard_normal((B, N))).astype(np.float32)


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_config2_batch_1024_vs_oracle(plan, dtype):
    """configs[1]: 1024 x 1 s -> [1024*99, 39]; every wave iterates over several frame groups."""
    B = 1024
    waves = _batch(11, B, dtype=dtype)
    out, fo = plan.mfcc_batch(waves, delta_n=2)
    assert out.shape == (B * 99, 39) and fo[-1] == B * 99
    assert np.isfinite(out).all()
    worst = 0.0
    for b in list(range(0, B, 37)) + [B - 1]:
        ref = dsp_oracle.mfcc_delta(waves[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
        worst = max(worst, normwise(out[fo[b]:fo[b + 1]], ref))
    assert worst <= TOL, worst


def test_batch_equals_single_calls(plan):
    """Sharding invariance: an utterance's features do not depend on its batch neighbours."""
    import features
    B = 300
    waves = _batch(12, B)
    out, fo = plan.mfcc_batch(waves, delta_n=0)
    for b in (0, 1, 149, 299):
        single = features.mfcc(waves[b], winfunc=np.hamming, **CFG)
        assert np.array_equal(out[fo[b]:fo[b + 1]], single.astype(np.float32)), b


def test_generic_and_fast_kernels_agree(plan):
    """The specialised NFFT=512 kernel and the table-driven generic kernel are independent
    implementations: same batch through both (dense and ragged layouts) must agree."""
    from features import _native as nat
    lib = nat.load()
    assert lib.dsp_plan_has_fast_path(plan.plan.handle) == 1
    B, N = 64, 16000
    waves = _batch(13, B, N)
    tail = _batch(14, 1, 4001)[0]   # one odd-length utterance makes the layout ragged and misaligned
    flat = np.concatenate([tail, waves.reshape(-1)])
    so = np.concatenate([[0], 4001 + np.arange(B + 1) * N]).astype(np.int64)
    out_fast, _ = plan.mfcc_batch(waves, delta_n=2)
    out_fast_r, fo = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
    try:
        nat.check(lib.dsp_debug_force_generic(1))
        out_gen, _ = plan.mfcc_batch(waves, delta_n=2)
        out_gen_r, _ = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
    finally:
        nat.check(lib.dsp_debug_force_generic(0))
    assert normwise(out_gen, out_fast) <= 5e-5          # two fp32 pipelines, each ~1e-5 from fp64
    assert normwise(out_gen_r, out_fast_r) <= 5e-5
    # same utterances through the dense and the ragged instantiation: identical algorithm, but hipcc
    # contracts the window multiply into the first butterfly only in the dense one -> last-bit noise
    assert normwise(out_fast_r[fo[1]:], out_fast) <= 2e-5


def test_ragged_batch_vs_oracle(plan):
    rng = np.random.default_rng(15)
    lens = [16000, 300, 400, 401, 12345, 8000, 559, 16001, 32000, 1, 2, 3, 7, 1603, 1999]
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = (0.25 * rng.standard_normal(so[-1])).astype(np.float32)
    out, fo = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
    for b, n in enumerate(lens):
        ref = dsp_oracle.mfcc_delta(flat[so[b]:so[b + 1]].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
        assert out[fo[b]:fo[b + 1]].shape == ref.shape
        assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOL, (b, n)


def test_ragged_calls_queued_back_to_back(plan):
    """Ragged calls build their index tables in pooled device workspaces.  Queue many calls with
    different layouts on the default stream WITHOUT host syncs in between: every result must equal
    (bitwise) the same call run alone.  Guards the workspace lifetime (a buffer may only be reused
    once the kernels of its previous call have finished)."""
    from features import _native as nat
    rng = np.random.default_rng(151)
    jobs = []
    for j in range(12):
        n_utt = int(rng.integers(3, 40))
        lens = rng.integers(1, 9000, n_utt)
        so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = (0.25 * rng.standard_normal(so[-1])).astype(np.float32)
        alone, fo = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
        jobs.append((flat, so, alone.copy()))
    held = []
    for flat, so, _ in jobs:  # allocations and uploads synchronise: do them all up front
        lay = plan.layout(flat, so)
        held.append((lay, nat.DeviceBuffer(flat.nbytes).upload(flat),
                     nat.DeviceBuffer(lay.total_frames * 39 * 4)))
    for rep in range(5):
        for lay, d_wave, d_out in held:
            nat.check(nat.load().dsp_memset(d_out.ptr, 0, d_out.nbytes, None))
        for lay, d_wave, d_out in held:
            plan.run_raw(d_wave.ptr, nat.WAVE_F32, lay, d_out.ptr, delta_n=2)
        for j, (lay, d_wave, d_out) in enumerate(held):
            got = d_out.download((lay.total_frames, 39), np.float32)
            assert np.array_equal(got, jobs[j][2]), (rep, j)


@pytest.mark.parametrize('B,N,delta_n,dtype', [
    (1024, 16000, 2, np.float32),     # configs[1]: 2 utterances per workgroup, 99 frames each (groups span the seam)
    (515, 16000, 2, np.int16),        # uneven split: workgroups of 1 and 2 utterances
    (700, 16880, 1, np.float32),      # T = 104 (a multiple of 8), delta window 1
    (1300, 9600, 2, np.float32),      # T = 59: short utterances, 2-3 per workgroup
])
def test_fused_delta_kernel_equals_the_two_kernel_path(plan, B, N, delta_n, dtype):
    """Dense batches of >= 2 x CUs utterances take the fused MFCC + delta + delta-delta kernel
    (kernels_fast512.h, "Fused delta": workgroups own whole utterances, rows trail the computation by 2 N frames,
    one barrier).  Its rows must equal, bit for bit, what the MFCC kernel followed by dsp_delta_batch gives
    (same cepstra arithmetic, same delta formula, base.py:70-79 twice), and match the oracle on sampled utterances."""
    import torch
    from features import _native as nat
    lib = nat.load()
    waves = _batch(91, B, N, dtype=dtype)
    lay = plan.layout(waves)
    T = lay.total_frames // B
    dev = torch.device('cuda', 0)
    d_wave = torch.from_numpy(waves).to(dev)
    wd = nat.WAVE_I16 if dtype == np.int16 else nat.WAVE_F32
    fused = torch.full((B * T, 39), float('nan'), device=dev)
    plan.run_raw(d_wave.data_ptr(), wd, lay, fused.data_ptr(), delta_n, None)
    two = torch.full((B * T, 39), float('nan'), device=dev)
    nat.check(lib.dsp_features_batch(plan.plan.handle, d_wave.data_ptr(), wd, None, None, B, B * T, N, nat.OUT_MFCC,
                                     two.data_ptr(), 39, None, None))
    nat.check(lib.dsp_delta_batch(two.data_ptr(), 39, None, B, B * T, T, 13, delta_n, two.data_ptr() + 13 * 4, 39,
                                  two.data_ptr() + 26 * 4, 39, None))
    torch.cuda.synchronize()
    fused, two = fused.cpu().numpy(), two.cpu().numpy()
    assert np.isfinite(fused).all()
    assert np.array_equal(fused, two), np.argwhere(fused != two)[:8]
    for b in (0, 1, B // 2, B - 2, B - 1):
        ref = dsp_oracle.mfcc_delta(waves[b].astype(np.float64), delta_n=delta_n, winfunc=np.hamming, **CFG)
        assert normwise(fused[b * T:(b + 1) * T], ref) <= TOL, b


def test_workspace_pool_stays_bounded_when_the_host_queues_ahead(plan):
    """dsp_mfcc_delta_batch leases its dense-cepstra scratch from the workspace pool.  A host
    that queues hundreds of steps without synchronising must not grow the pool with the queue depth: work on ONE
    stream reuses the stream's last buffer at once, and three streams need three buffers."""
    import ctypes as C
    import torch
    from features import _native as nat
    lib = nat.load()
    B = 256
    waves = _batch(77, B)
    lay = plan.layout(waves)
    dev = torch.device('cuda', 0)
    d_wave = torch.from_numpy(waves).to(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = [torch.empty((B * 99, 39), device=dev) for _ in streams]
    torch.cuda.synchronize()
    for k, s in enumerate(streams):                      # warm up: one buffer per stream
        plan.run_raw(d_wave.data_ptr(), nat.WAVE_F32, lay, outs[k].data_ptr(), 2, s.cuda_stream)
    torch.cuda.synchronize()
    n0, b0 = C.c_longlong(0), C.c_longlong(0)
    nat.check(lib.dsp_debug_pool_stats(C.byref(n0), C.byref(b0)))
    for i in range(600):                                 # ~25 ms of queued device work, no host sync
        k = i % 3
        plan.run_raw(d_wave.data_ptr(), nat.WAVE_F32, lay, outs[k].data_ptr(), 2, streams[k].cuda_stream)
    n1, b1 = C.c_longlong(0), C.c_longlong(0)
    nat.check(lib.dsp_debug_pool_stats(C.byref(n1), C.byref(b1)))
    # (the warm-up calls may have shared buffers whose earlier users had finished: allow one new buffer per stream)
    assert n1.value <= n0.value + 3 and b1.value <= b0.value + (8 << 20), (n0.value, b0.value, n1.value, b1.value)
    for i in range(600):                                 # ... and from then on nothing
        k = i % 3
        plan.run_raw(d_wave.data_ptr(), nat.WAVE_F32, lay, outs[k].data_ptr(), 2, streams[k].cuda_stream)
    n2, b2 = C.c_longlong(0), C.c_longlong(0)
    nat.check(lib.dsp_debug_pool_stats(C.byref(n2), C.byref(b2)))
    torch.cuda.synchronize()
    assert (n2.value, b2.value) == (n1.value, b1.value), (n1.value, b1.value, n2.value, b2.value)
    ref, _ = plan.mfcc_batch(waves, delta_n=2)
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref)


def test_a_nan_clip_does_not_leak_into_its_neighbours():
    """with flat grouping a frame group may span two utterances of a dense batch.  Pass 1 reads
    16 x NROWS samples per frame (400 for L <= 400), beyond L they meet a zero window -- but 0 x NaN is NaN, so
    the next utterance's samples must not sit there.  A clean clip's rows must not depend on its neighbours."""
    from features.batch import FeaturePlan
    cfg = dict(CFG)
    cfg.update(winlen=0.02, winstep=0.01)                # L = 320 < 400: rows 20..24 of every frame are padding
    plan = FeaturePlan(winfunc=np.hamming, **cfg)
    waves = _batch(78, 6, 16000)
    clean, _ = plan.mfcc_batch(waves, delta_n=2)
    bad = waves.copy()
    bad[1, :] = np.nan
    bad[3, :400] = np.inf
    got, fo = plan.mfcc_batch(bad, delta_n=2)
    for b in (0, 2, 4, 5):
        assert np.array_equal(got[fo[b]:fo[b + 1]], clean[fo[b]:fo[b + 1]]), b


def test_parseval_full_size():
    """Oracle-free property: sum_n frame[n]^2 == pspec[0] + 2 sum_{0<k<256} pspec[k] + pspec[256]
     on an 8 s signal through the framesig / powspec kernels."""
    import features
    x = _batch(16, 1, 16000 * 8)[0].astype(np.float64)
    frames = features.framesig(x, 400, 160, np.hamming)
    ps = features.powspec(frames, 512)
    lhs = np.sum(frames ** 2, axis=1)
    rhs = ps[:, 0] + 2 * ps[:, 1:256].sum(axis=1) + ps[:, 256]
    assert np.max(np.abs(lhs - rhs) / np.max(lhs)) <= 2e-6


def test_quadratic_scaling_of_fbank():
    """fbank energies are quadratic: scaling the waveform by a scales feat and energy by a^2."""
    import features
    x = _batch(17, 1)[0].astype(np.float64)
    f1, e1 = features.fbank(x, winfunc=np.hamming, **FB_CFG)
    f2, e2 = features.fbank(4.0 * x, winfunc=np.hamming, **FB_CFG)
    assert normwise(f2, 16.0 * f1) <= 1e-5 and normwise(e2, 16.0 * e1) <= 1e-5


def test_known_answers():
    """DC -> only bin 0; alternating +-1 -> ZCR = L - 1; framesig row t == sig[t*S : t*S+L];
    delta of a ramp == slope in the interior; delta(N < 1) raises like the reference."""
    import features
    ps = features.powspec(np.ones((3, 512)), 512)
    assert np.allclose(ps[:, 0], 512.0, rtol=1e-6) and np.max(np.abs(ps[:, 1:])) < 1e-6
    alt = np.tile(np.array([1, -1], dtype=np.int16), 240)[None, :]
    assert features.get_zcr(alt) == [479]
    x = np.arange(4000, dtype=np.float64)
    fr = features.framesig(x, 400, 160)
    assert np.array_equal(fr[3], x[480:880]) and fr.shape == (24, 400)
    ramp = np.outer(np.arange(50, dtype=np.float64), np.ones(4)) * 0.5
    d = features.delta(ramp, 2)
    assert np.allclose(d[2:-2], 0.5, atol=1e-6)
    with pytest.raises(ValueError):
        features.delta(ramp, 0)


def test_endpoint_batch_matches_single():
    from features.batch import EndpointPlan
    from golden_cases import make_signal
    import features
    clips = [make_signal(('vad', 50 + i, 16000 + 800 * i)) for i in range(12)]
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    ep = EndpointPlan(16000, 0.03, 0.01)
    got = ep.detect_batch(np.concatenate(clips), sample_offsets=so)
    for b, c in enumerate(clips):
        assert tuple(got[b]) == features.basic_endpoint_detection(c, 16000)
        assert tuple(got[b]) == dsp_oracle.basic_endpoint_detection(c, 16000)


@pytest.mark.parametrize('rate', [16000, 48000, 22050, 1600])
@pytest.mark.parametrize('dtype', [np.int16, np.float32])
def test_vad_tile_kernel_vs_per_frame_kernel_and_oracle(rate, dtype):
    """The tiled amplitude / ZCR kernel (16 or 4 frames per wave) against the one-wave-per-frame
    kernel (dsp_debug_force_generic) and the oracle, dense and ragged.  int16: every value exact.
    rate 22050 gives L=661, S=220 (odd sizes), rate 1600 gives L=48 < 64 (no tile kernel)."""
    from features.batch import EndpointPlan
    from features import _native as nat
    ep = EndpointPlan(rate, 0.03, 0.01)
    rng = np.random.default_rng(91)
    lens = [rate, rate // 3 + 1, ep.L, ep.L - 1, 1, 2 * rate + 3, ep.L + 1, 5 * ep.S + 7]
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = _batch(92, 1, int(so[-1]), dtype=dtype)[0]
    flat[rng.integers(0, len(flat), len(flat) // 10)] = 0          # exact zeros never count as crossings
    dense = _batch(93, 6, (rate // 2) // 4 * 4, dtype=dtype)
    results = {}
    for mode in (0, 1):
        try:
            nat.check(nat.load().dsp_debug_force_generic(mode))
            _, amp_r, zcr_r, fo_r = ep.detect_batch(flat, sample_offsets=so, return_feature=True)
            _, amp_d, zcr_d, fo_d = ep.detect_batch(dense, return_feature=True)
            results[mode] = (amp_r.copy(), zcr_r.copy(), amp_d.copy(), zcr_d.copy())
        finally:
            nat.check(nat.load().dsp_debug_force_generic(0))
    tile, per = results[0], results[1]
    assert np.array_equal(tile[1], per[1]) and np.array_equal(tile[3], per[3])       # ZCR: integers
    if dtype == np.int16:
        assert np.array_equal(tile[0], per[0]) and np.array_equal(tile[2], per[2])   # sums of integers: exact
    else:
        assert np.allclose(tile[0], per[0], rtol=1e-12, atol=0) and np.allclose(tile[2], per[2], rtol=1e-12, atol=0)
    for b in range(len(lens)):
        frames = dsp_oracle.to_frames(flat[so[b]:so[b + 1]].astype(np.float64), rate, t=0.03, step=0.01)
        assert fo_r[b + 1] - fo_r[b] == frames.shape[0]
        assert np.allclose(tile[0][fo_r[b]:fo_r[b + 1]], dsp_oracle.get_amplitude(frames), rtol=1e-6, atol=1e-9)
        assert list(tile[1][fo_r[b]:fo_r[b + 1]]) == list(dsp_oracle.get_zcr(frames))


@pytest.mark.parametrize('rate', [16000, 48000, 8000, 44100, 11025, 32000, 22050])
def test_vad_int16_extreme_values_are_exact(rate):
    """The integer amplitude / zero-crossing kernel (vad_scan_kernel) at the ends of the int16 range: every sample
    -32768 (the largest per-vector sum, no crossing), +32767 / -32768 alternating (a crossing at every pair), runs of
    zeros between opposite signs (never a crossing), ragged lengths and odd sample offsets.  Every value exact.  At 44.1,
    22.05 and 11.025 kHz frames are not whole 4-sample vectors (1323 / 441, 661 / 220, 330 / 110 samples): the kernel adds
    up to three samples in front of and behind a frame's whole vectors one by one."""
    from features.batch import EndpointPlan
    ep = EndpointPlan(rate, 0.03, 0.01)
    rng = np.random.default_rng(5)
    n = 2 * rate + 6
    clips = [np.full(n, -32768, dtype=np.int16),
             np.where(np.arange(n) % 2 == 0, 32767, -32768).astype(np.int16),
             np.where(np.arange(n) % 3 == 0, 0, np.where(np.arange(n) % 2 == 0, 1, -1)).astype(np.int16),
             rng.integers(-32768, 32768, n // 2 + 1).astype(np.int16),
             rng.integers(-2, 3, ep.L + 3 * ep.S + 1).astype(np.int16),
             np.zeros(0, dtype=np.int16),                       # an empty clip is one frame of zeros (sigproc.py:11-19)
             np.array([5], dtype=np.int16),
             np.zeros(0, dtype=np.int16)]                       # ... and so is an empty LAST clip: no sample address at all
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    _, amp, zcr, fo = ep.detect_batch(np.concatenate(clips), sample_offsets=so, return_feature=True)
    for b, c in enumerate(clips):
        frames = dsp_oracle.to_frames(c.astype(np.float64), rate, t=0.03, step=0.01)
        assert fo[b + 1] - fo[b] == frames.shape[0]
        assert np.array_equal(amp[fo[b]:fo[b + 1]], dsp_oracle.get_amplitude(frames)), b
        assert list(zcr[fo[b]:fo[b + 1]]) == list(dsp_oracle.get_zcr(frames)), b


@pytest.mark.parametrize('copy_trimmed', [False, True], ids=['in_place', 'trimmed_copy'])
@pytest.mark.parametrize('unit_variance', [False, True])
def test_config4_vad_trim_mfcc_pipeline(unit_variance, copy_trimmed):
    """configs[3]: endpointing -> trim (-> unit variance, model.py:63) -> MFCC+delta+delta2 with
    variable-length outputs, against the oracle run utterance by utterance.  Both forms of the feature stage:
    sig[left:right] read in place (unit variance as a shift of c0 from in-kernel fp64 sums,
    dsp_mfcc_delta_segments_batch) and the trimmed / scaled fp32 copy (dsp_trim_scale_batch)."""
    from features.pipeline import VadMfccPipeline
    from golden_cases import make_signal
    clips = [make_signal(('vad', 200 + i, 16000 + 1700 * i)) for i in range(10)]
    clips.append(make_signal(('int16', 77, 9000)))      # no burst: whole-clip fallback
    clips.append(make_signal(('bursts', 78, 32000)))
    clips.append((make_signal(('vad', 300, 20000)).astype(np.int32) + 6000).clip(-32768, 32767).astype(np.int16))  # DC offset >> noise floor
    hole = make_signal(('vad', 301, 24000)).copy()     # 700 samples of digital silence INSIDE the detected segment: frames
    mid = int(np.argmax(np.abs(hole.astype(np.int32))))   # of zero energy keep ln(eps) while their neighbours' c0 shifts
    hole[mid - 350:mid + 350] = 0                      # by ln(var) -- the deltas of c0 around them must see that
    clips.append(hole)
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    pipe = VadMfccPipeline(rate=16000, unit_variance=unit_variance, winfunc=np.hamming,
                           **{k: v for k, v in CFG.items() if k != 'samplerate'})
    pipe.copy_trimmed = copy_trimmed
    out, fo, ends = pipe.run(np.concatenate(clips), so, delta_n=2)
    assert fo[-1] == out.shape[0] and out.shape[1] == 39
    for b, c in enumerate(clips):
        lo, hi = dsp_oracle.basic_endpoint_detection(c, 16000)
        assert (lo, hi) == (int(ends[b, 0]), int(ends[b, 1])) or (lo, min(hi, len(c))) == tuple(ends[b])
        seg = np.asarray(c[lo:hi], dtype=np.float64)
        if unit_variance:
            seg = dsp_oracle.model_endpoint_scale(c, lo, hi).reshape(-1)
        ref = dsp_oracle.mfcc_delta(seg, delta_n=2, winfunc=np.hamming, **CFG)
        got = out[fo[b]:fo[b + 1]]
        assert got.shape == ref.shape, (b, got.shape, ref.shape)
        assert normwise(got, ref) <= TOL, (b, normwise(got, ref))


@pytest.mark.parametrize('cfg', [
    dict(winlen=0.03, winstep=0.01, nfilt=48, numcep=16),      # L=480 -> 30 rows, 6 mel groups: catch-all kernel
    dict(winlen=0.032, winstep=0.008, nfilt=64, numcep=13),    # L=512 (= NFFT), S=128; 64 filters: generic kernel
    dict(winlen=0.032, winstep=0.008, nfilt=47, numcep=13),    # L=512, odd filter count, 6 mel iterations
    dict(winlen=0.02, winstep=0.005, nfilt=26, numcep=12),     # L=320, S=80: 26-filter instantiation
    dict(winlen=0.025, winstep=0.0101, nfilt=40, numcep=13),   # S=162 (even, not /4)
    dict(winlen=0.025, winstep=0.01, nfilt=40, numcep=13, lowfreq=300, highfreq=3400, preemph=0.0),
], ids=['L480_M48_C16', 'L512_M64', 'L512_M47', 'L320_M26_C12', 'S162', 'band_nopre'])
@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_fast_kernel_instantiations(cfg, dtype):
    """Every <rows, mel groups, cepstra> instantiation of the specialised kernel, dense and ragged,
    fp32 and int16, against the oracle."""
    from features.batch import FeaturePlan
    from features import _native as nat
    full = dict(samplerate=16000, nfft=512, lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
    full.update(cfg)
    plan = FeaturePlan(winfunc=np.hamming, **full)
    # 64 filters on a 257-bin spectrum: the lowest ones are 1-2 bins wide (the first is the DC bin alone) and their
    # logs amplify the fp32 noise floor of single bins; the fused kernel measured 1.5e-4 there, the generic one
    # 5e-5, so plans with more than 48 filters are served by the generic kernel and every plan meets the plain bar
    assert nat.load().dsp_plan_has_fast_path(plan.plan.handle) == (1 if full['nfilt'] <= 48 else 0)
    tol = TOL
    dense = _batch(51, 24, 8000, dtype=dtype)
    out, fo = plan.mfcc_batch(dense, delta_n=2)
    lens = [8000, 513, 4097, 1, 7999, 12001, 640]
    so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    flat = _batch(52, 1, int(so[-1]), dtype=dtype)[0]
    out_r, fo_r = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
    for b in range(24):
        ref = dsp_oracle.mfcc_delta(dense[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **full)
        from conftest import record
        err = record(f"fast512_nfilt{full['nfilt']}", normwise(out[fo[b]:fo[b + 1]], ref))
        assert err <= tol, ('dense', b, err)
    for b in range(len(lens)):
        ref = dsp_oracle.mfcc_delta(flat[so[b]:so[b + 1]].astype(np.float64), delta_n=2, winfunc=np.hamming, **full)
        assert out_r[fo_r[b]:fo_r[b + 1]].shape == ref.shape
        err = record(f"fast512_nfilt{full['nfilt']}", normwise(out_r[fo_r[b]:fo_r[b + 1]], ref))
        assert err <= tol, ('ragged', b, lens[b], err)


@pytest.mark.parametrize('cfg', [
    dict(samplerate=48000, winlen=0.03, winstep=0.01, nfilt=26, numcep=13),                 # model.py:74 at 48 kHz
    dict(samplerate=44100, winlen=0.03, winstep=0.01, nfilt=26, numcep=13, preemph=0.0),    # L=1323, S=441 (odd)
    dict(samplerate=48000, winlen=0.032, winstep=0.012, nfilt=40, numcep=16),               # L=1536, 3 mel groups: catch-all
    dict(samplerate=16000, winlen=0.03, winstep=0.01, nfilt=26, numcep=13),                 # L=480 zero-padded to 1536
], ids=['48k', '44k1_nopre', 'L1536_M40_C16', '16k_L480'])
@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_fast1536_kernel(cfg, dtype):
    """The specialised NFFT=1536 kernel (radix-3 split into a real and a complex 512-point FFT),
    dense and ragged, fp32 and int16, against the oracle and against the generic kernel."""
    from features.batch import FeaturePlan
    from features import _native as nat
    full = dict(nfft=1536, lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
    full.update(cfg)
    plan = FeaturePlan(winfunc=np.hamming, **full)
    lib = nat.load()
    assert lib.dsp_plan_has_fast_path(plan.plan.handle) == 1
    rate = full['samplerate']
    dense = _batch(61, 10, rate // 2, dtype=dtype)
    out, fo = plan.mfcc_batch(dense, delta_n=3)
    lens = [rate // 2, 1537, 4097, 1, rate // 3 + 1, 20001, 1440]
    so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    flat = _batch(62, 1, int(so[-1]), dtype=dtype)[0]
    out_r, fo_r = plan.mfcc_batch(flat, sample_offsets=so, delta_n=3)
    try:
        nat.check(lib.dsp_debug_force_generic(1))
        gen, _ = plan.mfcc_batch(dense, delta_n=3)
        gen_r, _ = plan.mfcc_batch(flat, sample_offsets=so, delta_n=3)
    finally:
        nat.check(lib.dsp_debug_force_generic(0))
    assert normwise(out, gen) <= 5e-5 and normwise(out_r, gen_r) <= 5e-5
    for b in range(10):
        ref = dsp_oracle.mfcc_delta(dense[b].astype(np.float64), delta_n=3, winfunc=np.hamming, **full)
        assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOL, ('dense', b)
    for b in range(len(lens)):
        ref = dsp_oracle.mfcc_delta(flat[so[b]:so[b + 1]].astype(np.float64), delta_n=3, winfunc=np.hamming, **full)
        assert out_r[fo_r[b]:fo_r[b + 1]].shape == ref.shape
        assert normwise(out_r[fo_r[b]:fo_r[b + 1]], ref) <= TOL, ('ragged', b, lens[b])


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_dense_batches_with_odd_length_take_the_fused_kernels(dtype):
    """[B, N] batches whose N is not a multiple of 4 are handed to the fused kernels as ragged
    batches with arithmetic offsets (built on the device): same numbers as the table-driven kernels,
    and as fast as any other ragged batch.  MFCC at NFFT 512 and 1536, and the VAD features."""
    from features.batch import EndpointPlan, FeaturePlan
    from features import _native as nat
    lib = nat.load()
    for cfg, n in ((dict(CFG), 16001), (dict(samplerate=48000, winlen=0.03, winstep=0.01, numcep=13, nfilt=26,
                                              nfft=1536, preemph=0.97, ceplifter=22, appendEnergy=True), 24003)):
        plan = FeaturePlan(winfunc=np.hamming, **cfg)
        x = _batch(71, 9, n, dtype=dtype)
        out, fo = plan.mfcc_batch(x, delta_n=2)
        try:
            nat.check(lib.dsp_debug_force_generic(1))
            gen, _ = plan.mfcc_batch(x, delta_n=2)
        finally:
            nat.check(lib.dsp_debug_force_generic(0))
        assert normwise(out, gen) <= 5e-5
        assert not np.array_equal(out, gen)            # really two different kernels
        for b in (0, 8):
            ref = dsp_oracle.mfcc_delta(x[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **cfg)
            assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOL
    ep = EndpointPlan(16000, 0.03, 0.01)
    x = _batch(72, 7, 16001, dtype=dtype)
    _, amp, zcr, fo = ep.detect_batch(x, return_feature=True)
    for b in range(7):
        frames = dsp_oracle.to_frames(x[b].astype(np.float64), 16000, t=0.03, step=0.01)
        assert np.allclose(amp[fo[b]:fo[b + 1]], dsp_oracle.get_amplitude(frames), rtol=1e-6, atol=1e-9)
        assert list(zcr[fo[b]:fo[b + 1]]) == list(dsp_oracle.get_zcr(frames))


def test_odd_hop_uses_generic_kernel_and_matches():
    from features.batch import FeaturePlan
    from features import _native as nat
    full = dict(samplerate=16000, nfft=512, winlen=0.025, winstep=161 / 16000.0, nfilt=40, numcep=13)
    plan = FeaturePlan(winfunc=np.hamming, **full)
    assert nat.load().dsp_plan_has_fast_path(plan.plan.handle) == 0      # S = 161 is odd
    x = _batch(53, 3, 8000)
    out, fo = plan.mfcc_batch(x, delta_n=1)
    ref = dsp_oracle.mfcc_delta(x[1].astype(np.float64), delta_n=1, winfunc=np.hamming, **full)
    assert normwise(out[fo[1]:fo[2]], ref) <= TOL


def test_pipeline_without_a_host_round_trip():
    """configs[3] queued as one asynchronous sequence: a prepared layout, a device tensor in, device
    buffers out (download=False) -- and the result, read back afterwards, equals the host-array path.
    Training-time endpoint jitter (model.py:54-60) goes through the same device-side layout kernel."""
    import random
    import torch
    from features.pipeline import VadMfccPipeline
    from golden_cases import make_signal
    clips = [make_signal(('vad', 300 + i, 16000 + 2100 * i)) for i in range(9)]
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    flat = np.concatenate(clips)
    pipe = VadMfccPipeline(rate=16000, unit_variance=True, winfunc=np.hamming,
                           **{k: v for k, v in CFG.items() if k != 'samplerate'})
    ref_out, ref_fo, ref_ends = pipe.run(flat, so, delta_n=2)
    lay = pipe.prepare(so, delta_n=2)
    d_flat = torch.from_numpy(flat).cuda()
    for _ in range(3):          # the layout's buffers are reused call after call
        (d_out, lay2), fo, ends = pipe.run(d_flat, layout=lay, download=False)
        assert lay2 is lay and fo is None and ends is None
    fo = lay.d_frame_off.download((len(clips) + 1,), np.int64)
    assert np.array_equal(fo, ref_fo) and fo[-1] <= lay.frames_bound
    assert np.array_equal(lay.d_seg.download((len(clips), 2), np.int64), ref_ends)
    got = d_out.download((int(fo[-1]), 39), np.float32)
    assert np.array_equal(got, ref_out)
    # jitter: every utterance against the oracle's augment=True branch with the same draws
    rng = random.Random(11)
    jit = np.array([[-rng.randint(0, 1600), rng.randint(0, 1600)] for _ in clips], dtype=np.int64)
    out_j, fo_j, ends_j = pipe.run(flat, so, delta_n=2, jitter=jit)
    for b, c in enumerate(clips):
        lo, hi = dsp_oracle.basic_endpoint_detection(c, 16000)
        lo, hi = max(lo + int(jit[b, 0]), 0), max(hi + int(jit[b, 1]), 0)
        assert (min(lo, len(c)), min(hi, len(c))) == tuple(ends_j[b])
        seg = dsp_oracle.model_endpoint_scale(c, lo, hi).reshape(-1)
        ref = dsp_oracle.mfcc_delta(seg, delta_n=2, winfunc=np.hamming, **CFG)
        assert normwise(out_j[fo_j[b]:fo_j[b + 1]], ref) <= TOL, b


def test_two_threads_calling_the_drop_in_do_not_share_scratch():
    """The drop-in functions keep their device scratch per thread: two threads calling mfcc on
    different signals at the same time each get their own result."""
    import threading
    import features
    rng = np.random.default_rng(5)
    sigs = [(0.25 * rng.standard_normal(16000 + 3000 * i)).astype(np.float32) for i in range(2)]
    want = [features.mfcc(s, winfunc=np.hamming, **CFG) for s in sigs]
    errs = []

    def work(i):
        try:
            for _ in range(40):
                got = features.mfcc(sigs[i], winfunc=np.hamming, **CFG)
                if not np.array_equal(got, want[i]):
                    errs.append((i, float(np.max(np.abs(got - want[i])))))
                    return
        except Exception as e:      # noqa: BLE001 - reported to the main thread
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_pipeline_launch_does_not_wait_for_the_device():
    """configs[3] is a pure sequence of asynchronous launches: with ~0.2 s of other work already queued on
    the stream, queueing the whole pipeline behind it returns long before that work has finished (any host
    synchronisation, blocking copy or allocation inside would wait for it)."""
    import time
    import torch
    from features import _native as nat
    from features.pipeline import VadMfccPipeline
    from golden_cases import make_signal
    clips = [make_signal(('vad', 400 + i, 16000 + 900 * i)) for i in range(64)]
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    d_flat = torch.from_numpy(np.concatenate(clips)).cuda()
    pipe = VadMfccPipeline(rate=16000, unit_variance=True, winfunc=np.hamming,
                           **{k: v for k, v in CFG.items() if k != 'samplerate'})
    lay = pipe.prepare(so, delta_n=2)
    out = torch.empty((lay.frames_bound, lay.D), device='cuda')
    st = torch.cuda.current_stream()
    for _ in range(3):                                    # warm: pooled workspaces, code objects
        pipe.launch(d_flat.data_ptr(), nat.WAVE_I16, lay, out.data_ptr(), st)
    torch.cuda.synchronize()
    ref = out.clone()
    out.zero_()
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    t0 = time.perf_counter()
    torch.cuda._sleep(int(4e8))                           # ~0.2 s of device time ahead of the pipeline
    pipe.launch(d_flat.data_ptr(), nat.WAVE_I16, lay, out.data_ptr(), st)
    t_queue = time.perf_counter() - t0
    done.record(st)
    still_running = not done.query()
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    assert still_running and t_queue < 0.25 * t_total, (t_queue, t_total)
    fo = lay.d_frame_off.download((len(clips) + 1,), np.int64)
    assert torch.equal(out[:int(fo[-1])], ref[:int(fo[-1])])


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_flat_grouping_across_utterance_seams(dtype):
    """Dense batches cut their 8-frame groups from the flat frame sequence (kernels_fast512.h): a group may
    hold the last frames of one utterance and the first of the next.  Every frame count around the group
    size, batch sizes that end inside a group, against the oracle and against single-utterance calls."""
    from features.batch import FeaturePlan
    plan = FeaturePlan(winfunc=np.hamming, **CFG)
    for T, B in ((7, 5), (8, 3), (9, 4), (15, 2), (17, 33), (23, 1), (99, 6)):
        N = 400 + 160 * (T - 1) - 40          # T frames, the last one zero padded; multiple of 4
        assert N % 4 == 0
        waves = _batch(300 + T, B, N, dtype=dtype)
        out, fo = plan.mfcc_batch(waves, delta_n=2)
        assert fo[-1] == B * T and out.shape == (B * T, 39)
        for b in range(B):
            ref = dsp_oracle.mfcc_delta(waves[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
            assert ref.shape[0] == T
            assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOL, (T, B, b)
        single = np.concatenate([plan.mfcc_batch(waves[b:b + 1], delta_n=2)[0] for b in range(B)])
        assert np.array_equal(out, single), (T, B)       # where a frame sits in its group changes nothing
