"""GPU parity of the matrix-pipe NFFT = 512 kernels (csrc/kernels_mfma512.h: 16 frames per product, mode 1;
csrc/kernels_mfma512t.h: one frame per product, mode 2; both selected through dsp_debug_use_mfma512).
The path replaces sigproc.preemphasis / framesig / powspec and base.fbank / mfcc / delta (sigproc.py:66-98,136-158,
178-185; base.py:8-32,70-79) for dense batches; the oracle is the checker, the vector-pipe kernel a second opinion."""
import numpy as np
import pytest

from conftest import normwise, record
from oracle import dsp_oracle
import golden_cases as gc

pytestmark = pytest.mark.gpu

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, lowfreq=0,
           highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
TOL = 1e-4
KINDS = ('white', 'tone', 'harmonic', 'siltail', 'vadf', 'ramp', 'zeros', 'uniform')


MODES = (1, 2)


class _Mfma:
    """Routes the calling thread's dense MFCC calls to a matrix-pipe kernel for the duration of a with-block."""

    def __init__(self, mode=1):
        self.mode = mode

    def __enter__(self):
        from features import _native as nat
        nat.check(nat.load().dsp_debug_use_mfma512(self.mode))

    def __exit__(self, *a):
        from features import _native as nat
        nat.check(nat.load().dsp_debug_use_mfma512(-1))


def _batch(B, N, seed=3, dtype=np.float32):
    """The first utterances are the golden signal kinds (tone-like, silent tails, a ramp whose pre-emphasised samples
    are 30x smaller than the raw ones, digital zero); the rest white noise over four decades of amplitude."""
    rng = np.random.default_rng(seed)
    x = np.empty((B, N), np.float64)
    for b in range(B):
        if b < 2 * len(KINDS):
            x[b] = np.asarray(gc.make_signal((KINDS[b % len(KINDS)], 100 + b, N)), np.float64)[:N]
        else:
            x[b] = 0.25 * rng.standard_normal(N) * (10.0 ** rng.uniform(-3, 1))
    if dtype == np.int16:
        return np.clip(np.round(x * 3000), -32768, 32767).astype(np.int16)
    return x.astype(np.float32)


def _plan(**over):
    from features.batch import FeaturePlan
    return FeaturePlan(winfunc=np.hamming, **dict(CFG, **over))


def _served(plan, mode=1):
    from features import _native as nat
    return (nat.load().dsp_plan_has_mfma512(plan.plan.handle) >> (mode - 1)) & 1 == 1


def _worst(got, fo, waves, cfg, delta_n, idx):
    worst = 0.0
    for b in idx:
        x = waves[b].astype(np.float64)
        ref = dsp_oracle.mfcc_delta(x, delta_n=delta_n, winfunc=np.hamming, **cfg) if delta_n else \
            dsp_oracle.mfcc(x, winfunc=np.hamming, **cfg)
        g = got[fo[b]:fo[b + 1]]
        assert g.shape == ref.shape and np.isfinite(g).all(), b
        worst = max(worst, normwise(g, ref))
    return worst


@pytest.mark.parametrize('name,over,B,N,delta_n,dtype', [
    ('configs1_rows', {}, 1024, 16000, 2, np.float32),
    ('configs1_cepstra', {}, 1024, 16000, 0, np.float32),
    ('configs1_int16', {}, 1024, 16000, 2, np.int16),
    ('nfilt26', dict(nfilt=26), 1024, 16000, 2, np.float32),
    ('odd_length', {}, 777, 12345, 2, np.float32),          # N not a multiple of anything: partial last loads
    ('delta1_long', {}, 300, 48000, 1, np.float32),
    ('winlen20ms', dict(winlen=0.02), 640, 16000, 2, np.float32),
    ('no_energy_no_lifter', dict(appendEnergy=False, ceplifter=0), 600, 16000, 2, np.float32),
    ('numcep3_nfilt13', dict(numcep=3, nfilt=13), 600, 16000, 2, np.float32),      # one row tile of filters, partial quads
    ('numcep15', dict(numcep=15), 600, 16000, 1, np.float32),
])
@pytest.mark.parametrize('mode', MODES)
def test_matrix_pipe_kernel_vs_oracle(mode, name, over, B, N, delta_n, dtype):
    cfg = dict(CFG, **over)
    plan = _plan(**over)
    assert _served(plan, mode)
    waves = _batch(B, N, dtype=dtype)
    with _Mfma(mode):
        got, fo = plan.mfcc_batch(waves, delta_n=delta_n)
    idx = list(range(0, 2 * len(KINDS))) + list(range(2 * len(KINDS), B, max(1, B // 24))) + [B - 1]
    worst = record(('mfma512_' if mode == 1 else 'mfma512t_') + name, _worst(got, fo, waves, cfg, delta_n, idx))
    assert worst <= TOL, worst


@pytest.mark.parametrize('mode', MODES)
def test_matrix_pipe_and_vector_pipe_kernels_agree(mode):
    """Two independent implementations of the same path (fp16-split matrix products vs in-register fp32 butterflies)."""
    plan = _plan()
    waves = _batch(1024, 16000, seed=9)
    vec, fo = plan.mfcc_batch(waves, delta_n=2)
    with _Mfma(mode):
        mat, fo2 = plan.mfcc_batch(waves, delta_n=2)
    assert np.array_equal(fo, fo2)
    worst = max(normwise(mat[fo[b]:fo[b + 1]], vec[fo[b]:fo[b + 1]]) for b in range(0, 1024, 7))
    assert record('mfma512_vs_vector_pipe' if mode == 1 else 'mfma512t_vs_vector_pipe', worst) <= TOL


@pytest.mark.parametrize('mode', MODES)
def test_matrix_pipe_results_do_not_depend_on_the_batch(mode):
    """A wave owns whole row ranges of ONE utterance and scales each tile (mode 2: half tile) by its own largest sample,
    so an utterance's rows are the same bits whatever surrounds it -- as long as it is cut into the same ranges: the same
    utterance in two batches of one size, at different positions, next to different neighbours."""
    plan = _plan()
    a = _batch(1024, 16000, seed=21)
    b = a[::-1].copy()
    b[500] = np.nan                                    # a NaN clip must not leak into its neighbours
    with _Mfma(mode):
        ra, fo = plan.mfcc_batch(a, delta_n=2)
        rb, _ = plan.mfcc_batch(b, delta_n=2)
    ra = ra.reshape(1024, 99, 39)
    rb = rb.reshape(1024, 99, 39)[::-1]
    keep = np.ones(1024, bool)
    keep[1023 - 500] = False
    assert np.array_equal(ra[keep], rb[keep])
    assert np.isnan(rb[1023 - 500]).any()


@pytest.mark.parametrize('mode', MODES)
def test_plans_outside_the_kernel_fall_back(mode):
    """More than 47 filters, a hop that is not 160 samples, (mode 1) a filterbank outside the block pattern: the plan
    carries no matrix-pipe tables and the switch changes nothing."""
    waves = _batch(600, 16000, seed=5)
    for over in (dict(nfilt=64), dict(winstep=0.0125)) + ((dict(lowfreq=3000),) if mode == 1 else ()):
        plan = _plan(**over)
        assert not _served(plan, mode), over
        ref, fo = plan.mfcc_batch(waves, delta_n=2)
        with _Mfma(mode):
            got, _ = plan.mfcc_batch(waves, delta_n=2)
        assert np.array_equal(got, ref), over


@pytest.mark.parametrize('N,B', [(300, 40000), (401, 36000), (1000, 8000), (2965, 2500)])
@pytest.mark.parametrize('mode', MODES)
def test_short_utterances(mode, N, B):
    """One frame with zero padding (N < L), two frames, a handful: the first tile is also the last, the delta windows
    clamp on both sides, most of a tile's columns are padding."""
    plan = _plan()
    rng = np.random.default_rng(N)
    waves = (0.25 * rng.standard_normal((B, N))).astype(np.float32)
    with _Mfma(mode):
        got, fo = plan.mfcc_batch(waves, delta_n=2)
    idx = list(range(0, B, max(1, B // 40))) + [B - 1]
    assert record(f'mfma512{"t" if mode == 2 else ""}_short_{N}', _worst(got, fo, waves, CFG, 2, idx)) <= TOL


@pytest.mark.parametrize('N,B', [(401, 3000), (2966, 2500), (16001, 1024)])
@pytest.mark.parametrize('mode', MODES)
def test_int16_lengths_the_descriptor_cannot_read(mode, N, B):
    """int16 utterances of odd length (and one even, 4 k + 2): the kernels read whole dwords through a bounds-checked
    descriptor, so an odd length would lose its last sample -- such batches go to the vector-pipe path instead."""
    plan = _plan()
    rng = np.random.default_rng(N)
    waves = np.clip(np.round(3000 * rng.standard_normal((B, N))), -32768, 32767).astype(np.int16)
    with _Mfma(mode):
        got, fo = plan.mfcc_batch(waves, delta_n=2)
    idx = list(range(0, B, max(1, B // 24))) + [B - 1]
    assert record(f'mfma512{"t" if mode == 2 else ""}_int16_{N}', _worst(got, fo, waves, CFG, 2, idx)) <= TOL


@pytest.mark.parametrize('mode', MODES)
def test_extreme_amplitudes_and_silence(mode):
    """The tile scale is a power of two from the tile's largest sample: 1e-30 ... 1e30, exact zeros (the eps path of
    base.py:26,30) and a clip that is silent except for one sample."""
    plan = _plan()
    B, N = 600, 16000
    rng = np.random.default_rng(4)
    waves = (0.25 * rng.standard_normal((B, N))).astype(np.float32)
    amps = [1e-30, 1e-20, 1e-10, 1e10, 1e20, 1e30]
    for i, a in enumerate(amps):
        waves[i] *= np.float32(a)
    waves[10] = 0.0
    waves[11] = 0.0
    waves[11, 7777] = 1.0
    waves[12, 8000:] = 0.0
    with _Mfma(mode):
        got, fo = plan.mfcc_batch(waves, delta_n=2)
    assert np.isfinite(got).all()
    assert record('mfma512_extremes' if mode == 1 else 'mfma512t_extremes', _worst(got, fo, waves, CFG, 2, list(range(16)) + [B - 1])) <= TOL


@pytest.mark.parametrize('mode', MODES)
def test_random_plans(mode):
    """Twenty random plans inside the kernel's envelope (window length and shape, filter count and band, cepstra kept,
    lifter, pre-emphasis, energy swap): wherever the library builds matrix-pipe tables for a plan, the kernel must
    reproduce the oracle; a plan it refuses must fall back without a trace."""
    from features import _native as nat
    rng = np.random.default_rng(20260)
    wins = {'hamming': np.hamming, 'hanning': np.hanning, 'ones': dsp_oracle._ones}
    B, N = 400, 16000
    waves = _batch(B, N, seed=77)
    served = 0
    for k in range(20):
        nfilt = int(rng.integers(13, 48))
        cfg = dict(samplerate=16000, winlen=float(rng.choice([0.01, 0.016, 0.02, 0.025, 0.03, 0.032])), winstep=0.01,
                   numcep=int(rng.integers(1, min(16, nfilt) + 1)), nfilt=nfilt, nfft=512,
                   lowfreq=float(rng.choice([0, 0, 50, 300])), highfreq=rng.choice([None, None, 7000.0, 4000.0]),
                   preemph=float(rng.choice([0.97, 0.95, 0.5, 0.0])), ceplifter=int(rng.choice([22, 0, 10])),
                   appendEnergy=bool(rng.integers(0, 2)))
        wname = str(rng.choice(list(wins)))
        from features.batch import FeaturePlan
        plan = FeaturePlan(winfunc=wins[wname], **cfg)
        delta_n = int(rng.integers(0, 3))
        with _Mfma(mode):
            got, fo = plan.mfcc_batch(waves, delta_n=delta_n)
        if not _served(plan, mode):
            ref, _ = plan.mfcc_batch(waves, delta_n=delta_n)
            assert np.array_equal(got, ref), cfg
            continue
        served += 1
        worst = 0.0
        for b in list(range(0, 16)) + [100, 399]:
            if b < 16 and KINDS[b % len(KINDS)] == 'ramp' and (wname == 'hanning' or cfg['preemph'] < 0.9):
                continue   # a ramp under a Hann window, or without pre-emphasis, has no high-frequency content an fp32
                           # INPUT can carry: measured 8.3e-4 (vector-pipe kernel 6.9e-4) and 1.6e-4 (9.6e-5) -- the fp64
                           # reference alone resolves it; the ramp at the metric's plan is in the parity test above
            x = waves[b].astype(np.float64)
            ref = dsp_oracle.mfcc_delta(x, delta_n=delta_n, winfunc=wins[wname], **cfg) if delta_n else \
                dsp_oracle.mfcc(x, winfunc=wins[wname], **cfg)
            g = got[fo[b]:fo[b + 1]]
            assert g.shape == ref.shape and np.isfinite(g).all(), (cfg, wname, b)
            worst = max(worst, normwise(g, ref))
        assert record('mfma512_random_plans' if mode == 1 else 'mfma512t_random_plans', worst) <= TOL, (cfg, wname, delta_n, worst)
    assert served >= 8, served
