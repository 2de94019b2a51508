"""BASELINE.json full sizes on one MI355X, checked through size-independent properties
(SURVEY 8c/8d): configs[2]'s per-GPU share (12 500 x 1 s utterances = 1 237 500 frames, 0.8 GB of
fp32 waveform) and a configs[3]-sized ragged batch (12 500 utterances of 1-2 s, VAD -> trim ->
MFCC).  The oracle only sees a sample of the utterances; the rest is covered by

  * tiling invariance  -- the batch is a few thousand distinct utterances repeated, every repeat
    must be bitwise identical (an utterance's features do not depend on where it sits);
  * shard invariance   -- the first 1024 utterances equal a separate 1024-utterance launch bitwise;
  * a checksum of checksums -- per-utterance fp64 sums of the big launch equal those of the
    small launches they were first computed in;
  * delta linearity    -- the delta columns of the output equal base.delta applied to its own
    static columns (oracle arithmetic on the GPU's cepstra), on sampled utterances.
"""
import numpy as np
import pytest

from conftest import normwise
from oracle import dsp_oracle

pytestmark = pytest.mark.gpu

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, lowfreq=0,
           highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
TOL = 1e-4
SHARE = 12500   # 100 000 utterances / 8 GPUs
DISTINCT = 2500


@pytest.fixture(scope='module')
def plan():
    from features.batch import FeaturePlan
    return FeaturePlan(winfunc=np.hamming, **CFG)


def test_config3_share_dense(plan):
    rng = np.random.default_rng(3003)
    base = (0.25 * rng.standard_normal((DISTINCT, 16000))).astype(np.float32)
    reps = SHARE // DISTINCT
    waves = np.tile(base, (reps, 1))
    out, fo = plan.mfcc_batch(waves, delta_n=2)
    T = 99
    assert out.shape == (SHARE * T, 39) and fo[-1] == SHARE * T
    assert np.isfinite(out).all()
    out = out.reshape(reps, DISTINCT * T, 39)
    for r in range(1, reps):
        assert np.array_equal(out[r], out[0]), r
    first, _ = plan.mfcc_batch(base[:1024], delta_n=2)
    assert np.array_equal(out[0][:1024 * T], first)
    # ... and so do 8 utterances on their own, which take the OTHER form of the step (MFCC kernel + delta rows kernel;
    # the big launches are the one fused kernel): the rows do not depend on which kernels produced them
    few, _ = plan.mfcc_batch(base[:8], delta_n=2)
    assert np.array_equal(out[0][:8 * T], few)
    sums_big = out[0].reshape(DISTINCT, T * 39).astype(np.float64).sum(axis=1)
    small, _ = plan.mfcc_batch(base, delta_n=2)
    sums_small = small.reshape(DISTINCT, T * 39).astype(np.float64).sum(axis=1)
    assert np.array_equal(sums_big, sums_small)
    for b in (0, 1023, 1024, DISTINCT - 1):
        ref = dsp_oracle.mfcc_delta(base[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
        got = out[0][b * T:(b + 1) * T]
        assert normwise(got, ref) <= TOL, b
        d1 = dsp_oracle.delta(got[:, :13].astype(np.float64), 2)
        assert normwise(got[:, 13:26], d1) <= 1e-6, b
        assert normwise(got[:, 26:], dsp_oracle.delta(d1, 2)) <= 1e-6, b


def test_config4_share_ragged():
    """12 500 variable-length int16 utterances (class C of SURVEY 8d) through VAD -> trim ->
    MFCC+delta in one pipeline object; repeats must agree bitwise, sampled ones with the oracle."""
    from features.pipeline import VadMfccPipeline
    rng = np.random.default_rng(4004)
    distinct = 500
    sigs = []
    for u in range(distinct):
        n = int(rng.uniform(1.0, 2.0) * 16000)
        x = rng.normal(0, 30, n)
        blen = int(rng.uniform(0.5, 0.9) * n)
        b0 = int(rng.integers(0, n - blen))
        t = np.arange(blen) / 16000.0
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * rng.uniform(100, 300) * t) * np.hanning(blen)
        sigs.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    reps = SHARE // distinct
    lens = np.array([len(s) for s in sigs] * reps)
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.tile(np.concatenate(sigs), reps)
    pipe = VadMfccPipeline(rate=16000, unit_variance=False, winfunc=np.hamming,
                           **{k: v for k, v in CFG.items() if k != 'samplerate'})
    feats, fo, ends = pipe.run(flat, so, delta_n=2)
    assert len(fo) == SHARE + 1 and ends.shape == (SHARE, 2)
    assert np.isfinite(feats).all()
    per = fo[distinct]
    assert fo[-1] == per * reps
    for r in range(1, reps):
        assert np.array_equal(ends[r * distinct:(r + 1) * distinct], ends[:distinct]), r
        assert np.array_equal(feats[r * per:(r + 1) * per], feats[:per]), r
    for b in (0, 1, distinct // 2, distinct - 1):
        l, r = dsp_oracle.basic_endpoint_detection(sigs[b], 16000)
        assert (int(ends[b, 0]), int(ends[b, 1])) == (l, min(r, len(sigs[b]))), b
        ref = dsp_oracle.mfcc_delta(sigs[b][l:r].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
        assert normwise(feats[fo[b]:fo[b + 1]], ref) <= TOL, b


def test_config3_count_ragged_100k(plan):
    """configs[2]'s utterance count in ONE ragged launch: 100 000 utterances of 1..9000 samples
    (200 distinct clips repeated; group / tile tables with 100 001 entries).  Every repeat must equal
    the first bitwise, and the first must equal a separate 200-utterance launch."""
    rng = np.random.default_rng(2002)
    distinct, reps = 200, 500
    sigs = [np.round(3000 * rng.standard_normal(int(rng.integers(1, 9000)))).astype(np.int16) for _ in range(distinct)]
    lens = np.array([len(s) for s in sigs] * reps)
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.tile(np.concatenate(sigs), reps)
    out, fo = plan.mfcc_batch(flat, sample_offsets=so, delta_n=2)
    assert len(fo) == distinct * reps + 1 and out.shape == (fo[-1], 39) and np.isfinite(out).all()
    per = fo[distinct]
    for r in range(1, reps):
        assert np.array_equal(out[r * per:(r + 1) * per], out[:per]), r
    small, _ = plan.mfcc_batch(np.concatenate(sigs), sample_offsets=so[:distinct + 1], delta_n=2)
    assert np.array_equal(small, out[:per])
    b = 17
    ref = dsp_oracle.mfcc_delta(sigs[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **CFG)
    assert normwise(out[fo[b]:fo[b + 1]], ref) <= TOL


def test_nfft1536_large_batch_tiling_invariance():
    """Row f-2 at scale: 2 000 x 1 s utterances at 48 kHz (NFFT = 1536, 26 mel; 196 000 frames, every
    wave of the persistent grid runs many rounds): repeats bitwise identical, first 64 equal a separate
    64-utterance launch, sampled utterances within 1e-4 of the oracle."""
    from features.batch import FeaturePlan
    cfg = dict(samplerate=48000, winlen=0.03, winstep=0.01, numcep=13, nfilt=26, nfft=1536, lowfreq=0,
               highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
    plan = FeaturePlan(winfunc=np.hamming, **cfg)
    rng = np.random.default_rng(1536)
    distinct, reps, n = 250, 8, 48000
    base = (0.25 * rng.standard_normal((distinct, n))).astype(np.float32)
    out, fo = plan.mfcc_batch(np.tile(base, (reps, 1)), delta_n=3)
    T = 98
    assert out.shape == (distinct * reps * T, 39) and np.isfinite(out).all()
    out = out.reshape(reps, distinct * T, 39)
    for r in range(1, reps):
        assert np.array_equal(out[r], out[0]), r
    first, _ = plan.mfcc_batch(base[:64], delta_n=3)
    assert np.array_equal(out[0][:64 * T], first)
    for b in (0, 63, 64, distinct - 1):
        ref = dsp_oracle.mfcc_delta(base[b].astype(np.float64), delta_n=3, winfunc=np.hamming, **cfg)
        assert normwise(out[0][b * T:(b + 1) * T], ref) <= TOL, b
