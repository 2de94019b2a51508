"""Shared description of the golden-vector cases (inputs by seed, call + kwargs by name).

Used by ``tests/golden/make_golden.py`` (which runs the *reference* on these cases, in the build
container only) and by the parity tests (which run the oracle and the HIP path on the same cases).
Nothing here touches the reference.
"""
from __future__ import annotations

import numpy as np

WINFUNCS = {
    'ones': lambda n: np.ones((n,)),
    'hamming': np.hamming,
    'hanning': np.hanning,
}


def make_signal(spec):
    """spec = (kind, seed, n[, extra]) -> 1-D numpy array (float64 unless the kind says int16)."""
    kind, seed, n = spec[0], spec[1], spec[2]
    rng = np.random.default_rng(1_000_003 * 7 + seed)
    t = np.arange(n) / 16000.0
    if kind == 'white':
        return 0.25 * rng.standard_normal(n)
    if kind == 'white32':  # exactly representable in fp32 (what the GPU bench feeds)
        return (0.25 * rng.standard_normal(n)).astype(np.float32).astype(np.float64)
    if kind == 'uniform':
        return rng.uniform(-1.0, 1.0, n)
    if kind == 'int16':
        return np.clip(np.round(3000.0 * rng.standard_normal(n)), -32768, 32767).astype(np.int16)
    if kind == 'tone':
        return np.sin(2 * np.pi * 440.0 * t) + 1e-3 * rng.standard_normal(n)
    if kind == 'harmonic':
        x = np.zeros(n)
        for h in range(1, 30):
            x += np.sin(2 * np.pi * 120.0 * h * t) / h
        return x * np.hanning(n) + 1e-4 * rng.standard_normal(n)
    if kind == 'zeros':
        return np.zeros(n)
    if kind == 'siltail':  # noise followed by digital silence (eps path in some frames)
        x = 0.25 * rng.standard_normal(n)
        x[n // 2:] = 0.0
        return x
    if kind == 'ramp':
        return np.arange(n, dtype=np.float64) / n
    if kind == 'vad':  # class C of SURVEY 8d: int16 background + one hann-shaped voiced burst
        rate = spec[3] if len(spec) > 3 else 16000
        frac = spec[4] if len(spec) > 4 else rng.uniform(0.5, 0.9)
        x = 30.0 * rng.standard_normal(n)
        blen = int(frac * n)
        start = int(rng.integers(0, max(1, n - blen)))
        f0 = rng.uniform(100.0, 300.0)
        tt = np.arange(blen) / float(rate)
        x[start:start + blen] += 8000.0 * np.sin(2 * np.pi * f0 * tt) * np.hanning(blen)
        return np.clip(np.round(x), -32768, 32767).astype(np.int16)
    if kind == 'vadf':  # same, as float64 after a scale (what model.py feeds downstream)
        return make_signal(('vad',) + tuple(spec[1:])).astype(np.float64) / 1000.0
    if kind == 'bursts':  # two short bursts -> multi-segment amplitude_rule output
        x = 20.0 * rng.standard_normal(n)
        for a, b in ((0.15, 0.40), (0.55, 0.85)):
            s, e = int(a * n), int(b * n)
            tt = np.arange(e - s) / 16000.0
            x[s:e] += 6000.0 * np.sin(2 * np.pi * 180.0 * tt) * np.hanning(e - s)
        return np.clip(np.round(x), -32768, 32767).astype(np.int16)
    raise KeyError(kind)


BASE_CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512,
                lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True,
                winfunc='hamming')
MODEL_CFG = dict(samplerate=44100, winlen=0.03, winstep=0.01, numcep=13, nfilt=26, nfft=1536,
                 lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True,
                 winfunc='hamming')


def _fb(cfg):
    return {k: v for k, v in cfg.items() if k not in ('numcep', 'ceplifter', 'appendEnergy')}


CASES = []


def _add(name, fn, sig=None, **kw):
    CASES.append(dict(name=name, fn=fn, sig=sig, kw=kw))


# (1) preemphasis -- 1-D, (1,N) quirk, (2,N) oddity, coeff 0
_add('preemph_1d', 'preemphasis', ('white', 1, 2000), coeff=0.97)
_add('preemph_default', 'preemphasis', ('int16', 2, 2000))
_add('preemph_zero', 'preemphasis', ('white', 3, 500), coeff=0.0)
_add('preemph_row', 'preemphasis', ('white', 4, 2000), coeff=0.97, reshape=(1, -1))
_add('preemph_2rows', 'preemphasis', ('white', 5, 2000), coeff=0.97, reshape=(2, -1))

# (2) framesig -- N<L, N==L, exact multiple, remainder; ones vs hamming; stride both; fractional sizes
for nm, n in (('short', 300), ('equal', 400), ('exact', 400 + 160 * 9), ('rem', 2000)):
    _add(f'framesig_{nm}_ones', 'framesig', ('white', 10, n), frame_len=400, frame_step=160,
         winfunc='ones')
    _add(f'framesig_{nm}_ham', 'framesig', ('white', 11, n), frame_len=400.0, frame_step=160.0,
         winfunc='hamming')
_add('framesig_nostride', 'framesig', ('white', 12, 2000), frame_len=400, frame_step=160,
     winfunc='hamming', stride_trick=False)
_add('framesig_frac', 'framesig', ('white', 13, 3000), frame_len=551.25, frame_step=220.5,
     winfunc='hamming')
_add('framesig_int16', 'framesig', ('int16', 14, 2000), frame_len=400, frame_step=160,
     winfunc='hamming')
_add('to_frames_default', 'to_frames', ('white', 15, 2000), rate=16000)
_add('to_frames_vad', 'to_frames', ('int16', 16, 4000), rate=16000, t=0.03, step=0.01)
_add('to_frames_trunc', 'to_frames', ('white', 17, 6000), rate=22050, t=0.025, step=0.01)

# (3) magspec / powspec / logpowspec -- L<NFFT, L==NFFT, L>NFFT (truncation + warning)
for nm, L, nfft in (('lt', 400, 512), ('eq', 512, 512), ('gt', 600, 512), ('small', 100, 128),
                    ('n1536', 1323, 1536)):
    _add(f'powspec_{nm}', 'powspec', ('white', 20, L * 6), frame_len=L, frame_step=max(1, L // 2),
         NFFT=nfft)
_add('magspec_lt', 'magspec', ('tone', 21, 2400), frame_len=400, frame_step=160, NFFT=512)
_add('logpowspec_norm', 'logpowspec', ('white', 22, 2400), frame_len=400, frame_step=160, NFFT=512,
     norm=1)
_add('logpowspec_nonorm', 'logpowspec', ('siltail', 23, 2400), frame_len=400, frame_step=160,
     NFFT=512, norm=0)

# (4) get_filterbanks
_add('fb_26_512_16k', 'get_filterbanks', None, nfilt=26, nfft=512, samplerate=16000)
_add('fb_40_512_16k', 'get_filterbanks', None, nfilt=40, nfft=512, samplerate=16000)
_add('fb_26_1536_44k', 'get_filterbanks', None, nfilt=26, nfft=1536, samplerate=44100)
_add('fb_26_1536_48k', 'get_filterbanks', None, nfilt=26, nfft=1536, samplerate=48000)
_add('fb_default', 'get_filterbanks', None)
_add('fb_band', 'get_filterbanks', None, nfilt=20, nfft=512, samplerate=16000, lowfreq=300,
     highfreq=3400)
_add('fb_dense', 'get_filterbanks', None, nfilt=64, nfft=256, samplerate=8000)  # repeated edges

# (5) fbank / mfcc at the BASELINE config, the defaults, and model.py:74's config
for kind in ('white', 'white32', 'uniform', 'int16', 'tone', 'harmonic', 'zeros', 'siltail'):
    _add(f'mfcc_base_{kind}', 'mfcc', (kind, 30, 16000), **BASE_CFG)
_add('mfcc_base_short', 'mfcc', ('white', 31, 300), **BASE_CFG)
_add('mfcc_base_odd', 'mfcc', ('white', 32, 12345), **BASE_CFG)
_add('fbank_base_white', 'fbank', ('white', 30, 16000), **_fb(BASE_CFG))
_add('fbank_base_siltail', 'fbank', ('siltail', 30, 16000), **_fb(BASE_CFG))
_add('fbank_base_zeros', 'fbank', ('zeros', 30, 4000), **_fb(BASE_CFG))
_add('mfcc_defaults', 'mfcc', ('white', 33, 16000))
_add('fbank_defaults', 'fbank', ('int16', 34, 8000))
_add('mfcc_noenergy', 'mfcc', ('white', 35, 8000), **dict(BASE_CFG, appendEnergy=False))
_add('mfcc_nolifter', 'mfcc', ('white', 36, 8000), **dict(BASE_CFG, ceplifter=0))
_add('mfcc_26cep20', 'mfcc', ('harmonic', 37, 8000), **dict(BASE_CFG, nfilt=26, numcep=20))
_add('mfcc_band', 'mfcc', ('white', 38, 8000), **dict(BASE_CFG, lowfreq=300, highfreq=3400))
_add('mfcc_22k', 'mfcc', ('white', 39, 22050), **dict(BASE_CFG, samplerate=22050, nfft=1024))
_add('mfcc_model_1d', 'mfcc', ('white', 40, 30000), **MODEL_CFG)
_add('mfcc_model_row', 'mfcc', ('vadf', 41, 30000, 44100, 0.7), reshape=(1, -1), **MODEL_CFG)
_add('mfcc_model_48k', 'mfcc', ('white', 42, 30000), **dict(MODEL_CFG, samplerate=48000))
_add('mfcc_trunc', 'mfcc', ('white', 43, 8000), **dict(BASE_CFG, nfft=256))  # L=400 > nfft

# (6) lifter / delta
_add('lifter_22', 'lifter', ('mfcc_of', 30), L=22)
_add('lifter_0', 'lifter', ('mfcc_of', 30), L=0)
for N in (1, 2, 3):
    _add(f'delta_{N}', 'delta', ('mfcc_of', 30), N=N)
    _add(f'delta2_{N}', 'delta2', ('mfcc_of', 30), N=N)
_add('delta_tiny', 'delta', ('mfcc_of', 31), N=3)  # T=1 < N
_add('delta2_tiny', 'delta2', ('mfcc_of', 31), N=2)

# (7) endpointing (cfg.frame=0.03, cfg.step=0.01)
for i, spec in enumerate((('vad', 50, 25600), ('vad', 51, 19000), ('vad', 52, 32000),
                          ('vad', 53, 16000, 16000, 0.3), ('vadf', 54, 24000), ('bursts', 55, 32000),
                          ('int16', 56, 16000), ('white', 57, 16000), ('vad', 58, 52920, 44100, 0.7),
                          ('vad', 59, 4000))):
    rate = spec[3] if len(spec) > 3 else 16000
    _add(f'endpoint_{i}', 'basic_endpoint_detection', spec, rate=rate)
_add('amplitude_feature', 'amplitude_feature', ('vad', 60, 20000), rate=16000, winlen=0.03, step=0.01)
_add('amp_hamming', 'get_amplitude', ('int16', 61, 3000), frame_len=480, frame_step=160,
     window='hamming')
_add('amp_sq', 'get_amplitude', ('white', 62, 3000), frame_len=480, frame_step=160, use_sq=True)
_add('zcr_alt', 'get_zcr', ('int16', 63, 3000), frame_len=480, frame_step=160)
_add('amprule_bursts', 'amplitude_rule', ('bursts', 55, 32000), mh=0.25)
_add('amprule_bursts_mh', 'amplitude_rule', ('bursts', 55, 32000), mh=0.125)

# next-row f-4: autocorrelation-gated endpointing
for i, spec in enumerate((('vad', 90, 25600), ('vad', 91, 32000), ('bursts', 92, 32000), ('int16', 93, 16000),
                          ('vad', 94, 52920, 44100, 0.7))):
    _add(f'robust_endpoint_{i}', 'robust_endpoint_detection', spec, rate=(spec[3] if len(spec) > 3 else 16000))

# next-row f-1: model.py glue (endpoint_detect without augmentation, feature_extract_mfcc)
_add('model_feat_44k', 'model_feature_extract_mfcc', ('vad', 70, 52920, 44100, 0.7), rate=44100)
_add('model_feat_48k', 'model_feature_extract_mfcc', ('vad', 71, 60000, 48000, 0.6), rate=48000)

# next-row f-4: pitch scores (pitch.pitch_detect_sr as model.py:92 calls it: winlen = cfg.frame, step = cfg.step)
_add('pitch_sr_16k', 'pitch_detect_sr', ('vad', 95, 19200, 16000, 0.7), rate=16000, winlen=0.03, step=0.01)
_add('pitch_sr_44k', 'pitch_detect_sr', ('vad', 96, 39690, 44100, 0.6), rate=44100, winlen=0.03, step=0.01)
_add('pitch_sr_harm', 'pitch_detect_sr', ('harmonic', 97, 12000), rate=16000, winlen=0.0512, step=0.01)

# model.py:90-101: the optional pitch / amplitude streams (cfg.use_pitch, cfg.use_timefeat)
_add('model_side_streams', 'model_side_streams', ('vadf', 98, 24000), rate=16000)

# training path of model.py:52-64: endpoint jitter (augment=True) drawn from Python's global `random`, seeded
_add('model_feat_jitter_44k', 'model_feature_extract_mfcc_aug', ('vad', 75, 52920, 44100, 0.6), rate=44100, seed=1234)
_add('model_feat_jitter_16k', 'model_feature_extract_mfcc_aug', ('vad', 76, 28000, 16000, 0.7), rate=16000, seed=7)

# surface crumbs: endpoint.get_noise (endpoint.py:94-107), sigproc.rolling_window (sigproc.py:59-63)
_add('get_noise_bursts', 'get_noise', ('bursts', 55, 32000), mh=0.25)
_add('get_noise_whole', 'get_noise', ('int16', 56, 16000), mh=0.25)
_add('rolling_window', 'rolling_window', ('ramp', 81, 50), window=7, step=3)
_add('rolling_window_1', 'rolling_window', ('white', 82, 20), window=20, step=1)

# deframesig (API-surface extra)
_add('deframesig', 'deframesig', ('white', 80, 2000), frame_len=400, frame_step=160,
     winfunc='hamming')


def case_by_name(name):
    for c in CASES:
        if c['name'] == name:
            return c
    raise KeyError(name)


def resolve_kwargs(kw):
    """Turn the serialisable kwargs into call kwargs (winfunc names -> callables)."""
    out = dict(kw)
    if 'winfunc' in out:
        out['winfunc'] = WINFUNCS[out['winfunc']]
    return out


def run_case(case, api):
    """Evaluate one case against ``api`` -- any namespace with the reference's function names
    (the reference ``features`` package, the oracle module, or the HIP-backed mirror).

    Returns a dict of named numpy arrays.
    """
    fn, kw = case['fn'], resolve_kwargs(case['kw'])
    reshape = kw.pop('reshape', None)
    spec = case['sig']
    if spec is not None and spec[0] == 'mfcc_of':
        base = case_by_name('mfcc_base_white' if spec[1] == 30 else 'mfcc_base_short')
        x = api.mfcc(make_signal(base['sig']), **resolve_kwargs(base['kw']))
    elif spec is not None:
        x = make_signal(spec)
        if reshape is not None:
            x = x.reshape(reshape)
    else:
        x = None

    if fn in ('preemphasis',):
        return {'out': np.asarray(api.preemphasis(x, **kw))}
    if fn == 'framesig':
        return {'out': np.asarray(api.framesig(x, **kw))}
    if fn == 'to_frames':
        return {'out': np.asarray(api.to_frames(x, **kw))}
    if fn in ('magspec', 'powspec', 'logpowspec'):
        fl, fs = kw.pop('frame_len'), kw.pop('frame_step')
        frames = api.framesig(x, fl, fs, WINFUNCS['hamming'])
        return {'out': np.asarray(getattr(api, fn)(frames, **kw))}
    if fn == 'get_filterbanks':
        return {'out': np.asarray(api.get_filterbanks(**kw))}
    if fn == 'fbank':
        feat, energy = api.fbank(x, **kw)
        return {'feat': np.asarray(feat), 'energy': np.asarray(energy)}
    if fn == 'mfcc':
        return {'out': np.asarray(api.mfcc(x, **kw))}
    if fn == 'lifter':
        return {'out': np.asarray(api.lifter(x, **kw))}
    if fn == 'delta':
        return {'out': np.asarray(api.delta(x, kw['N']))}
    if fn == 'delta2':
        return {'out': np.asarray(api.delta(api.delta(x, kw['N']), kw['N']))}
    if fn == 'basic_endpoint_detection':
        lo, hi, amp, zcr = api.basic_endpoint_detection(x, kw['rate'], return_feature=True)
        return {'endpoints': np.array([lo, hi], dtype=np.int64), 'amp': np.asarray(amp, dtype=np.float64),
                'zcr': np.asarray(zcr, dtype=np.int64)}
    if fn == 'robust_endpoint_detection':
        lo, hi = api.robust_endpoint_detection(x, kw['rate'])
        return {'endpoints': np.array([lo, hi], dtype=np.int64)}
    if fn == 'amplitude_feature':
        return {'out': np.asarray(api.amplitude_feature(x, **kw), dtype=np.float64)}
    if fn == 'get_amplitude':
        frames = api.framesig(x, kw.pop('frame_len'), kw.pop('frame_step'))
        return {'out': np.asarray(api.get_amplitude(frames, **kw), dtype=np.float64)}
    if fn == 'get_zcr':
        frames = api.framesig(x, kw.pop('frame_len'), kw.pop('frame_step'))
        return {'out': np.asarray(api.get_zcr(frames), dtype=np.int64)}
    if fn == 'amplitude_rule':
        amp = api.get_amplitude(api.to_frames(x, 16000, t=0.03, step=0.01))
        seg = api.amplitude_rule(amp, **kw)
        return {'out': np.asarray(seg, dtype=np.int64).reshape(-1, 2)}
    if fn == 'model_feature_extract_mfcc':
        (m0, m1, m2), n = api.model_pipeline(x, kw['rate'])
        return {'m0': np.asarray(m0), 'm1': np.asarray(m1), 'm2': np.asarray(m2),
                'len': np.array([n], dtype=np.int64)}
    if fn == 'model_feature_extract_mfcc_aug':
        (m0, m1, m2), n = api.model_pipeline_aug(x, kw['rate'], kw['seed'])
        return {'m0': np.asarray(m0), 'm1': np.asarray(m1), 'm2': np.asarray(m2),
                'len': np.array([n], dtype=np.int64)}
    if fn == 'get_noise':
        amp = api.get_amplitude(api.to_frames(x, 16000, t=0.03, step=0.01))
        seg = api.amplitude_rule(amp, **kw)
        return {'out': np.array([api.get_noise(amp, seg)], dtype=np.float64)}
    if fn == 'rolling_window':
        return {'out': np.array(api.rolling_window(np.asarray(x, dtype=np.float64), **kw))}
    if fn == 'pitch_detect_sr':
        down = np.asarray(api.downsampling(x, kw['rate'], 10000))
        frames = api.to_frames(down, 10000, kw['winlen'], kw['step'])
        scores = np.array([api.pitch_detect_frame_sr(api.center_clip(fr, False), 10000) for fr in frames],
                          dtype=np.float64)
        pitch, _ = api.pitch_detect_sr(x, kw['rate'], winlen=kw['winlen'], step=kw['step'])
        return {'down': down.astype(np.float64), 'scores': scores, 'pitch': np.asarray(pitch, dtype=np.float64)}
    if fn == 'model_side_streams':
        p0, p1 = api.model_feature_extract_pitch(x, kw['rate'])
        a0, a1 = api.model_feature_extract_timespace(x, kw['rate'])
        return {'pitch0': np.asarray(p0, dtype=np.float64), 'pitch1': np.asarray(p1, dtype=np.float64),
                'amp0': np.asarray(a0, dtype=np.float64), 'amp1': np.asarray(a1, dtype=np.float64)}
    if fn == 'deframesig':
        frames = api.framesig(x, kw['frame_len'], kw['frame_step'], kw['winfunc'])
        return {'out': np.asarray(api.deframesig(frames, len(x), kw['frame_len'], kw['frame_step'],
                                                 kw['winfunc']))}
    raise KeyError(fn)
