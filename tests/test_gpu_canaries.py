"""Out-of-bounds canaries around every device buffer the hot path writes (the pool has no GPU
AddressSanitizer).  The kernels use element-aligned vector loads and bounds-checked descriptors exactly at buffer
edges, so every output -- feature rows, cepstra, per-frame amplitude sums and zero-crossing counts -- is placed
between sentinel words inside a larger allocation, for ragged, misaligned (+1..+3 elements), 1-sample and 7-sample
utterances, dense odd lengths, both fused kernels and both matrix-pipe kernels; the sentinels must come back untouched
and the payload must be fully written (and equal to an unpadded run)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAD = 4096                      # sentinel bytes either side (a multiple of every alignment the kernels assume)
SENT32 = np.int32(0x7fc0beef - (1 << 32) if 0x7fc0beef >= (1 << 31) else 0x7fc0beef)   # a quiet-NaN pattern: never a result


def _guarded(nbytes, dev, shift=0):
    """(whole uint8 tensor, payload pointer, check()) -- payload starts PAD + shift bytes in."""
    import torch
    total = PAD + shift + nbytes + PAD
    total = (total + 3) // 4 * 4
    buf = torch.empty(total // 4, dtype=torch.int32, device=dev)
    buf.fill_(int(SENT32))
    raw = buf.view(torch.uint8)

    def check(what):
        torch.cuda.synchronize(dev)
        host = raw.cpu().numpy()
        ref = np.full(total // 4, SENT32, dtype=np.int32).view(np.uint8)
        lo, hi = PAD + shift, PAD + shift + nbytes
        assert np.array_equal(host[:lo], ref[:lo]), f'{what}: bytes BEFORE the buffer were written'
        assert np.array_equal(host[hi:], ref[hi:]), f'{what}: bytes AFTER the buffer were written'
        return host[lo:hi]
    return buf, buf.data_ptr() + PAD + shift, check


def _ragged_lengths(rng, n, rate):
    lens = [1, 7, 399, 400, 401, 560, 561] + [int(x) for x in rng.integers(1, 3 * rate, n)]
    rng.shuffle(lens)
    return lens


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
@pytest.mark.parametrize('mis', [0, 1, 2, 3])
@pytest.mark.parametrize('nfft,rate,winlen,nfilt', [(512, 16000, 0.025, 40), (1536, 48000, 0.03, 26)])
def test_ragged_misaligned_batches_stay_inside_their_buffers(nfft, rate, winlen, nfilt, mis, dtype):
    import torch
    from features import _native as nat
    from features.batch import FeaturePlan
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(1000 * mis + nfft)
    lens = _ragged_lengths(rng, 40, rate // 8)
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = (3000 * rng.standard_normal(int(so[-1])))
    flat = np.clip(np.round(flat), -32768, 32767).astype(np.int16) if dtype == np.int16 else (flat / 3000).astype(np.float32)
    plan = FeaturePlan(samplerate=rate, winlen=winlen, winstep=0.01, numcep=13, nfilt=nfilt, nfft=nfft, preemph=0.97,
                       ceplifter=22, appendEnergy=True, winfunc=np.hamming)
    lay = plan.layout(flat, so)
    # the WAVE buffer sits `mis` elements into an allocation whose neighbours are NaN / extreme: a read past an
    # utterance's end must not leak into any result either (checked against the unpadded run below)
    elem = flat.dtype.itemsize
    wbuf = torch.full((flat.size + 64,), float('nan') if dtype == np.float32 else 32767, dtype=torch.float32 if dtype == np.float32 else torch.int16, device=dev)
    wbuf[16 + mis:16 + mis + flat.size] = torch.from_numpy(flat).to(dev)
    d_wave = wbuf.data_ptr() + (16 + mis) * elem
    wd = nat.WAVE_I16 if dtype == np.int16 else nat.WAVE_F32
    ref_rows, _ = plan.mfcc_batch(torch.from_numpy(flat).to(dev), so, delta_n=2)
    ref_cep, _ = plan.mfcc_batch(torch.from_numpy(flat).to(dev), so, delta_n=0)
    for delta_n, ref in ((2, ref_rows), (0, ref_cep)):
        nbytes = lay.total_frames * plan.width(delta_n) * 4
        buf, ptr, check = _guarded(nbytes, dev)
        plan.run_raw(d_wave, wd, lay, ptr, delta_n, torch.cuda.current_stream(dev))
        got = check(f'nfft {nfft} delta_n {delta_n} mis {mis}').view(np.float32).reshape(lay.total_frames, -1)
        assert np.isfinite(got).all()
        want = ref.cpu().numpy()
        # a misaligned start takes the ragged-view path: same kernels, results equal to the last bits of fp contraction
        assert np.max(np.abs(got - want)) <= 1e-5 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('B,N,dtype', [(1024, 16000, np.float32), (777, 12345, np.float32), (600, 16002, np.int16), (3000, 401, np.float32), (5000, 7, np.float32)])
def test_dense_batches_stay_inside_their_buffers(mode, B, N, dtype):
    """Dense [B, N] batches through the fused vector-pipe kernel (mode 0) and both matrix-pipe kernels (modes 1, 2;
    batches they do not serve fall through): odd lengths end inside a 16-byte vector, the last utterance ends with
    the allocation."""
    import torch
    from features import _native as nat
    from features.batch import FeaturePlan
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(B + N)
    x = 0.25 * rng.standard_normal((B, N))
    waves = np.clip(np.round(3000 * x), -32768, 32767).astype(np.int16) if dtype == np.int16 else x.astype(np.float32)
    plan = FeaturePlan(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, preemph=0.97,
                       ceplifter=22, appendEnergy=True, winfunc=np.hamming)
    lay = plan.layout(waves)
    # the waveform ends exactly with its allocation's payload; the guard behind it holds NaN patterns
    wbytes = waves.size * waves.dtype.itemsize
    wguard, wptr, wcheck = _guarded(wbytes, dev)
    wguard.view(torch.uint8)[PAD:PAD + wbytes] = torch.from_numpy(waves.reshape(-1).view(np.uint8)).to(dev)
    wd = nat.WAVE_I16 if dtype == np.int16 else nat.WAVE_F32
    lib = nat.load()
    nat.check(lib.dsp_debug_use_mfma512(mode))
    try:
        for delta_n in (2, 0):
            nbytes = lay.total_frames * plan.width(delta_n) * 4
            buf, ptr, check = _guarded(nbytes, dev)
            plan.run_raw(wptr, wd, lay, ptr, delta_n, torch.cuda.current_stream(dev))
            got = check(f'mode {mode} B {B} N {N} delta_n {delta_n}').view(np.float32)
            assert np.isfinite(got).all()      # every element written (the fill is a NaN pattern), none from beyond the input
    finally:
        nat.check(lib.dsp_debug_use_mfma512(-1))
    wcheck('waveform guard')


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
@pytest.mark.parametrize('mis', [0, 1, 3])
def test_vad_outputs_stay_inside_their_buffers(mis, dtype):
    """dsp_vad_features_batch (amp sums fp64, zero-crossing counts int32) on ragged, misaligned batches with 1-sample
    and 7-sample utterances, and on a dense batch: sentinels around d_amp and d_zcr."""
    import torch
    from features import _native as nat
    from features.batch import EndpointPlan
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(77 + mis)
    ep = EndpointPlan(16000, 0.03, 0.01)
    lib = nat.load()
    for dense in (False, True):
        if dense:
            waves = rng.standard_normal((300, 4001))
            so = None
        else:
            lens = _ragged_lengths(rng, 60, 4000)
            so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            waves = rng.standard_normal(int(so[-1]))
        waves = np.clip(np.round(3000 * waves), -32768, 32767).astype(np.int16) if dtype == np.int16 else waves.astype(np.float32)
        lay = ep.layout(waves, so)
        elem = waves.dtype.itemsize
        wbuf = torch.zeros(waves.size + 64, dtype=torch.float32 if dtype == np.float32 else torch.int16, device=dev)
        wbuf[8 + mis:8 + mis + waves.size] = torch.from_numpy(waves.reshape(-1)).to(dev)
        d_wave = wbuf.data_ptr() + (8 + mis) * elem
        abuf, aptr, acheck = _guarded(lay.total_frames * 8, dev)
        zbuf, zptr, zcheck = _guarded(lay.total_frames * 4, dev)
        nat.check(lib.dsp_vad_features_batch(d_wave, nat.WAVE_I16 if dtype == np.int16 else nat.WAVE_F32, lay.p_sample, lay.p_frame,
                                             lay.n_utt, lay.total_frames, lay.uniform_samples, ep.L, ep.S, 0, aptr, zptr, None))
        amp = acheck(f'amp dense={dense}').view(np.float64)
        zcr = zcheck(f'zcr dense={dense}').view(np.int32)
        assert np.isfinite(amp).all() and (amp >= 0).all() and (zcr >= 0).all() and (zcr < ep.L).all()
