"""CPU-side checks (no GPU): host logic of the drop-in package and the C-ABI surface.
The library must LOAD and export every symbol declared in include/dsp_frontend.h; no compute
entry point is called here."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import dsp_oracle


@pytest.fixture(scope='module')
def nat():
    from features import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _native


def test_header_symbols_exported(nat):
    hdr = open(os.path.join(ROOT, 'include', 'dsp_frontend.h')).read()
    declared = set(re.findall(r'\b(dsp_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'dsp_plan_desc'}
    lib = nat.load()
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert lib.dsp_abi_version() == 1


def test_library_exports_nothing_the_header_does_not_declare(nat):
    """The other direction: every `dsp_*` function the shipped library exports is declared in include/dsp_frontend.h
    (the per-phase stamp readers exist only in diagnostic builds compiled with -DF512_STAMPS / -DM512_STAMPS /
    -DM512T_STAMPS; the product build must not carry them)."""
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-D', '--defined-only', nat.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('dsp_') and ' T ' in ln}
    hdr = open(os.path.join(ROOT, 'include', 'dsp_frontend.h')).read()
    declared = set(re.findall(r'\b(dsp_[a-z0-9_]+)\s*\(', hdr))
    assert exported and exported <= declared, sorted(exported - declared)


def test_plan_desc_layout_matches_header(nat):
    import ctypes as C
    # 6 int32 + 1 float (28 bytes, padded to 32) + 5 pointers
    assert C.sizeof(nat.PlanDesc) == 32 + 5 * 8
    assert nat.PlanDesc.h_window.offset == 32


def test_frame_count_host(nat):
    for n, L, S in ((0, 400, 160), (300, 400, 160), (400, 400, 160), (401, 400, 160), (16000, 400, 160),
                    (16000, 480, 160), (1840, 400, 160), (12345, 551, 221)):
        assert nat.frame_count(n, L, S) == dsp_oracle.frame_geometry(n, L, S)[2]
    so = np.array([0, 16000, 16300, 20000, 20000 + 12345], dtype=np.int64)
    fo = nat.frame_offsets(so, 400, 160)
    exp = np.concatenate(([0], np.cumsum([dsp_oracle.frame_geometry(int(n), 400, 160)[2] for n in np.diff(so)])))
    assert np.array_equal(fo, exp)


def test_round_half_up_and_sizes():
    from features import _plan
    assert _plan.round_half_up(0.01 * 22050) == 221 and round(0.01 * 22050) == 220
    assert _plan.round_half_up(0.025 * 22050) == 551
    assert _plan.frame_sizes(0.025 * 16000, 0.01 * 16000) == (400, 160)
    for x in (0.5, 1.5, 2.5, 399.5, 400.49999, 1323.0):
        assert _plan.round_half_up(x) == dsp_oracle.round_half_up(x)


@pytest.mark.parametrize('name', ['fb_26_512_16k', 'fb_40_512_16k', 'fb_26_1536_44k', 'fb_26_1536_48k',
                                  'fb_default', 'fb_band', 'fb_dense'])
def test_filterbank_table_matches_reference(name, golden):
    from features import _plan
    from golden_cases import case_by_name
    kw = case_by_name(name)['kw']
    fb = _plan.filterbank_matrix(**kw)
    assert np.array_equal(fb, golden[f'{name}/out'])
    start, count, w = _plan.mel_csr(fb)
    dense = np.zeros_like(fb)
    off = 0
    for j in range(fb.shape[0]):
        dense[j, start[j]:start[j] + count[j]] = w[off:off + count[j]]
        off += count[j]
    assert np.max(np.abs(dense - fb)) <= 1e-7


def test_baseline_filterbank_facts():
    """454 non-zeros, documented bin edges, cols 0 and 256 carry no weight."""
    from features import _plan
    fb = _plan.filterbank_matrix(40, 512, 16000, 0, None)
    assert np.count_nonzero(fb) == 454
    assert not fb[:, 0].any() and not fb[:, 256].any()
    edges = _plan.mel_edges(40, 512, 16000, 0, None).astype(int)
    assert list(edges[:6]) == [0, 1, 2, 4, 6, 8] and list(edges[-3:]) == [224, 239, 256]


def test_dct_lifter_table():
    from features import _plan
    from scipy.fftpack import dct
    x = np.random.default_rng(3).standard_normal((5, 40))
    ref = dct(x, type=2, axis=1, norm='ortho')[:, :13] * dsp_oracle.lifter_vector(13, 22)
    got = x @ _plan.dct_lifter_matrix(40, 13, 22).T
    assert np.max(np.abs(got - ref)) < 1e-12
    assert np.allclose(_plan.lifter_vector(13, 22)[:3], [1, 2.5655, 4.0991], atol=1e-4)
    assert np.array_equal(_plan.lifter_vector(5, 0), np.ones(5))


def test_host_rules_match_oracle():
    """amplitude_rule / zcr_rule are host control logic in the drop-in; pin them to the oracle."""
    from features import endpoint
    from golden_cases import make_signal
    for spec in (('bursts', 55, 32000), ('vad', 50, 25600), ('int16', 56, 16000)):
        x = make_signal(spec)
        frames = dsp_oracle.to_frames(x, 16000, 0.03, 0.01)
        amp, zcr = dsp_oracle.get_amplitude(frames), dsp_oracle.get_zcr(frames)
        for mh in (0.25, 0.125):
            assert endpoint.amplitude_rule(amp, mh) == dsp_oracle.amplitude_rule(amp, mh)
        seg = dsp_oracle.amplitude_rule(amp)
        assert endpoint.zcr_rule(zcr, seg[0][0], seg[-1][1]) == dsp_oracle.zcr_rule(zcr, seg[0][0], seg[-1][1])


def test_no_gpu_means_loud_failure(nat):
    """Without a device the product path must raise, never fall back to a CPU route."""
    import ctypes as C
    n = C.c_int(0)
    rc = nat.load().dsp_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is present')
    import features
    with pytest.raises(nat.DspError):
        features.mfcc(np.zeros(16000))
    with pytest.raises(nat.DspError):
        features.basic_endpoint_detection(np.zeros(16000, dtype=np.int16), 16000)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'dsp-speech-recognition_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no oracle', ''), f'{f} mentions the oracle'


def nat_has_device():
    import ctypes as C
    from features import _native as nat
    n = C.c_int(0)
    return nat.load().dsp_device_count(C.byref(n)) == nat.OK and n.value >= 1


def test_pitch_host_helpers_match_oracle():
    """The host-side pieces of the pitch path (decimation indices, centre clipping, band-pass taps,
    in-place smoothing, octave repair) against the oracle restatements pinned to the reference."""
    from features import pitch as gp
    from features.preprocess import downsampling
    from oracle import dsp_oracle
    rng = np.random.default_rng(9)
    x = rng.standard_normal(7001)
    for rate in (10000, 11025, 16000, 22050, 44100, 48000):
        assert np.array_equal(downsampling(x, rate, 10000), dsp_oracle.downsampling(x, rate, 10000)), rate
    literal, ticks = [], -1                      # the reference's loop, spelled out once
    for i in range(len(x)):
        if i * 10000 / 44100 > ticks + 1e-8:
            ticks += 1
            literal.append(i)
    assert np.array_equal(downsampling(x, 44100, 10000), x[literal])
    fr = np.round(300 * rng.standard_normal(300))
    for binary in (True, False):
        assert np.array_equal(gp.center_clip(fr, binary), dsp_oracle.center_clip(fr, binary))
    assert np.array_equal(gp.center_clip(-np.abs(fr) - 1, False), np.zeros(300))    # no non-negative sample
    assert np.allclose(gp.bandpass_taps(300, 10000, 50, 900, 'hamming'),
                       dsp_oracle.bandpass_taps(300, 10000, 50, 900, 'hamming'), rtol=0, atol=0)
    assert np.allclose(gp.window(fr, 10000, 50, 900, 'hamming'), dsp_oracle.window(fr, 10000, 50, 900, 'hamming'),
                       rtol=0, atol=1e-12)
    # (smooth / max_pitch / robust_max_pitch run on the device since round 4: tests/test_gpu_fuzz.py)
    with pytest.raises(Exception, match='no CPU fallback'):
        if nat_has_device():
            raise RuntimeError('no CPU fallback')          # a GPU box: nothing to check here
        gp.smooth(rng.random((5, 180)), 2)                  # ... and without a device they fail loudly, no host route


def test_scratch_slots_are_per_thread_and_never_shared():
    """The slot-leasing logic behind the drop-in calls (features/_native.py), without a GPU: slots are
    keyed by (thread, device, name), grow by replacement inside their own thread only, and two threads
    asking for the same name never see the same buffer."""
    import threading
    from features import _native as nat

    class Fake:
        live = 0

        def __init__(self, nbytes):
            self.nbytes, self.freed = nbytes, False
            Fake.live += 1

        def free(self):
            assert not self.freed
            self.freed = True
            Fake.live -= 1

    sc = nat.Scratch(alloc=Fake)
    a = sc.get('wave', 1000, device=0)
    assert sc.get('wave', 500, device=0) is a and not a.freed           # reuse, no shrink
    b = sc.get('wave', 100000, device=0)
    assert b is not a and a.freed and b.nbytes >= 100000                  # growth replaces in the owner thread
    assert sc.get('wave', 10, device=1) is not b                          # per device
    seen = {}

    def other():
        seen['buf'] = sc.get('wave', 10, device=0)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen['buf'] is not b and not b.freed                           # another thread: its own slot
    assert Fake.live == 3


def test_prologue_of_the_dense_nfft512_kernels_waits_for_exactly_its_table_loads():
    """the prologue of mfcc512_kernel issues its table loads as inline asm (which the compiler's wait-count
    insertion does not see) and waits for them with a hand-written s_waitcnt vmcnt(NSTAGE).  tools/asm_check_prologue.py
    checks on the PRODUCT build's listing that exactly NSTAGE vector-memory instructions sit between them, on every path,
    and that nothing touches the loads' destination registers before the wait -- for every dense instantiation."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc: the listing cannot be built here')
    if os.environ.get('DSP_HOST_ASAN'):
        pytest.skip('the sanitizer run checks host code; the listing is checked by the plain CPU suite')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'asm_check_prologue.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'instantiations checked, 0 failed' in r.stdout
