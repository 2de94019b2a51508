#!/usr/bin/env python3
"""Headline benchmark: MFCC frames/sec at 16 kHz, 25 ms / 10 ms, nfft=512, 40 mel, 13 cep
(BASELINE.json `metric`), on BASELINE.json configs[1]:

    one step = one pass of the hot path over a batch of 1024 synthetic 1 s utterances
               -> [1024*99, 39] = MFCC | delta | delta-delta   (dsp_mfcc_delta_batch, C ABI: ONE fused kernel)

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Inputs are resident in HBM before the timed region; successive steps rotate over distinct input
batches whose total size exceeds the 256 MiB Infinity Cache, so reads come from HBM.  Utterances
shard across ranks with no data-path collective (weak scaling: every rank owns its own batches);
configs[2] (12 500 utterances per GPU in one launch, then the RCCL gather of the [12500*99, 39] result to rank 0)
is timed separately and reported under "gather" -- it is not part of `value`.

rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, lowfreq=0,
           highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
DELTA_N = 2
B, N = 1024, 16000              # utterances per step, samples per utterance (1 s @ 16 kHz)
T = 99                          # frames per utterance: 1 + ceil((16000-400)/160)
BYTES_PER_FRAME_MFCC = 4.0 * N / T + 4.0 * 13        # 698.5: fp32 wave read once + 13 fp32 out
BYTES_PER_FRAME_ALL = 4.0 * N / T + 4.0 * 39         # 802.5: SURVEY 8d, MFCC+delta+delta2
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# Second roof (fp32 vector issue): wave instructions the step's kernel executes per frame x 64 lanes, from the
# rocprofv3 SQ_INSTS_VALU pass in profiles/ (DESIGN.md section 7), against 256 CUs x 4 SIMDs x 32 lanes x clock --
# at the nominal 2.4 GHz and at the shader clock the chip actually holds under this kernel (stamped in-kernel,
# same profiles file).
VALU_LANES = 256 * 4 * 32
NOMINAL_GHZ = 2.4


def compute_roof(kernel_ms):
    lane_ops, clock, src = None, None, None
    for name in ('r4_instr.json', 'r3_instr.json', 'r2_instr.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as fh:
                ij = json.load(fh)
            lane_ops = float(ij['valu_lane_ops_per_frame'])
            clock = float(ij.get('shader_clock_ghz_under_load', 0)) or None
            src = f'profiles/{name}'
            break
        except (OSError, ValueError, KeyError):
            continue
    if lane_ops is None:
        return None
    ach = lane_ops * B * T / (kernel_ms * 1e-3)
    out = {'bound': 'valu', 'achieved': ach / 1e12, 'peak': VALU_LANES * NOMINAL_GHZ * 1e9 / 1e12, 'unit': 'T lane-ops/s',
           'frac': ach / (VALU_LANES * NOMINAL_GHZ * 1e9), 'clock_ghz_nominal': NOMINAL_GHZ,
           'lane_ops_per_frame': lane_ops, 'source': src}
    if clock:
        out.update({'clock_ghz_measured': clock, 'peak_at_measured_clock': VALU_LANES * clock * 1e9 / 1e12,
                    'frac_at_measured_clock': ach / (VALU_LANES * clock * 1e9)})
    return out


def synth_batch(seed):
    """Class-A throughput signal of SURVEY 8d: 0.25 * N(0,1), fp32."""
    rng = np.random.default_rng(1_000_003 * 17 + seed)
    return (0.25 * rng.standard_normal((B, N), dtype=np.float32)).astype(np.float32)


def _oracle_worker(args):
    """Run the oracle on `count` synthetic utterances (generated in chunks, generation not timed)."""
    seed, count = args
    from oracle import dsp_oracle
    busy, done = 0.0, 0
    while done < count:
        n = min(B, count - done)
        x = synth_batch(seed)[:n].astype(np.float64)
        seed += 7919
        t0 = time.perf_counter()
        for b in range(n):
            dsp_oracle.mfcc_delta(x[b], delta_n=DELTA_N, winfunc=np.hamming, **CFG)
        busy += time.perf_counter() - t0
        done += n
    return busy, count * T


def cpu_baseline(budget_s=8.0):
    """The NumPy oracle (parity-pinned port of the reference's path) timed on this box's host cores
    on a bounded sample of the same workload: all cores via multiprocessing, plus 1 core."""
    import multiprocessing as mp
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    dt, frames = _oracle_worker((0, 64))           # calibrate on 64 utterances, single core
    per_utt = dt / 64
    single = frames / dt
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))   # the GPU box's CPU share for one GPU
    per_worker = int(max(16, min(32 * B, budget_s / per_utt)))
    ctx = mp.get_context('spawn')
    with ctx.Pool(cores) as pool:
        pool.map(_oracle_worker, [(i, 2) for i in range(cores)])   # warm the workers (imports)
        res = pool.map(_oracle_worker, [(100 + i, per_worker) for i in range(cores)])
    total_frames = sum(r[1] for r in res)
    busy = max(r[0] for r in res)   # workers run concurrently; the slowest one bounds the rate
    return {
        'value': total_frames / busy, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
        'sample': f'{cores} processes x {per_worker} utterances (1 s, 16 kHz) of the same synthetic '
                  f'workload, NumPy oracle mfcc+delta+delta2, {busy:.1f} s of CPU work per process',
        'single_core_value': single,
    }


SHARE_UTT = 12500     # configs[2]: 100 000 utterances over 8 GPUs


def share_launch(dev, plan, reps=20):
    """configs[2]'s per-GPU share: 12 500 x 1 s utterances (800 MB of fp32, generated on the device) through ONE
    launch of the step's kernel -> [12500*99, 39], HIP-event timed.  Returns (result tensor, dict)."""
    import torch
    from features import _native as nat
    g = torch.Generator(device=dev).manual_seed(12)
    w = torch.empty((SHARE_UTT, N), dtype=torch.float32, device=dev)
    for i in range(0, SHARE_UTT, 2500):                      # in slices: randn's temporaries stay small
        w[i:i + 2500] = 0.25 * torch.randn((2500, N), device=dev, generator=g)
    lay = plan.layout(np.empty((SHARE_UTT, N), dtype=np.float32))
    out = torch.empty((SHARE_UTT * T, plan.width(DELTA_N)), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev)
    for _ in range(10):      # ~5 ms of work first: the clock ramps after the host-side pauses before this point
        plan.run_raw(w.data_ptr(), nat.WAVE_F32, lay, out.data_ptr(), DELTA_N, st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(st)
    for _ in range(reps):
        plan.run_raw(w.data_ptr(), nat.WAVE_F32, lay, out.data_ptr(), DELTA_N, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    del w
    frames = SHARE_UTT * T
    return out, {'workload': f'configs[2] share of one GPU: {SHARE_UTT} x 1 s utterances in ONE launch -> [{frames}, 39]',
                 'frames': frames, 'ms_per_launch': ms, 'frames_per_s': frames / ms * 1e3,
                 'algorithmic_GBps': BYTES_PER_FRAME_ALL * frames / ms / 1e6,
                 'frac_of_hbm_peak': BYTES_PER_FRAME_ALL * frames / ms / 1e6 / HBM_PEAK_GBPS}


def model_path(dev, batch=512):
    """configs[4]: the reference's classifier input (model.py:113-135, test path) from device-resident 44.1 kHz int16
    clips -- VAD, endpoint rule, trim, unit variance, NFFT=1536 MFCC, mean removal, delta(3) x 2, z-score,
    [200, B, 39] -- followed by the forward pass of the reference's vanilla RNN classifier (features/classifier.py,
    pinned to the reference class by tests/test_classifier_golden.py; random weights: the reference ships none, so
    accuracy is not measured)."""
    import torch
    from features.model_glue import ModelFeatureBatch
    from features.classifier import RNNHead, fill_parameters
    rate = 44100
    rng = np.random.default_rng(9)
    clips = []
    for _ in range(batch):
        n = int(rng.uniform(1.0, 2.0) * rate)
        x = rng.normal(0, 30, n)
        blen = int(rng.uniform(0.5, 0.9) * n)
        b0 = int(rng.integers(0, n - blen))
        t = np.arange(blen) / rate
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * rng.uniform(100, 300) * t) * np.hanning(blen)
        clips.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    src = torch.from_numpy(np.concatenate(clips)).to(dev)
    mfb = ModelFeatureBatch(rate=rate)
    lay = mfb.pipe.prepare(so, delta_n=0)
    head = RNNHead().to(dev).eval()
    fill_parameters(head, 1)
    for _ in range(2):
        inp, len0, _ = mfb.run(src, layout=lay)
        with torch.no_grad():
            logits = head(inp, len0)
    torch.cuda.synchronize(dev)
    reps, t_eager = 5, 0.0
    for _ in range(reps):      # the per-call form: Python launches, result tensors allocated, lengths downloaded
        t0 = time.perf_counter()
        inp, len0, _ = mfb.run(src, layout=lay)               # ends with a host synchronisation (downloads the lengths)
        t_eager += time.perf_counter() - t0
    # the deployment form: the same launches captured once into a HIP graph over pre-allocated buffers
    # (ModelFeatureBatch.capture); one replay + one synchronisation per batch
    graph = mfb.capture(src, lay)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(dev)
    t_fe, t_clf = 0.0, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        inp, d_len0 = graph.replay()
        len0 = d_len0.cpu().numpy()                            # the lengths the classifier packs by: the one host synchronisation
        t1 = time.perf_counter()
        with torch.no_grad():
            logits = head(inp, len0)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        t_fe += t1 - t0
        t_clf += t2 - t1
    t_fe, t_clf, t_eager = t_fe / reps, t_clf / reps, t_eager / reps
    return {'workload': f'configs[4]: {batch} int16 clips of 1-2 s at 44.1 kHz ({int(so[-1])} samples) resident on the '
                        f'device -> model.py feature pipeline -> [200, {batch}, 39] -> 3-layer bidirectional GRU(200) '
                        f'classifier forward (random weights)',
            'front_end_ms': t_fe * 1e3, 'front_end_form': 'HIP graph replay of the captured launches + download of the 512 lengths',
            'front_end_eager_ms': t_eager * 1e3, 'classifier_ms': t_clf * 1e3, 'utterances_per_s': batch / (t_fe + t_clf),
            'front_end_utterances_per_s': batch / t_fe, 'logits_shape': list(logits.shape),
            'accuracy': 'unpinned: the reference ships neither data nor weights'}


def other_paths(dev):
    """Short timings of the other kernels on the path (not part of `value`): the NFFT=1536 kernel
    (48 kHz, 30 ms / 10 ms, 26 mel, 512 x 1 s) and the configs[3] stages on 1024
    variable-length int16 utterances.  Reported for the record next to the headline number."""
    import torch
    from features import _native as nat
    from features.batch import FeaturePlan
    lib = nat.load()
    st = torch.cuda.current_stream(dev).cuda_stream

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps * 1e3          # us

    out = {}
    # --- NFFT = 1536 ---
    b2, n2 = 512, 48000
    p2 = FeaturePlan(samplerate=48000, winlen=0.03, winstep=0.01, numcep=13, nfilt=26, nfft=1536, preemph=0.97,
                     ceplifter=22, appendEnergy=True, winfunc=np.hamming)
    lay2 = p2.layout(np.empty((b2, n2), dtype=np.float32))
    g = torch.Generator(device=dev).manual_seed(2)
    w2 = 0.25 * torch.randn((b2, n2), device=dev, generator=g)
    o2 = torch.empty((lay2.total_frames, 13), device=dev)
    us = timed(lambda: p2.run_raw(w2.data_ptr(), nat.WAVE_F32, lay2, o2.data_ptr(), 0, st))
    byt = 4.0 * b2 * n2 + 4.0 * lay2.total_frames * 13
    out['nfft1536_mfcc'] = {'workload': '512 x 1 s at 48 kHz, 30 ms / 10 ms, nfft=1536, 26 mel -> 13 cep',
                            'frames': lay2.total_frames, 'us_per_launch': us,
                            'frames_per_s': lay2.total_frames / us * 1e6, 'algorithmic_GBps': byt / us / 1e3}
    # --- configs[3]: VAD features -> endpoint rule -> device-side layout -> trim + unit variance -> ragged
    #     MFCC+delta+delta2, queued as ONE asynchronous sequence (features/pipeline.py); the time is the
    #     end-to-end HIP-event time per call, launches back to back, not a sum of stage timings ---
    from features.pipeline import VadMfccPipeline
    rng = np.random.default_rng(7)
    sigs = []
    for _ in range(B):
        n = int(rng.uniform(1.0, 2.0) * 16000)
        x = rng.normal(0, 30, n)
        blen = int(rng.uniform(0.5, 0.9) * n)
        b0 = int(rng.integers(0, n - blen))
        t = np.arange(blen) / 16000.0
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * rng.uniform(100, 300) * t) * np.hanning(blen)
        sigs.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    so = np.concatenate([[0], np.cumsum([len(s_) for s_ in sigs])]).astype(np.int64)
    d_wave = torch.from_numpy(np.concatenate(sigs)).to(dev)
    pipe = VadMfccPipeline(rate=16000, frame=0.03, step=0.01, unit_variance=True, winfunc=np.hamming,
                           **{k: v for k, v in CFG.items() if k != 'samplerate'})
    lay = pipe.prepare(so, DELTA_N)
    d_feat = torch.empty((lay.frames_bound, lay.D), device=dev)
    us_eager = timed(lambda: pipe.launch(d_wave.data_ptr(), nat.WAVE_I16, lay, d_feat.data_ptr(), st))
    # ... and as a replayed HIP graph of the same five launches (nothing is allocated or synchronised inside, the tables
    # are rebuilt on the device at every replay): the per-call Python / ctypes path of the eager form takes about as long
    # as the device work, so the eager figure is partly the host's
    us, form = us_eager, 'eager launches'
    try:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pipe.launch(d_wave.data_ptr(), nat.WAVE_I16, lay, d_feat.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        us_graph = timed(graph.replay)
        if us_graph < us_eager:
            us, form = us_graph, 'HIP graph replay of the five launches'
    except Exception as e:      # noqa: BLE001 -- the eager figure stands
        form = f'eager launches (graph capture failed: {type(e).__name__})'
    # ... and as a stream of independent batches: three pipelines (each with its own layout, work and result buffers) on
    # three HIP streams, calls dealt round robin -- the launch / drain seams of one batch's five kernels overlap the next
    # batch's, as in the timed region of the headline metric.  Time per call = span of 3 x 100 calls / 300.
    us_overlap = None
    try:
        nstr = 3
        lays = [lay] + [pipe.prepare(so, DELTA_N) for _ in range(nstr - 1)]
        feats_o = [d_feat] + [torch.empty((lay.frames_bound, lay.D), device=dev) for _ in range(nstr - 1)]
        strs = [torch.cuda.Stream(dev) for _ in range(nstr)]

        def one_round():
            for k in range(nstr):
                pipe.launch(d_wave.data_ptr(), nat.WAVE_I16, lays[k], feats_o[k].data_ptr(), strs[k].cuda_stream)
        cur = torch.cuda.current_stream(dev)
        for _ in range(5):
            one_round()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for s_ in strs:
            s_.wait_event(e0)
        for _ in range(100):
            one_round()
        for s_ in strs:
            ev = torch.cuda.Event()
            ev.record(s_)
            cur.wait_event(ev)
        e1.record(cur)
        torch.cuda.synchronize(dev)
        us_overlap = e0.elapsed_time(e1) / (100 * nstr) * 1e3
        del lays, feats_o
    except Exception:      # noqa: BLE001 -- the single-stream figures stand
        us_overlap = None
    frames = int(lay.d_frame_off.download((B + 1,), np.int64)[-1])
    out['configs3_vad_pipeline'] = {
        'workload': f'{B} int16 utterances of 1-2 s at 16 kHz ({int(so[-1])} samples), burst in noise: VAD -> rule '
                    f'-> trim + unit variance -> ragged MFCC+delta+delta2, no host round trip',
        'end_to_end_us': us, 'end_to_end_form': form, 'end_to_end_eager_us': us_eager, 'mfcc_frames': frames,
        'utterances_per_s': B / (us * 1e-6), 'input_GBps': 2.0 * int(so[-1]) / us / 1e3,
        'three_streams_us_per_call': us_overlap,
        'three_streams_note': 'independent batches on three HIP streams (own layout / work / result buffers each): time per '
                              'call under overlap, the analogue of ms_per_step; end_to_end_us is one batch alone on one stream'}
    del d_wave, d_feat, w2, o2
    out['configs4_model'] = model_path(dev)
    plan = FeaturePlan(winfunc=np.hamming, **CFG)
    share_out, out['configs2_share'] = share_launch(dev, plan)
    del share_out
    # --- the opt-in matrix-pipe kernel (csrc/kernels_mfma512.h) on the metric's workload, for the record: one launch
    #     = the whole step; same algorithmic bytes as the step's fused kernel ---
    has_m = lib.dsp_plan_has_mfma512(plan.plan.handle)
    if has_m & 3:
        lay1 = plan.layout(np.empty((B, N), dtype=np.float32))
        g = torch.Generator(device=dev).manual_seed(3)
        w1 = [0.25 * torch.randn((B, N), device=dev, generator=g) for _ in range(4)]
        o1 = torch.empty((lay1.total_frames, 3 * CFG['numcep']), device=dev)
        k = [0]

        def one():
            k[0] += 1
            plan.run_raw(w1[k[0] % 4].data_ptr(), nat.WAVE_F32, lay1, o1.data_ptr(), DELTA_N, st)
        byt = 4.0 * B * N + 4.0 * lay1.total_frames * 3 * CFG['numcep']
        for mode, key, what in ((1, 'mfma512_kernel', 'mfcc512m_kernel (kernels_mfma512.h): 16 frames per product, per-column stage-1 matrices in LDS'),
                                (2, 'mfma512t_kernel', 'mfcc512t_kernel (kernels_mfma512t.h): one frame per product, one register-resident '
                                                       'matrix per DFT stage, window and twiddle on the vector pipe')):
            if not (has_m >> (mode - 1)) & 1:
                continue
            nat.check(lib.dsp_debug_use_mfma512(mode))
            try:
                us = timed(one, reps=100)
            finally:
                nat.check(lib.dsp_debug_use_mfma512(-1))
            out[key] = {
                'kernel': what + ': DFT, mel and DCT as fp16 / bf16 (hi, lo) products on v_mfma_f32_16x16x32, opt-in '
                                 f'(dsp_debug_use_mfma512({mode})); NOT the path `value` times',
                'workload': f'{B} x 1 s, MFCC+delta+delta2 rows in one launch', 'us_per_launch': us,
                'frames_per_s': lay1.total_frames / us * 1e6, 'algorithmic_GBps': byt / us / 1e3,
                'frac_of_hbm_roofline': byt / us / 1e3 / HBM_PEAK_GBPS}
        del w1, o1
    return out


# Rehearsal of the N > 1 path on a box with fewer GPUs (tests only): BENCH_REHEARSE_GLOO=1 lets every rank use
# cuda:0 and runs the control-plane collectives over gloo on CPU tensors.  Never set by the driver; the line
# then says "rehearsal": true and its value is meaningless.
REHEARSE = os.environ.get('BENCH_REHEARSE_GLOO') == '1'


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes ourselves.  The parent
    makes NO GPU call (torch.cuda.device_count() does not initialise the runtime on this image); the
    ranks are fresh children, never an exec of a process that has touched the GPU."""
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    if args.gpus > n_dev and not (REHEARSE and n_dev >= 1):
        sys.stderr.write(f'bench.py: --gpus {args.gpus} requested but only {n_dev} GPU(s) are visible on this '
                         f'node; refusing to report an {args.gpus}-GPU number from fewer devices\n')
        raise SystemExit(2)
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if REHEARSE else r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: a rank that dies at start-up would leave the others blocked in their first barrier for ever
    deadline = time.time() + float(os.environ.get('BENCH_RANKS_TIMEOUT_S', '1500'))
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        failed = [rc for rc in rcs if rc not in (None, 0)]
        if failed or time.time() > deadline:
            for pr in procs:
                if pr.poll() is None:
                    pr.terminate()
            for pr in procs:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
            sys.stderr.write('bench.py: a rank failed or the run timed out; the other ranks were stopped\n')
            raise SystemExit(max([abs(rc) for rc in failed] + [1]))
        time.sleep(0.05)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--buffers', type=int, default=8, help='distinct input batches rotated over (x65.5 MB)')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams the independent steps alternate over')
    ap.add_argument('--min-time', type=float, default=0.6,
                    help='the K-step timed block is repeated until this many seconds have been timed')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gather', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the NFFT=1536 / configs[3] side timings')
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1:
        raise SystemExit('bench.py: --gpus and --steps must be >= 1')

    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            launch_ranks(args)          # does not return
        rank = local_rank = 0
        world = 1
    else:
        rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        world = int(os.environ['WORLD_SIZE'])
        if world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    if REHEARSE:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f'bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, '
                         f'{torch.cuda.device_count()} visible)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if REHEARSE:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    cdev = torch.device('cpu') if REHEARSE else dev       # where the control-plane collectives live

    from features import _native as nat
    from features.batch import FeaturePlan
    nat.check(nat.load().dsp_set_device(local_rank))

    plan = FeaturePlan(winfunc=np.hamming, **CFG)
    layout = plan.layout(np.empty((B, N), dtype=np.float32))
    assert layout.total_frames == B * T
    D = plan.width(DELTA_N)

    host0 = synth_batch(rank * 1000)
    waves = [torch.from_numpy(host0).to(dev)]
    for i in range(1, args.buffers):
        waves.append(torch.from_numpy(synth_batch(rank * 1000 + i)).to(dev))
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    # Steps are independent (own input batch, own output buffer), so consecutive steps alternate over
    # `--streams` HIP streams: the tail of one launch overlaps the head of the next.  Every stream
    # owns two output buffers, so no buffer is ever touched from two streams.
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else [stream]
    ns = len(streams)
    outs = [[torch.empty((B * T, D), dtype=torch.float32, device=dev) for _ in range(2)] for _ in range(ns)]

    def step(i):
        k = i % ns
        plan.run_raw(waves[i % len(waves)].data_ptr(), nat.WAVE_F32, layout, outs[k][(i // ns) & 1].data_ptr(),
                     DELTA_N, streams[k].cuda_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]

    def timed_block(i0):
        """EXACTLY args.steps steps between two barrier + synchronize pairs; returns (host wall seconds,
        device-side span in ms measured with HIP events on every launch stream)."""
        sync_all()
        for e, s_ in zip(ev0, streams):
            e.record(s_)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i0 + i)
        for e, s_ in zip(ev1, streams):
            e.record(s_)
        sync_all()
        dt = time.perf_counter() - t0
        span = max(a.elapsed_time(b) for a, b in zip(ev0, ev1))
        return max_over_ranks(dt), span

    # Untimed pre-warm (clock ramp, TLBs, code objects): a fixed 0.3 s of steps before the W warmup
    # steps -- the first few hundred steps after an idle period run ~13 % slower on this device.
    t_pre = time.perf_counter()
    i_pre = 0
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(50):
            step(i_pre)
            i_pre += 1
        torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    # The K-step block is timed R times (R from the first block, identical on all ranks) so that the
    # timed region covers >= --min-time seconds even when K is small; the line reports the median.
    blocks = [timed_block(0)]
    n_blocks = int(min(1000, max(3, np.ceil(args.min_time / max(blocks[0][0], 1e-6)))))
    for r in range(1, n_blocks):
        blocks.append(timed_block(r * args.steps))
    walls = np.array([b[0] for b in blocks])
    spans = np.array([b[1] for b in blocks])
    dt = float(np.median(walls))

    # --- dominant kernel: the step's ONE kernel (fused MFCC + delta + delta-delta, dsp_mfcc_delta_batch) timed alone,
    #     launches back to back on ONE stream, HIP events on that stream, same input rotation; and, for the record,
    #     the MFCC-only kernel of dsp_features_batch (what rounds 1-2 reported) ---
    lib = nat.load()
    cep = [torch.empty((B * T, plan.C), dtype=torch.float32, device=dev) for _ in range(2)]

    def step_one_stream(i):
        plan.run_raw(waves[i % len(waves)].data_ptr(), nat.WAVE_F32, layout, outs[0][i & 1].data_ptr(), DELTA_N, sp)

    def mfcc_only(i):
        nat.check(lib.dsp_features_batch(plan.plan.handle, waves[i % len(waves)].data_ptr(), nat.WAVE_F32, None,
                                         None, B, B * T, N, nat.OUT_MFCC, cep[i & 1].data_ptr(), plan.C,
                                         None, sp))

    def event_timed(fn, ksteps=200):
        kev0, kev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ktimes = []
        for rep in range(5):
            for i in range(5):
                fn(i)
            torch.cuda.synchronize(dev)
            kev0.record(stream)
            for i in range(ksteps):
                fn(i)
            kev1.record(stream)
            torch.cuda.synchronize(dev)
            ktimes.append(kev0.elapsed_time(kev1) / ksteps)
        return float(np.median(ktimes))

    kernel_ms = event_timed(step_one_stream)
    mfcc_only_ms = event_timed(mfcc_only)

    # --- parity guard on batch 0 (outside the timed region): HIP path vs the CPU oracle ---
    parity = None
    if rank == 0:
        from oracle import dsp_oracle
        step(0)
        torch.cuda.synchronize(dev)
        got = outs[0][0].cpu().numpy().reshape(B, T, D)
        parity = 0.0
        for b in (0, 1, B // 2, B - 1):
            ref = dsp_oracle.mfcc_delta(host0[b].astype(np.float64), delta_n=DELTA_N, winfunc=np.hamming, **CFG)
            parity = max(parity, float(np.max(np.abs(got[b] - ref)) / np.max(np.abs(ref))))

    # --- configs[2]: every rank runs its 12 500-utterance share in ONE launch, then the [12500*99, 39] results are
    #     gathered to rank 0 over RCCL (features/distributed.py: grouped send/recv, other ranks receive nothing).
    #     Timed on its own, never part of `value`. ---
    gather = None
    gather_failed = False
    if world > 1 and not args.no_gather:
        from features.distributed import gather_features
        share_out, share = share_launch(dev, plan, reps=3 if REHEARSE else 10)
        share_ms = max_over_ranks(share['ms_per_launch'])
        nbytes = SHARE_UTT * T * D * 4
        gather = {'share_launch_ms': share_ms, 'share_frames_per_s_all_ranks': world * SHARE_UTT * T / share_ms * 1e3,
                  'bytes_per_rank': nbytes}
        if REHEARSE:     # the rehearsal's collectives run over gloo: the same gather code on host copies of the rows
            share_out = share_out.cpu()
            gather['rehearsal_note'] = 'gloo on CPU tensors (ranks share cuda:0): the code path, not the RCCL timing'
        try:      # a failure of the collective is reported in the line AND in the exit code, after the line is out
            if REHEARSE and os.environ.get('BENCH_FAIL_GATHER') == '1':
                raise RuntimeError('BENCH_FAIL_GATHER=1 (test of the failure path)')
            rows, _ = gather_features(share_out, dst=0)          # warm-up (communicator set-up, allocations)
            del rows
            sync_all()
            g0 = time.perf_counter()
            rows, counts = gather_features(share_out, dst=0)
            sync_all()
            gms = max_over_ranks((time.perf_counter() - g0) * 1e3)
            gather.update({'collective': ('gloo' if REHEARSE else 'rccl') + ' gather to rank 0 (grouped send/recv, features/distributed.py::gather_features(dst=0))',
                           'ms': gms, 'root_ingest_GBps': (world - 1) * nbytes / gms / 1e6,
                           'rows_at_root': None if rows is None else int(rows.shape[0])})
            if rank == 0 and (rows is None or int(rows.shape[0]) != world * SHARE_UTT * T):
                raise RuntimeError(f'gather returned {None if rows is None else int(rows.shape[0])} rows at the root, expected {world * SHARE_UTT * T}')
            del rows
        except Exception as exc:                                 # noqa: BLE001 -- reported in the line
            gather['error'] = repr(exc)[:300]
            gather_failed = True
        del share_out
        if world > 1:    # every rank learns of a failure anywhere (the exit code must not depend on the rank)
            flag = torch.tensor([1.0 if gather_failed else 0.0], dtype=torch.float64, device=cdev)
            try:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                gather_failed = bool(flag.item() > 0)
            except Exception:                                    # noqa: BLE001
                gather_failed = True
    per_rank_ms = None
    if world > 1:
        mine = torch.tensor([float(np.median([b[1] for b in blocks])) / args.steps], dtype=torch.float64, device=cdev)
        parts = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank_ms = [float(x.item()) for x in parts]

    # HBM traffic per launch measured with rocprofv3 PMC passes (cannot be collected inside this process)
    traffic, step_traffic, traffic_source = None, None, None
    for name in ('r4_traffic.json', 'r3_traffic.json'):     # FETCH / WRITE passes of THIS kernel (the fused one); older files describe the round-2 kernel
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as fh:
                tj = json.load(fh)
            if tj.get('frames_per_launch') == B * T:
                traffic = tj['traffic_bytes_per_launch']
                step_traffic = tj.get('step_traffic_bytes')
                traffic_source = (f'profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                  f'workload (FETCH doubled per the gfx950 guide); a constant read from that file, '
                                  f'not measured in this run')
                break
        except (OSError, ValueError, KeyError):
            continue

    frames_total = float(world) * B * T * args.steps
    value = frames_total / dt
    achieved = BYTES_PER_FRAME_ALL * B * T / (kernel_ms * 1e-3) / 1e9
    res = {
        'metric': 'MFCC frames/sec at 16 kHz, 25 ms/10 ms, nfft=512, 40 mel, 13 cep',
        'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'timing': {'blocks_of_K_steps': int(n_blocks), 'timed_seconds': float(walls.sum()),
                   'ms_per_step_median': dt / args.steps * 1e3,
                   'ms_per_step_min': float(walls.min()) / args.steps * 1e3,
                   'ms_per_step_max': float(walls.max()) / args.steps * 1e3,
                   'note': 'every block = exactly K steps between barrier+synchronize pairs, max over ranks; '
                           'value uses the median block'},
        'config': {'workload': 'configs[1]: batch of 1024 synthetic 1 s 16 kHz utterances -> MFCC+delta+delta2 '
                               '[1024*99, 39] per step per GPU', 'utterances_per_step_per_gpu': B,
                   'frames_per_utterance': T, 'delta_n': DELTA_N, 'input_buffers_rotated': len(waves),
                   'hip_streams': ns,
                   'sharding': f'{world} ranks x independent batches, no data-path collective'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_source,
                     'step_traffic': step_traffic,
                     'frac_of_measured_copy_ceiling': achieved / 6290.0,   # MI355X_MICROARCH.md: 6.29 TB/s copy
                     'algorithmic_bytes_per_launch': BYTES_PER_FRAME_ALL * B * T,
                     'kernel': 'fused MFCC + delta + delta-delta kernel (dsp_mfcc_delta_batch: the whole step is this one '
                               'launch; reads 4 B x 16000 per utterance, writes 156 B per frame)',
                     'kernel_ms': kernel_ms, 'kernel_ms_note': 'one launch stream, back to back, HIP events',
                     'kernel_ms_overlapped': float(np.median(spans)) / args.steps,
                     'kernel_ms_overlapped_note': 'device-side span of a timed block (HIP events on every launch '
                                                  'stream, max over streams) / K: the same kernel under the stream '
                                                  'overlap of the timed region; <= ms_per_step',
                     'bytes_per_frame': BYTES_PER_FRAME_ALL, 'frames_per_launch': B * T,
                     'mfcc_only_kernel': {'kernel': 'MFCC-only kernel (dsp_features_batch, DSP_OUT_MFCC -> [sum T, 13]), the '
                                                    'kernel rounds 1-2 reported', 'kernel_ms': mfcc_only_ms,
                                          'bytes_per_frame': BYTES_PER_FRAME_MFCC,
                                          'achieved': BYTES_PER_FRAME_MFCC * B * T / (mfcc_only_ms * 1e-3) / 1e9,
                                          'frac': BYTES_PER_FRAME_MFCC * B * T / (mfcc_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                     'whole_step_GBps': BYTES_PER_FRAME_ALL * value / world / 1e9,
                     'compute': compute_roof(kernel_ms)},
        'parity_normwise_vs_oracle': parity,
    }
    if REHEARSE:
        res['rehearsal'] = True
    if per_rank_ms is not None:
        res['per_rank_ms_per_step'] = per_rank_ms
    if gather is not None:
        res['gather'] = gather
        res['gather_failed'] = bool(gather_failed)
    if rank == 0 and world == 1 and not args.no_extras:
        res['other_paths'] = other_paths(dev)
    if rank == 0 and not args.no_cpu_baseline:      # rank 0's host cores, at every N (the other ranks wait at the final barrier)
        res['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        try:
            dist.barrier()
        except Exception:                                        # noqa: BLE001
            pass
        dist.destroy_process_group()
    if gather_failed:
        raise SystemExit(3)       # the line is out; the failed collective still fails the run


if __name__ == '__main__':
    main()
