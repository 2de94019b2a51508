"""N>1 path on CPU: two gloo ranks shard a ragged batch, each computes its share (the oracle
stands in for the GPU kernels -- this test covers the sharding / gather plumbing only), and the
gathered result must equal the single-process result row for row."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, preemph=0.97,
           ceplifter=22, appendEnergy=True)


def _batch():
    rng = np.random.default_rng(5)
    lens = [4000, 1600, 8000, 300, 2400, 5000, 1234]
    so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    return (0.25 * rng.standard_normal(so[-1])).astype(np.float32), so


def _compute(waves, so):
    from oracle import dsp_oracle
    rows, fo = [], [0]
    for b in range(len(so) - 1):
        m = dsp_oracle.mfcc_delta(np.asarray(waves[so[b]:so[b + 1]], dtype=np.float64), delta_n=2,
                                  winfunc=np.hamming, **CFG)
        rows.append(m)
        fo.append(fo[-1] + len(m))
    return torch.from_numpy(np.concatenate(rows).astype(np.float32)), np.asarray(fo)


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, 'dsp-speech-recognition_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from features import distributed as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        waves, so = _batch()
        rows, counts = D.extract_sharded(_compute, waves, so)
        lo, hi, local_so = D.local_slice(so, world, rank)
        local_feats, _ = _compute(waves[so[lo]:so[hi]], local_so)
        only0, _ = D.gather_features(local_feats, dst=0)
        q.put((rank, rows.numpy(), counts, None if only0 is None else only0.numpy(), None))
    except Exception as e:  # surface the failure instead of letting the parent time out
        q.put((rank, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_rank_shard_and_gather_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[4] is None, r[4]
    waves, so = _batch()
    full, _ = _compute(waves, so)
    for rank, rows, counts, only0, _ in res:
        assert sum(counts) == full.shape[0]
        assert np.array_equal(rows, full.numpy()), f'rank {rank} gathered rows differ'
        if rank == 0:
            assert np.array_equal(only0, full.numpy())
        else:
            assert only0 is None


@pytest.mark.parametrize('lens,world', [([5, 5, 5, 5], 2), ([100, 1, 1, 1, 1, 1], 3), ([7], 1), ([3, 9], 2),
                                        (list(range(1, 41)), 8), ([4, 4, 4], 8)])
def test_shard_bounds_cover_and_balance(lens, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'dsp-speech-recognition_amd'))
    from features.distributed import shard_bounds
    b = shard_bounds(lens, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == len(lens)
    for (l0, h0), (l1, h1) in zip(b, b[1:]):
        assert h0 == l1 and l0 <= h0
    if len(lens) >= world:
        assert all(hi > lo for lo, hi in b)
