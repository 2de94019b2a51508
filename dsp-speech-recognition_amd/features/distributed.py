"""Multi-GPU layer: utterances are independent (no cross-utterance state anywhere in the path,
delta edge padding is per utterance), so they shard across ranks -- one process per GPU -- with NO
collective on the data path.  The only exchange is the optional gather of the per-utterance feature
tensors at the end (BASELINE.json configs[2]), done with torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for the tests).

torch.distributed is plumbing here: every function takes ready tensors / arrays and never computes
features itself.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(lengths, world):
    """Contiguous utterance ranges [(lo, hi)] * world, balanced by total SAMPLES (variable-length
    batches), every rank non-empty when there are at least `world` utterances."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if world <= 0:
        raise ValueError('world must be >= 1')
    csum = np.concatenate(([0], np.cumsum(lengths)))
    total = csum[-1]
    bounds, lo = [], 0
    for r in range(world):
        if r == world - 1:
            hi = n
        else:
            target = total * (r + 1) / world
            hi = int(np.searchsorted(csum, target, side='left'))
            hi = max(hi, lo + 1) if n - lo > world - r - 1 else lo
            hi = min(hi, n - (world - r - 1)) if n >= world else min(hi, n)
            hi = max(hi, lo)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def local_slice(sample_offsets, world, rank):
    """(lo, hi, local_sample_offsets) of rank's share of a concatenated batch."""
    so = np.asarray(sample_offsets, dtype=np.int64)
    lo, hi = shard_bounds(np.diff(so), world)[rank]
    return lo, hi, so[lo:hi + 1] - so[lo]


def gather_features(local, group=None, dst=None):
    """Gather variable-length [n_r, D] feature tensors from every rank.

    One small all_gather of the row counts, then
      * dst is None: ONE all_gather_into_tensor of the payload padded to the largest count (per-link bound on
        xGMI: fewer, larger collectives) -- every rank returns the rows of all ranks in rank order;
      * dst = r: a real gather (BASELINE.json configs[2], "RCCL gather of MFCC tensors"): every other rank SENDS
        exactly its rows and rank r receives them straight into their place in the result (one grouped
        send/recv, no padding, no staging copy); the other ranks receive nothing and return (None, counts).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local, D = int(local.shape[0]), int(local.shape[1])
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    mine = torch.tensor([n_local], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = [int(c) for c in counts.cpu()]
    if dst is not None:
        local = local.contiguous()
        if rank != dst:
            if n_local > 0:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, _global_rank(dst, group), group)]):
                    w.wait()
            return None, counts_h
        rows = torch.empty((sum(counts_h), D), dtype=local.dtype, device=local.device)
        ops, at = [], 0
        for r in range(world):
            part = rows[at:at + counts_h[r]]
            if r == rank:
                part.copy_(local)
            elif counts_h[r] > 0:
                ops.append(dist.P2POp(dist.irecv, part, _global_rank(r, group), group))
            at += counts_h[r]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return rows, counts_h
    n_max = max(counts_h)
    padded = local
    if n_local < n_max:
        padded = torch.zeros((n_max, D), dtype=local.dtype, device=local.device)
        padded[:n_local] = local
    buf = torch.empty((world * n_max, D), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded.contiguous(), group=group)
    buf = buf.view(world, n_max, D)
    rows = torch.cat([buf[r, :counts_h[r]] for r in range(world)], dim=0)
    return rows, counts_h


def _global_rank(group_rank, group):
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def extract_sharded(compute, waves, sample_offsets, group=None, gather=True):
    """Shard a concatenated batch over the ranks of `group`, run `compute(local_waves,
    local_sample_offsets) -> (features [n, D] tensor, frame_offsets)` on each, optionally gather.

    `compute` is the rank-local feature extractor (FeaturePlan.mfcc_batch on a GPU rank).
    Returns (features, frame_counts_per_rank) -- features are the global rows when gather=True.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    so = np.asarray(sample_offsets, dtype=np.int64)
    lo, hi, local_so = local_slice(so, world, rank)
    local_waves = waves[so[lo]:so[hi]]
    feats, _ = compute(local_waves, local_so)
    if not gather:
        return feats, None
    return gather_features(feats, group=group)
