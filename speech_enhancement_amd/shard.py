"""Utterance sharding over the GPUs of one node, one process per GPU.

Utterances are independent (the reference's only parallel harness is a thread pool pulling file
indices from a shared counter:
function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:111-121,
311-327), so the data path needs NO collective: every rank denoises its own shard.  The only
cross-rank data are the frame count (sum) and the timed region (max), reduced once at the end
on the host (gloo).
"""
import numpy as np


def lpt_shards(lengths, world):
    """Longest-processing-time greedy partition of utterance indices into `world` shards balanced by
    total samples (processing time is proportional to the number of frames).  Deterministic."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    loads = np.zeros(world, dtype=np.int64)
    shards = [[] for _ in range(world)]
    for idx in order:
        r = int(np.argmin(loads))       # ties -> lowest rank, deterministic
        shards[r].append(int(idx))
        loads[r] += int(lengths[idx])
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def block_shard(n_utt, world, rank):
    """Contiguous block of utterance indices for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_utt, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def reduce_job(frames_local, seconds_local, dist=None, device=None):
    """Whole-job aggregate: (total frames over all ranks, max seconds over ranks).  Two scalars on
    the HOST side of torch.distributed (gloo): the data path itself has no collective, so RCCL and
    xGMI stay out of it entirely.  `device` is accepted for older callers and ignored."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(frames_local), float(seconds_local)
    import torch
    f = torch.tensor([int(frames_local)], dtype=torch.int64)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(f.item()), float(t.item())
