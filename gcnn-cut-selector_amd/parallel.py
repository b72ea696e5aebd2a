"""Data parallelism over the GPUs of one node: one process per GPU, replicated parameters, ONE RCCL all-reduce per step.

The reference has no distributed code.  A stacked mini-batch is a disjoint union of per-sample graphs
(/root/reference/utils.py:401-407), no edge crosses samples, so samples shard across ranks with no data-path exchange; the
only collective is the sum of one flat fp32 buffer [93,121 gradients | local cut count | pad] (372 KB, latency-bound over
xGMI), after which every rank divides by the GLOBAL cut count -- the reference's loss is the mean over ALL cuts of the
batch (model_trainer.py:271), which a mean of per-rank means would not reproduce when ranks hold different cut counts."""

from __future__ import annotations

import numpy as np


def shard_samples(sizes, world_size: int):
    """Assign sample indices to ranks, balancing the number of EDGES (the cost driver), not the number of samples:
    longest-processing-time greedy, ties broken by index so every rank computes the same assignment.
    `sizes[i]` = edge count of sample i.  Returns a list of `world_size` sorted index lists."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(x) for x in out]


def pack(grads: np.ndarray, n_cuts: int) -> np.ndarray:
    """[local gradient of the SUM of squared errors | local cut count | pad] -- the buffer that is all-reduced."""
    buf = np.zeros(grads.size + 4, np.float32)
    buf[:grads.size] = grads
    buf[grads.size] = n_cuts
    return buf


def unpack_mean(buf: np.ndarray) -> np.ndarray:
    """After the SUM all-reduce: gradient of the mean over all cuts of the global batch."""
    n = buf.size - 4
    return buf[:n] / buf[n]
