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


def chan_merge(stats):
    """Merge streaming (count, mean, var) triples in list order with the update of PreNormLayer.update_params
    (/root/reference/model.py:415-423; Chan et al.), in fp32 like the reference.  `stats`: iterable of (count, mean[u], var[u]);
    triples with count 0 are skipped.  Returns (count, mean, var)."""
    f = np.float32
    count, mean, var = f(0), None, None
    for c, m, v in stats:
        c, m, v = f(c), np.asarray(m, f), np.asarray(v, f)
        if c == 0:
            continue
        if mean is None:       # the reference starts from count = 0, mean = 0, var = 0 and applies the same update
            mean, var = np.zeros_like(m), np.zeros_like(v)
        delta = m - mean
        m2 = var * count + v * c + delta ** 2 * count * c / (count + c)
        count = f(count + c)
        mean = (mean + delta * c / count).astype(f)
        var = (m2 / count).astype(f)
    return count, mean, var


def allgather_prenorm(count, mean, var, received: bool, process_group, device):
    """Data-parallel PreNorm fitting: every rank contributes the triple it accumulated over ITS batches;
    the triples are merged in rank order, so all ranks end up with bit-identical statistics.  One small all-gather per layer.
    Returns (count, mean, var, received_anywhere)."""
    import torch
    import torch.distributed as dist
    units = len(mean)
    vec = torch.tensor(np.concatenate([[float(received), float(count)], mean, var]).astype(np.float32))
    on_gpu = dist.get_backend(process_group) == "nccl"
    if on_gpu:
        vec = vec.to(device)
    out = [torch.empty_like(vec) for _ in range(dist.get_world_size(process_group))]
    dist.all_gather(out, vec, group=process_group)
    rows = [t.cpu().numpy() for t in out]
    c, m, v = chan_merge((r[1], r[2:2 + units], r[2 + units:2 + 2 * units]) for r in rows)
    if m is None:
        m, v = np.zeros(units, np.float32), np.zeros(units, np.float32)
    return c, m, v, any(r[0] != 0 for r in rows)


def allgather_concat(t, process_group):
    """Ragged all-gather along dim 0: every rank passes a tensor with its own number of rows (same trailing shape and dtype),
    every rank gets the concatenation in rank order.  One small all-gather of the row counts, one of the tensors padded to the
    longest (over RCCL the payload stays on the device; a gloo group moves it through host memory)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    on_gpu = dist.get_backend(process_group) == "nccl"
    dev = t.device
    work = t if (on_gpu or not t.is_cuda) else t.cpu()
    n = torch.tensor([work.shape[0]], dtype=torch.int64, device=work.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=process_group)
    counts = [int(c) for c in counts]
    longest = max(counts)
    if longest == 0:
        return t
    padded = work.new_zeros((longest,) + tuple(work.shape[1:]))
    padded[:work.shape[0]] = work
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=process_group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).to(dev)


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
