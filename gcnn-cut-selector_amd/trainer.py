"""Training-step counterpart of the reference's `model_trainer.py` for the HIP GCNN.

Mirrors (file:line in /root/reference):
  process(model, dataloader, fractions, loss_fn, optimizer)  model_trainer.py:239-316  -> process() (same positional order)
  pretrain(model, dataloader)                                 model_trainer.py:194-236  -> pretrain()
  MeanSquaredError / Adam(learning_rate=lambda: lr)           model_trainer.py:131-132  -> mse_loss() / Adam
  ranking-prefix accuracy                                     model_trainer.py:280-302  -> ranking_fraction()
`train_step` is the fused fast path (no autograd graph): forward -> MSE head -> backward -> [RCCL all-reduce of ONE flat
gradient buffer] -> Keras-form Adam, all on the current HIP stream.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .graph import _ptr, _stream
from .model import GCNN, Batch
from .store import StoreBatch


def mse_loss(scores: torch.Tensor, targets: torch.Tensor, scale: float | None = None, want_grad=True):
    """Keras `MeanSquaredError` on 1-D inputs (model_trainer.py:132,271).  Returns (loss[1], d_scores|None)."""
    n = scores.numel()
    scale = (1.0 / n if n else 0.0) if scale is None else scale
    loss = torch.empty(1, dtype=torch.float32, device=scores.device)
    d = torch.empty_like(scores, memory_format=torch.contiguous_format) if want_grad else None
    with torch.cuda.device(scores.device):
        _lib.check(_lib.lib().gcnn_mse_loss(_ptr(scores), _ptr(targets), n, scale, _ptr(loss), _ptr(d),
                                            _stream(scores.device)), "gcnn_mse_loss")
    return loss, d


class Adam:
    """Keras-2.7 Adam (model_trainer.py:131): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps), eps=1e-7.
    One fused kernel over the model's flat parameter buffer.  `learning_rate` may be a float or a zero-arg callable
    (the reference passes `lambda: lr` so the plateau schedule can change it)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations = 0
        self.m = self.v = None
        self._dev = None   # device-resident {lr, b1, b2, eps, t, lr_t} for graph replay

    def _lr(self):
        return float(self.learning_rate() if callable(self.learning_rate) else self.learning_rate)

    def apply_flat(self, model: GCNN, flat_grad: torch.Tensor, grad_scale: torch.Tensor | None = None, divide=False):
        """grad_scale: optional device scalar multiplying (divide=False) or dividing (divide=True) every gradient."""
        flat = model.flat_parameters.detach()
        if self.m is None:
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.iterations += 1
        t = self.iterations
        lr_t = self._lr() * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        with torch.cuda.device(flat.device):
            _lib.check(_lib.lib().gcnn_adam_step(_ptr(flat), _ptr(flat_grad), _ptr(self.m), _ptr(self.v), flat.numel(),
                                                 lr_t, self.beta_1, self.beta_2, self.epsilon, _ptr(grad_scale),
                                                 int(divide), _stream(flat.device)), "gcnn_adam_step")

    def fused_args(self, model: GCNN):
        """Advance the step counter and return the `gcnn_adam_args` for `GCNN._backward_into(adam=...)`: the same update as
        `apply_flat`, executed by the backward pass's last launch instead of a launch of its own."""
        flat = model.flat_parameters.detach()
        if self.m is None:
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.iterations += 1
        t = self.iterations
        lr_t = self._lr() * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        return _lib.AdamArgs(flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), lr_t, self.beta_1, self.beta_2, self.epsilon)

    def apply_flat_dev(self, model: GCNN, flat_grad: torch.Tensor, grad_scale: torch.Tensor | None = None, divide=False):
        """The same update with hyper-parameters and step counter resident on the device (gcnn_adam_step_dev): nothing
        step-dependent crosses the host, so the call can sit inside a captured hipGraph and be replayed."""
        flat = model.flat_parameters.detach()
        if self.m is None:
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        if self._dev is None:
            self._dev = torch.tensor([self._lr(), self.beta_1, self.beta_2, self.epsilon, float(self.iterations), 0.0],
                                     dtype=torch.float32, device=flat.device)
        with torch.cuda.device(flat.device):
            _lib.check(_lib.lib().gcnn_adam_step_dev(_ptr(flat), _ptr(flat_grad), _ptr(self.m), _ptr(self.v), flat.numel(),
                                                     _ptr(self._dev), _ptr(grad_scale), int(divide), _stream(flat.device)),
                       "gcnn_adam_step_dev")

    def sync_from_device(self):
        """After graph replays: pull the step counter back; push the (possibly changed) learning rate."""
        if self._dev is not None:
            self.iterations = int(self._dev[4].item())
            self._dev[0] = self._lr()

    def apply_gradients(self, model: GCNN):
        """After `loss.backward()`: update from `model.flat_parameters.grad` (the reference's
        `optimizer.apply_gradients(zip(grads, model.trainable_variables))`, model_trainer.py:273)."""
        g = model.flat_parameters.grad
        if g is None:
            raise RuntimeError("no gradients: call loss.backward() first")
        self.apply_flat(model, g)


class TrainState:
    """Buffers reused across steps by `train_step` (flat gradient + the data-parallel count slot)."""

    def __init__(self, model: GCNN):
        n = model.flat_parameters.numel()
        # [gradients | local cut count | pad]: ONE buffer => ONE all-reduce per step
        self.buf = torch.zeros(n + 4, dtype=torch.float32, device=model.device)
        self.grads = self.buf[:n]
        self.count = self.buf[n:n + 1]


def train_step(model: GCNN, batch: Batch, targets: torch.Tensor, optimizer: Adam | None, state: TrainState,
               process_group=None, device_optimizer=False):
    """One training step on a prepared batch: forward + MSE + backward (+ all-reduce) (+ Adam).  Returns (loss, scores).

    Single GPU: loss = mean over this batch's cuts (model_trainer.py:271).  Data parallel (`process_group` given): each
    rank back-propagates the local SUM of squared errors; gradients and cut counts are summed with one RCCL all-reduce
    and Adam divides by the global cut count -- the mean over ALL cuts of the global batch, not a mean of per-rank means."""
    flat = model.flat_parameters.detach()
    ws = model._take_workspace(batch)
    n_cuts = batch.dims.n_cuts
    if targets.dtype != torch.float32 or not targets.is_contiguous():
        targets = targets.to(torch.float32).contiguous()
    if targets.numel() != n_cuts:
        raise ValueError(f"expected {n_cuts} targets, got {targets.numel()}")
    loss = torch.empty(1, dtype=torch.float32, device=model.device)
    # The MSE head rides in the forward's last launch (gcnn_forward_loss), so the backward starts at the readout's hidden layer
    if process_group is None:
        scores = model._forward_loss_into(flat, batch, ws, targets, 1.0 / max(n_cuts, 1))
        fuse = optimizer is not None and not device_optimizer   # host-parameterised Adam: rides in the backward's last launch
        model._backward_into(flat, batch, ws, None, state.grads, loss_out=loss, adam=optimizer.fused_args(model) if fuse else None)
        model._give_workspace(ws)
        if optimizer is not None and not fuse:
            optimizer.apply_flat_dev(model, state.grads)
        return loss, scores
    import torch.distributed as dist
    scores = model._forward_loss_into(flat, batch, ws, targets, 1.0)   # local SUM of squared errors
    model._backward_into(flat, batch, ws, None, state.grads, count_slot=state.count, loss_out=loss)  # also stores the local cut count
    model._give_workspace(ws)
    dist.all_reduce(state.buf, op=dist.ReduceOp.SUM, group=process_group)   # ONE collective: gradients + cut count
    if optimizer is not None:  # Adam divides by the global cut count: the mean over ALL cuts of the global batch
        (optimizer.apply_flat_dev if device_optimizer else optimizer.apply_flat)(model, state.grads, grad_scale=state.count,
                                                                              divide=True)
    return loss, scores


class GraphedTrainStep:
    """One training step on a FIXED prepared batch, captured once into a hipGraph and replayed.

    A step is 20 kernel launches on one stream; capture (torch.cuda.CUDAGraph over the same `train_step`) removes the host
    from the loop.  Everything step-dependent lives on the device (Adam's step counter and learning rate:
    `Adam.apply_flat_dev`).  Measured: no faster than eager issue -- the GPU-side launch sequence is the limit.
    The graph is tied to the batch's buffers and sizes: use it when batches have a fixed shape / are replayed (benchmarks,
    fixed-capacity loaders); variable-shape training uses the eager `train_step`."""

    def __init__(self, model: GCNN, batch: Batch, targets: torch.Tensor, optimizer: Adam | None, state: TrainState,
                 process_group=None, warmup=2):
        self.optimizer = optimizer
        args = (model, batch, targets, optimizer, state, process_group, True)
        cur = torch.cuda.current_stream(model.device)
        side = torch.cuda.Stream(device=model.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):           # eager warm-up: lazy HIP state (streams, events, attributes) and buffers
            for _ in range(warmup):
                train_step(*args)
        cur.wait_stream(side)
        torch.cuda.synchronize(model.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.scores = train_step(*args)
        self.steps = 0

    def __call__(self):
        self.graph.replay()
        self.steps += 1
        return self.loss, self.scores



def ranking_fraction(pred: np.ndarray, true: np.ndarray) -> float:
    """model_trainer.py:288-301: length of the ranking prefix on which prediction and truth agree, over #cuts.
    Python's `sorted(..., reverse=True)` is stable, i.e. ties keep index order: a stable argsort of the negated key."""
    pr = np.argsort(-np.asarray(pred), kind="stable")
    tr = np.argsort(-np.asarray(true), kind="stable")
    diff = pr != tr
    return (int(np.argmax(diff)) if diff.any() else len(pr)) / len(pr)


def ranking_metric(pred: torch.Tensor, true: torch.Tensor, n_cuts, fractions_dev: torch.Tensor, acc: torch.Tensor,
                   loss: torch.Tensor | None = None, loss_acc: torch.Tensor | None = None, loss_weight: float | None = None):
    """Device-side ranking-prefix accuracy (gcnn_ranking_metric): acc[f] += #samples with frac >= fractions[f]; optionally
    loss_acc += loss * loss_weight (default: total_cuts, the cut-weighted loss of model_trainer.py:304).  Returns per-sample
    fractions (device)."""
    n_cuts = np.asarray(n_cuts, dtype=np.int64).reshape(-1)
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(n_cuts)]).astype(np.int32)).to(pred.device, non_blocking=True)
    frac = torch.empty(len(n_cuts), dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(_lib.lib().gcnn_ranking_metric(_ptr(pred), _ptr(true), _ptr(offsets), len(n_cuts),
                                                  int(n_cuts.max()) if len(n_cuts) else 0, _ptr(fractions_dev),
                                                  fractions_dev.numel(), _ptr(acc), _ptr(frac), _ptr(loss),
                                                  float(n_cuts.sum()) if loss_weight is None else float(loss_weight), _ptr(loss_acc),
                                                  _stream(pred.device)),
                   "gcnn_ranking_metric")
    return frac


def _unpack_batch(model: GCNN, batch):
    """`utils.load_batch` 11-tuple or `SampleStore` batch -> (prepared Batch, per-sample n_cuts, device targets)."""
    if isinstance(batch, StoreBatch):     # collated on the device by a SampleStore: nothing left to move
        return batch.batch, np.asarray(batch.n_cuts).reshape(-1), batch.improvements
    (c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts, improvements) = batch
    n_cuts = np.asarray(n_cuts).reshape(-1)
    prepared = model.prepare((c, cei, cef, v, k, kei, kef, int(np.sum(n_cons)), int(np.sum(n_vars)), int(n_cuts.sum())))
    y = torch.as_tensor(np.asarray(improvements), dtype=torch.float32).to(model.device, non_blocking=True)
    return prepared, n_cuts, y


def process(model: GCNN, dataloader, fractions: np.ndarray, loss_fn=None, optimizer: Adam | None = None, *,
            process_group=None):
    """Counterpart of model_trainer.process (model_trainer.py:239-316), same positional order:
    `process(model, dataloader, fractions, loss_fn, optimizer=None)`.  `loss_fn` keeps the reference's slot so its call sites
    (model_trainer.py:156,161,182) bind unchanged; the loss is always the reference's `MeanSquaredError`
    (model_trainer.py:132) evaluated by the fused loss head, so a callable (or None) is accepted and not called.
    `dataloader` yields the 11-tuples of `utils.load_batch` (per-sample count vectors + improvements) or `SampleStore`
    batches.  Returns (cut-weighted mean loss, accuracy per fraction).  Loss and ranking accuracy accumulate ON THE DEVICE;
    the host reads them once at the end (no per-batch sync).

    Data parallel (`process_group` given, keyword only): every rank iterates over ITS shard of each global batch (e.g.
    `store.batches(ids, batch_size, rank, world_size)`, the same number of batches on every rank, empty shards included);
    gradients are all-reduced per step (`train_step`) and the returned loss / accuracies are those of the GLOBAL data."""
    if isinstance(loss_fn, Adam):
        raise TypeError("process(model, dataloader, fractions, loss_fn, optimizer): the fourth positional argument is the "
                        "reference's loss_fn slot; pass the optimizer fifth (or optimizer=...)")
    if loss_fn is not None and not callable(loss_fn):
        raise TypeError("loss_fn must be None or a callable (it is accepted for call compatibility and not called)")
    dev = model.device
    fractions = np.asarray(fractions, dtype=np.float32)
    frac_dev = torch.from_numpy(fractions).to(dev)
    acc_dev = torch.zeros(len(fractions), dtype=torch.float32, device=dev)
    loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
    host_acc, host_loss = np.zeros(len(fractions)), 0.0   # samples too large for the device metric (> 4096 cuts)
    n_samples = cut_count = 0
    state = TrainState(model) if optimizer is not None else None
    for batch in dataloader:
        try:
            prepared, n_cuts, y = _unpack_batch(model, batch)
            total = int(n_cuts.sum())
            if optimizer is not None:
                # data parallel: `loss` is the local SUM of squared errors (train_step), else the mean over this batch
                loss, predictions = train_step(model, prepared, y, optimizer, state, process_group=process_group)
                weight = 1.0 if process_group is not None else float(total)
            else:
                with torch.no_grad():
                    predictions = model(prepared, False)
                loss, _ = mse_loss(predictions, y, want_grad=False)
                weight = float(total)
            if len(n_cuts) == 0:
                pass
            elif n_cuts.max() <= 4096:
                ranking_metric(predictions.detach().as_subclass(torch.Tensor), y, n_cuts, frac_dev, acc_dev, loss, loss_dev, weight)
            else:
                pred, true = predictions.detach().cpu().numpy(), y.cpu().numpy()
                start = 0
                for nk in n_cuts:
                    host_acc += ranking_fraction(pred[start:start + nk], true[start:start + nk]) >= fractions
                    start += nk
                host_loss += float(loss) * weight
            n_samples += len(n_cuts)
            cut_count += total
        except torch.OutOfMemoryError:  # the reference skips batches that exhaust memory (model_trainer.py:308-311)
            if process_group is not None:
                # Data parallel: the peers are in (or heading for) this step's all-reduce; a rank that skipped on its own
                # would pair that collective with its next batch.  Agreeing on a skip costs a blocking collective per batch,
                # so the error propagates instead (with the store resident in HBM a batch that does not fit is a sizing bug).
                raise
            print("WARNING: batch skipped.")
    totals = torch.cat([loss_dev.double() + host_loss, acc_dev.double() + torch.from_numpy(host_acc).to(dev),
                        torch.tensor([float(n_samples), float(cut_count)], dtype=torch.float64, device=dev)])
    if process_group is not None:
        import torch.distributed as dist
        dist.all_reduce(totals, op=dist.ReduceOp.SUM, group=process_group)
    totals = totals.cpu().numpy()
    mean_loss = float(totals[0]) / max(totals[-1], 1.0)
    mean_acc = totals[1:-2] / max(totals[-2], 1.0)
    return mean_loss, mean_acc


def pretrain(model: GCNN, dataloader, process_group=None):
    """Counterpart of model_trainer.pretrain (model_trainer.py:194-236): fit PreNorm layers one at a time.  `dataloader` is
    iterated once per layer (a list, or any re-iterable).  Data parallel (`process_group` given): every rank passes ITS shard of
    the pretraining batches (possibly none); after each pass the ranks merge their streaming statistics with one small
    all-gather (`GCNN.pretrain_sync`), so all ranks freeze identical shift / scale values -- 11 passes over 1/N of the data
    each instead of every rank redoing all of it."""
    model.pretrain_init()
    i = 0
    while True:
        for batch in dataloader:
            if isinstance(batch, StoreBatch):
                inputs = batch.batch
            else:
                inputs = tuple(batch[:7]) + (int(np.sum(batch[7])), int(np.sum(batch[8])), int(np.sum(batch[9])))
            try:
                if not model.pretrain(inputs, True):
                    break
            except torch.OutOfMemoryError:
                print("WARNING: batch skipped.")
        model.pretrain_sync(process_group)
        if model.pretrain_next() is None:
            break
        i += 1
    return i
