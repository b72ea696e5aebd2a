"""Test-set counterpart of the reference's `model_tester.process` (/root/reference/model_tester.py:173-237) for the HIP GCNN.

The reference's tester differs from the trainer's validation pass (model_trainer.py:239-316) in what it returns: the
cut-weighted mean squared error (model_tester.py:199, 234) and the MEAN ranking fraction over all samples (`acc += frac`,
model_tester.py:224, 235) instead of thresholded accuracies.  Forward only, inference kernels (nothing is stored for a
backward pass); loss and fractions accumulate on the device and are read once at the end.
"""

from __future__ import annotations

import numpy as np
import torch

from .model import GCNN
from .trainer import _unpack_batch, mse_loss, ranking_fraction, ranking_metric


def process(model: GCNN, dataloader):
    """`model_tester.process(model, dataloader)` (model_tester.py:173-237): returns (loss, mean_acc) where
    loss = sum_b n_cuts_b * MSE_b / sum_b n_cuts_b and mean_acc = mean over samples of the ranking-prefix fraction
    (first position where the predicted and the true descending rankings differ, over the number of cuts).
    `dataloader` yields `utils.load_batch` 11-tuples or `SampleStore` batches."""
    dev = model.device
    none = torch.zeros(0, dtype=torch.float32, device=dev)          # no thresholds: only the per-sample fractions
    loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
    frac_dev = torch.zeros(1, dtype=torch.float64, device=dev)
    host_frac, host_loss = 0.0, 0.0                                  # samples too large for the device metric (> 4096 cuts)
    n_samples = cut_count = 0
    for batch in dataloader:
        try:
            prepared, n_cuts, y = _unpack_batch(model, batch)
            total = int(n_cuts.sum())
            with torch.no_grad():
                predictions = model(prepared, False)                 # model_tester.py:198
            loss, _ = mse_loss(predictions, y, want_grad=False)
            if len(n_cuts) == 0:
                pass
            elif n_cuts.max() <= 4096:
                frac = ranking_metric(predictions.detach().as_subclass(torch.Tensor), y, n_cuts, none, none, loss, loss_dev,
                                      float(total))
                frac_dev += frac.double().sum()
            else:
                pred, true = predictions.detach().cpu().numpy(), y.cpu().numpy()
                start = 0
                for nk in n_cuts:
                    host_frac += ranking_fraction(pred[start:start + nk], true[start:start + nk])
                    start += nk
                host_loss += float(loss) * total
            n_samples += len(n_cuts)
            cut_count += total
        except torch.OutOfMemoryError:   # model_tester.py:229-232
            print("WARNING: batch skipped.")
    loss = (float(loss_dev) + host_loss) / max(cut_count, 1)
    mean_acc = (float(frac_dev) + host_frac) / max(n_samples, 1)
    return loss, mean_acc
