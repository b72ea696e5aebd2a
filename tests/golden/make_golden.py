#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle (fp64).  The reference ships no golden vectors and cannot run here
(TensorFlow absent), so these fixtures pin the ORACLE (and, through it, the HIP path) against regressions; they are not
outputs of the reference itself -- parity with the reference stays "unpinned" (see oracle/gcnn_oracle.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gcnn_cut_selector_amd import synthetic  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STATE_KEYS = ["cons_feats", "cons_edge_inds", "cons_edge_feats", "var_feats", "cut_feats", "cut_edge_inds", "cut_edge_feats"]


def pack_state(prefix, state):
    out = {f"{prefix}{k}": np.asarray(v) for k, v in zip(STATE_KEYS, state[:7])}
    out[f"{prefix}counts"] = np.array(state[7:10], np.int64)
    return out


def main():
    # --- case 1: forward / loss / gradients / one Keras-Adam step on a small stacked setcov batch -------------------------
    params = O.randomize_params(O.init_params(11, np.float64), 12)
    state, y, n_cuts = synthetic.make_batch("setcov", 3, first_sample=40, scale=0.06)
    pred, loss, grads = O.loss_and_grads(params, state, y, torch.float64)
    out = pack_state("in_", state)
    out["targets"] = y.astype(np.float64); out["n_cuts_per_sample"] = n_cuts
    for name in O.PARAM_NAMES:
        out["w_" + name.replace("/", "__")] = params[name].astype(np.float32)  # the weights ARE fp32 values
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}
    pred, loss, grads = O.loss_and_grads(params, state, y, torch.float64)
    out["scores"] = pred; out["loss"] = np.float64(loss)
    lr = 1e-3
    for name, g in grads.items():
        out["g_" + name.replace("/", "__")] = g.astype(np.float32)
        th, m, v = O.keras_adam_step(params[name], g, np.zeros_like(g), np.zeros_like(g), 1, lr)
        th2, _, _ = O.keras_adam_step(th, g, m, v, 2, lr)  # second step with the same gradient: exercises m/v and t=2
        out["a1_" + name.replace("/", "__")] = th.astype(np.float32)
        out["a2_" + name.replace("/", "__")] = th2.astype(np.float32)
    out["adam_lr"] = np.float64(lr)
    np.savez_compressed(os.path.join(HERE, "setcov_small.npz"), **out)

    # --- case 2: PreNorm pretraining over three batches (the 58 fitted scalars) ---------------------------------------
    p0 = {k: v.astype(np.float32).astype(np.float64) for k, v in O.init_params(21, np.float64).items()}
    batches = [synthetic.make_batch("combauc", 2, first_sample=60 + 2 * i)[0] for i in range(3)]
    fitted, n = O.pretrain(p0, batches, torch.float64)
    assert n == 11
    out = {}
    for b, st in enumerate(batches):
        out.update(pack_state(f"b{b}_", st))
    for name in O.PARAM_NAMES:
        out["w_" + name.replace("/", "__")] = p0[name].astype(np.float32)
    for shift, scale, _ in O.PRENORM_LAYERS:
        if shift:
            out["fit_" + shift.replace("/", "__")] = fitted[shift]
        out["fit_" + scale.replace("/", "__")] = fitted[scale]
    np.savez_compressed(os.path.join(HERE, "pretrain_combauc.npz"), **out)
    for f in ("setcov_small.npz", "pretrain_combauc.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
