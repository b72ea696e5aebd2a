"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the oracle still reproduces them (regression pin of the checker).  GPU: the HIP path reproduces them."""
import os

import numpy as np
import pytest
import torch

from oracle import gcnn_oracle as O

STATE_KEYS = ["cons_feats", "cons_edge_inds", "cons_edge_feats", "var_feats", "cut_feats", "cut_edge_inds", "cut_edge_feats"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _state(z, prefix):
    c = z[prefix + "counts"]
    return tuple(z[prefix + k] for k in STATE_KEYS) + (int(c[0]), int(c[1]), int(c[2]))


def _weights(z, dtype=np.float64):
    return {n: z["w_" + n.replace("/", "__")].astype(dtype) for n in O.PARAM_NAMES}


def test_oracle_reproduces_golden_scores_loss_grads(golden_dir):
    z = _load(golden_dir, "setcov_small.npz")
    pred, loss, grads = O.loss_and_grads(_weights(z), _state(z, "in_"), z["targets"], torch.float64)
    np.testing.assert_allclose(pred, z["scores"], rtol=1e-10, atol=1e-12)
    assert abs(loss - float(z["loss"])) <= 1e-10 * max(1.0, abs(loss))
    for name, g in grads.items():
        ref = z["g_" + name.replace("/", "__")]
        np.testing.assert_allclose(g, ref, rtol=2e-6, atol=2e-6 * max(np.abs(ref).max(), 1e-30), err_msg=name)


def test_oracle_reproduces_golden_adam(golden_dir):
    z = _load(golden_dir, "setcov_small.npz")
    w = _weights(z)
    for name, _, trainable in O.PARAM_SPEC:
        if not trainable:
            continue
        key = name.replace("/", "__")
        g = z["g_" + key].astype(np.float64)
        th, m, v = O.keras_adam_step(w[name], g, np.zeros_like(g), np.zeros_like(g), 1, float(z["adam_lr"]))
        th2, _, _ = O.keras_adam_step(th, g, m, v, 2, float(z["adam_lr"]))
        np.testing.assert_allclose(th, z["a1_" + key], rtol=1e-6, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(th2, z["a2_" + key], rtol=1e-6, atol=1e-7, err_msg=name)


def test_oracle_reproduces_golden_pretrain(golden_dir):
    z = _load(golden_dir, "pretrain_combauc.npz")
    fitted, n = O.pretrain(_weights(z), [_state(z, f"b{b}_") for b in range(3)], torch.float64)
    assert n == 11
    for shift, scale, _ in O.PRENORM_LAYERS:
        for name in (shift, scale):
            if name:
                np.testing.assert_allclose(fitted[name], z["fit_" + name.replace("/", "__")], rtol=1e-9, err_msg=name)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_hip_reproduces_golden_step(dev, golden_dir):
    """Scores within 1e-4 (BASELINE.json north star); loss, all 46 gradients and two fused Keras-Adam steps."""
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
    z = _load(golden_dir, "setcov_small.npz")
    m = GCNN(device=dev)
    m.set_weights([z["w_" + n.replace("/", "__")] for n in O.PARAM_NAMES])
    batch = m.prepare(_state(z, "in_"))
    np.testing.assert_allclose(m(batch, False).numpy(), z["scores"], rtol=1e-4, atol=1e-4)
    opt, ts = Adam(float(z["adam_lr"])), TrainState(m)
    y = torch.as_tensor(z["targets"], dtype=torch.float32).to(dev)
    loss, _ = train_step(m, batch, y, None, ts)       # gradients only
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * max(1.0, float(z["loss"]))
    grads = [g.cpu().numpy() for g in m.gradients(ts.grads)]
    names = [n for n, _, t in O.PARAM_SPEC if t]
    for name, g in zip(names, grads):
        ref = z["g_" + name.replace("/", "__")]
        assert np.abs(g - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-6) + 1e-7, name
    # two optimizer steps with the GOLDEN gradient (isolates the Adam kernel from gradient noise)
    flat_g = torch.zeros_like(m.flat_parameters.detach())
    for (off, rows, cols, tr), name in zip(m._layout, O.PARAM_NAMES):
        if tr:
            flat_g[off:off + rows * cols] = torch.from_numpy(z["g_" + name.replace("/", "__")].reshape(-1)).to(dev)
    opt.apply_flat(m, flat_g)
    w1 = m.get_weights()
    opt.apply_flat(m, flat_g)
    w2 = m.get_weights()
    for name, a1, a2, (_, _, tr) in zip(O.PARAM_NAMES, w1, w2, O.PARAM_SPEC):
        key = name.replace("/", "__")
        if tr:
            np.testing.assert_allclose(a1, z["a1_" + key], rtol=2e-6, atol=2e-7, err_msg=name)
            np.testing.assert_allclose(a2, z["a2_" + key], rtol=2e-6, atol=2e-7, err_msg=name)
        else:  # PreNorm variables are not trainable: untouched
            np.testing.assert_array_equal(a1, z["w_" + key])


@pytest.mark.gpu
def test_hip_reproduces_golden_pretrain(dev, golden_dir):
    """The 58 fitted PreNorm scalars (11 layers, fitted one at a time over three batches), rtol 1e-4."""
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.trainer import pretrain
    z = _load(golden_dir, "pretrain_combauc.npz")
    m = GCNN(device=dev)
    m.set_weights([z["w_" + n.replace("/", "__")] for n in O.PARAM_NAMES])
    batches = []
    for b in range(3):
        st = _state(z, f"b{b}_")
        batches.append(st[:7] + (np.array([st[7]]), np.array([st[8]]), np.array([st[9]]), np.zeros(st[9], np.float32)))
    assert pretrain(m, batches) == 11
    for shift, scale, _ in O.PRENORM_LAYERS:
        for name in (shift, scale):
            if name:
                got = m.get_variable(name).cpu().numpy()
                np.testing.assert_allclose(got, z["fit_" + name.replace("/", "__")], rtol=1e-4, atol=1e-6, err_msg=name)
