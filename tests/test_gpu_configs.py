"""Every BASELINE.json shape under parity at its REAL size (GPU): setcov-500 x 32, combauc 100/500 x 32, capfac 100 x 32,
indset-750 x 64.

  * inference kernels (torch.no_grad(): k_edge_fwd<*, false>, row programs with their stores skipped) against the fp64
    oracle on all four problems, incl. the full-size batches;
  * capfac at its real instance size (10,201 rows of length {2, 100, 101} per sample: SLOTS = 1 edge kernels, hub rows,
    650 k-row row programs) forward AND backward against the oracle, at a batch that also reaches the scaled weight-gradient
    chunks (rows_per_wave > WG_ROWS in gcnn_backward);
  * at the full batch sizes: loss and all 46 gradients against the fp64 autograd oracle, and size-independent properties
    -- determinism, disjoint-union batching invariance, finite gradients, linearity in d_scores, additivity of the SUM-loss
    gradient over sample shards.
Tolerances: scores 1e-4 absolute/relative (BASELINE.json north star); gradients as in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gcnn_cut_selector_amd import synthetic  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)

from test_gpu_model import _grad_check, _model  # noqa: E402

FULL = [("setcov", 32), ("combauc", 32), ("capfac", 32), ("indset", 64)]   # BASELINE.json configs[1..4]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _oracle_scores(params, state):
    return O.scores({k: v.astype(np.float64) for k, v in params.items()}, state, torch.float64)


@pytest.mark.parametrize("problem,batch", [("setcov", 4), ("combauc", 8), ("capfac", 2), ("indset", 4)])
def test_inference_kernels_match_oracle(dev, problem, batch):
    """The no-grad path (what the SCIP plugin and validation run) vs the oracle -- not only vs the saving path."""
    m, params = _model(50, dev)
    state, _, _ = synthetic.make_batch(problem, batch)
    with torch.no_grad():
        got = m(state, False).numpy()
    np.testing.assert_allclose(got, _oracle_scores(params, state), rtol=1e-4, atol=1e-4)
    with torch.enable_grad():
        saved = m(state, True).numpy()
    # the two paths may give a segment a different number of lanes (inference on small graphs: a whole wave each), i.e. sum
    # its edges in a different order: equal up to fp32 rounding
    np.testing.assert_allclose(got, saved, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("problem,batch", FULL)
def test_full_batch_forward_matches_oracle(dev, problem, batch):
    m, params = _model(51, dev)
    state, _, _ = synthetic.make_batch(problem, batch)
    with torch.no_grad():
        got = m(state, False).numpy()
    np.testing.assert_allclose(got, _oracle_scores(params, state), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("batch", [2, 8])
def test_capfac_real_size_forward_backward(dev, batch):
    """capfac 100x100 instances (BASELINE configs[3]); batch 8 = 81,608 + 80,800 rows: rows_per_wave = 256 in k_wgrad."""
    m, params = _model(52, dev)
    state, y, _ = synthetic.make_batch("capfac", batch)
    assert state[0].shape[0] == 10201 * batch and state[1].shape[1] == 40200 * batch
    got = m(state, True).numpy()
    np.testing.assert_allclose(got, _oracle_scores(params, state), rtol=1e-4, atol=1e-4)
    _grad_check(m, params, state, y)


@pytest.mark.parametrize("problem,batch", [("indset", 8), ("combauc", 32)])
def test_backward_parity_larger_batches(dev, problem, batch):
    m, params = _model(53, dev)
    state, y, _ = synthetic.make_batch(problem, batch)
    _grad_check(m, params, state, y)


@pytest.mark.parametrize("problem,batch", FULL)
def test_full_batch_backward_matches_oracle(dev, problem, batch):
    """Loss and all 46 gradients at the BASELINE batch sizes against the fp64 autograd oracle (10-20 GB of host memory for
    the [E,64] tensors the reference dataflow materialises; seconds on the GPU box's host cores)."""
    m, params = _model(55, dev)
    state, y, _ = synthetic.make_batch(problem, batch)
    _grad_check(m, params, state, y)


def _sum_loss_grads(m, batch, y):
    """Gradient of the local SUM of squared errors (the data-parallel convention of trainer.train_step), unfused path."""
    from gcnn_cut_selector_amd.trainer import mse_loss
    flat = m.flat_parameters.detach()
    ws = m._take_workspace(batch)
    scores = m._forward_into(flat, batch, ws)
    loss, d = mse_loss(scores, y, 1.0)
    g = torch.zeros_like(flat)
    m._backward_into(flat, batch, ws, d, g)
    m._give_workspace(ws)
    return float(loss), g


@pytest.mark.parametrize("problem,batch", FULL[1:])     # setcov x 32: tests/test_gpu_train.py
def test_full_batch_properties(dev, problem, batch):
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    m, _ = _model(54, dev)
    samples = [synthetic.make_sample(problem, i) for i in range(batch)]
    full = synthetic.stack_samples(samples)
    totals = lambda b: (int(b[7].sum()), int(b[8].sum()), int(b[9].sum()))
    prepared = m.prepare(full[:7] + totals(full))
    with torch.no_grad():
        a = m(prepared, False).numpy()
        assert np.array_equal(a, m(prepared, False).numpy())              # deterministic (no float atomics)
        nk = full[9]
        for sl, s in ((slice(0, nk[0]), samples[:1]), (slice(len(a) - nk[-1], len(a)), samples[-1:])):
            one = synthetic.stack_samples(s)
            np.testing.assert_allclose(a[sl], m(one[:7] + totals(one), False).numpy(), rtol=1e-5, atol=1e-5)
    ts = TrainState(m)
    yt = torch.as_tensor(full[10]).to(dev)
    l1, s1 = train_step(m, prepared, yt, None, ts); g1 = ts.grads.clone()
    l2, _ = train_step(m, prepared, yt, None, ts)
    assert torch.equal(g1, ts.grads) and torch.equal(l1, l2)             # bitwise reproducible loss and gradients
    # saving path vs inference path: the same function; the inference forward gives small row sets a whole wave per segment while
    # the training forward picks the lane group by mean degree (and splits hub rows over four waves), so segment sums are
    # added in another (fixed) order
    np.testing.assert_allclose(s1.cpu().numpy(), a, rtol=1e-4, atol=1e-5)
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    # linearity of the backward pass in d_scores
    flat = m.flat_parameters.detach()
    ws = m._take_workspace(prepared)
    scores = m._forward_into(flat, prepared, ws)
    d = torch.randn_like(scores)
    ga, gb = torch.zeros_like(flat), torch.zeros_like(flat)
    m._backward_into(flat, prepared, ws, d, ga)
    m._forward_into(flat, prepared, ws)
    m._backward_into(flat, prepared, ws, 2 * d, gb)
    m._give_workspace(ws)
    np.testing.assert_allclose(gb.cpu().numpy(), 2 * ga.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(ga.abs().max()))
    # additivity over a disjoint union: SUM-loss gradient of the full batch == sum over 4 shards (different chunking of every
    # row program, edge pass and weight-gradient job); mean-loss gradient == that / n_cuts
    loss_full, g_full = _sum_loss_grads(m, prepared, yt)
    acc, loss_acc = torch.zeros_like(g_full, dtype=torch.float64), 0.0
    step = batch // 4
    for i in range(0, batch, step):
        part = synthetic.stack_samples(samples[i:i + step])
        lp, gp = _sum_loss_grads(m, m.prepare(part[:7] + totals(part)), torch.as_tensor(part[10]).to(dev))
        acc += gp.double(); loss_acc += lp
    scale = float(g_full.abs().max())
    np.testing.assert_allclose(g_full.cpu().numpy(), acc.cpu().numpy(), rtol=2e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(loss_full, loss_acc, rtol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), (acc / len(a)).cpu().numpy(), rtol=2e-4, atol=2e-5 * scale / len(a))
