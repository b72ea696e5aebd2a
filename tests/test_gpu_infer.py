"""Single-state inference (GPU): `GCNN.get_concrete_function()` / `score_state` -- the SCIP cut selector's call shape
(model_evaluator.py:82-111) through the one-call path gcnn_infer -- against the oracle, against the general path bit for bit,
and its ranking output against the reference's `sorted(range(n), key=quality, reverse=True)`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gcnn_cut_selector_amd import synthetic, utils  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)

from test_gpu_model import _model  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _oracle(params, inp):
    return O.scores({k: v.astype(np.float64) for k, v in params.items()}, inp, torch.float64)


@pytest.mark.parametrize("problem", ["setcov", "combauc", "capfac", "indset"])
def test_concrete_function_matches_oracle_and_general_path(dev, problem):
    m, params = _model(80, dev)
    f = m.get_concrete_function()
    for i in (3, 4):
        state, _ = synthetic.make_sample(problem, i)
        inp = utils.state_to_inputs(state)
        q = f(inp, False, rank=True)
        assert m._session is not None and q.numpy().dtype == np.float32 and q.numpy().shape == (inp[9],)
        np.testing.assert_allclose(q.numpy(), _oracle(params, inp), rtol=1e-4, atol=1e-4)
        with torch.no_grad():
            general = m(inp, False).numpy()
        assert np.array_equal(q.numpy(), general)                       # same plan (stable order), same kernels: same bits
        want = sorted(range(len(q)), key=lambda x: q[x], reverse=True)  # model_evaluator.py:110
        assert list(q.rankings) == want
        qd = f(inp, False, rank="device")                                # the device ranking kernel (default above 1,024 cuts)
        assert np.array_equal(qd.numpy(), q.numpy()) and list(qd.rankings) == want
        # float64 / int64 host arrays as get_state produces them (utils.py:35-238): converted while packing
        q64 = f(tuple(np.asarray(a, np.float64) if np.asarray(a).dtype.kind == "f" else np.asarray(a, np.int64) if hasattr(a, "shape") else a
                      for a in inp), False)
        assert np.array_equal(q64.numpy(), q.numpy())


@pytest.mark.parametrize("problem", ["setcov", "combauc", "capfac", "indset"])
def test_concrete_function_on_unsorted_coo(dev, problem):
    """Edge lists in arbitrary order: sorted by row on the host while packing (stable), then the specialised plan -- same scores as
    the sorted state up to the order of the by-variable sums, and no detour through the general path."""
    m, params = _model(81, dev)
    f = m.get_concrete_function()
    state, _ = synthetic.make_sample(problem, 5)
    inp = utils.state_to_inputs(state)
    rng = np.random.default_rng(0)
    p1, p2 = rng.permutation(inp[1].shape[1]), rng.permutation(inp[5].shape[1])
    shuffled = (inp[0], inp[1][:, p1], inp[2][p1], inp[3], inp[4], inp[5][:, p2], inp[6][p2]) + inp[7:]
    calls = []
    general = m.call
    m.call = lambda *a, **k: calls.append(1) or general(*a, **k)
    q = f(shuffled, False, rank=True)
    m.call = general
    assert not calls                                                     # answered by gcnn_infer, not by prepare + forward
    np.testing.assert_allclose(q.numpy(), _oracle(params, inp), rtol=1e-4, atol=1e-4)
    assert list(q.rankings) == sorted(range(len(q)), key=lambda x: q[x], reverse=True)


def test_score_state_edge_cases(dev):
    m, params = _model(82, dev)
    rng = np.random.default_rng(2)
    f32, i32 = np.float32, np.int32
    z2 = np.zeros((2, 0), i32)
    # duplicate (row, col) pairs and ties in the ranking
    C, V, K = 30, 20, 12
    rows = np.sort(rng.integers(0, C - 3, 200)); cols = rng.integers(0, V - 2, 200)
    cols[50:60] = cols[50]; rows[50:60] = rows[50]                      # ten copies of one entry, different coefficients
    krows = np.sort(rng.integers(0, K, 60)); kcols = rng.integers(0, V, 60)
    cut = rng.standard_normal((K, 6)).astype(f32); cut[5] = cut[2]      # two identical cuts with identical supports -> equal scores
    sel = krows == 2
    krows = np.concatenate([krows[krows != 5], np.full(sel.sum(), 5)]); kcols = np.concatenate([kcols[:len(krows) - sel.sum()], kcols[sel]])
    order = np.argsort(krows, kind="stable"); krows, kcols = krows[order], kcols[order]
    kvals = rng.standard_normal(len(krows)).astype(f32)
    kvals[krows == 5] = kvals[krows == 2]
    inp = (rng.standard_normal((C, 4)).astype(f32), np.stack([rows, cols]).astype(i32), rng.standard_normal(200).astype(f32).reshape(-1, 1),
           rng.standard_normal((V, 14)).astype(f32), cut, np.stack([krows, kcols]).astype(i32), kvals.reshape(-1, 1), C, V, K)
    q = m.score_state(inp, rank="device")
    np.testing.assert_allclose(q.numpy(), _oracle(params, inp), rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        assert np.array_equal(q.numpy(), m(inp, False).numpy())
    assert q[2] == q[5] and list(q.rankings) == sorted(range(K), key=lambda x: q[x], reverse=True)
    # NaN scores (diverged weights): the device ranking must still be a permutation of 0..n-1 (NaNs rank last, in index order)
    from gcnn_cut_selector_amd.model import GCNN
    mn = GCNN(device=dev, seed=5)
    w = mn.get_weights(); w[-1][:] = np.nan; mn.set_weights(w)           # out_2/bias = NaN -> every score NaN
    qn = mn.score_state(inp, rank="device")
    assert np.isnan(qn.numpy()).all() and sorted(qn.rankings.tolist()) == list(range(K)) and list(qn.rankings) == list(range(K))
    # an int64 index beyond int32 must not wrap into range while it is packed
    big = list(inp); big[1] = inp[1].astype(np.int64); big[1][1, 3] += 2 ** 32
    with pytest.raises(ValueError):
        m.score_state(tuple(big))
    # no cuts, no edges
    st = (rng.standard_normal((4, 4)).astype(f32), np.array([[0, 1], [1, 0]], i32), rng.standard_normal((2, 1)).astype(f32),
          rng.standard_normal((3, 14)).astype(f32), np.zeros((0, 6), f32), z2, np.zeros((0, 1), f32), 4, 3, 0)
    assert m.score_state(st, rank=True).numpy().shape == (0,)
    st = (rng.standard_normal((4, 4)).astype(f32), z2, np.zeros((0, 1), f32), rng.standard_normal((3, 14)).astype(f32),
          rng.standard_normal((5, 6)).astype(f32), z2, np.zeros((0, 1), f32), 4, 3, 5)
    np.testing.assert_allclose(m.score_state(st).numpy(), _oracle(params, st), rtol=1e-4, atol=1e-4)
    # a hub variable beyond the specialised plan's degree bound: declined, answered by the general path
    C = 3000
    rows = np.arange(C); cols = np.zeros(C, np.int64)
    hub = (rng.standard_normal((C, 4)).astype(f32), np.stack([rows, cols]).astype(i32), rng.standard_normal((C, 1)).astype(f32),
           rng.standard_normal((2, 14)).astype(f32), rng.standard_normal((3, 6)).astype(f32),
           np.array([[0, 1, 2], [0, 1, 0]], i32), rng.standard_normal((3, 1)).astype(f32), C, 2, 3)
    np.testing.assert_allclose(m.score_state(hub).numpy(), _oracle(params, hub), rtol=1e-4, atol=1e-4)
    # validation: out-of-range index (detected on the device, reported after the call), wrong widths
    bad = list(inp); bad[1] = inp[1].copy(); bad[1][1, 7] = V
    with pytest.raises(ValueError):
        m.score_state(tuple(bad))
    bad = list(inp); bad[3] = inp[3][:, :13]
    with pytest.raises(ValueError):
        m.score_state(tuple(bad))
    # the session survives an error and a size change
    np.testing.assert_allclose(m.score_state(inp).numpy(), _oracle(params, inp), rtol=1e-4, atol=1e-4)
