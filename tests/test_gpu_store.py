"""Device-resident sample store (GPU): `SampleStore.batch` must reproduce, array for array and bit for bit, what the
host path (`utils.collate` -> `GCNN.prepare`) builds for the same samples -- the reference's `utils.load_batch`
(utils.py:339-426) followed by the CSR plan -- and training through it must give the same numbers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gcnn_cut_selector_amd import synthetic, utils  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)

GRAPH_FIELDS = ("l_ptr", "l_oth", "l_coef", "v_ptr", "v_oth", "v_coef")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _samples():
    out = [synthetic.make_sample("setcov", i, scale=0.1) for i in range(5)]
    out += [synthetic.make_sample("combauc", i) for i in range(4)]
    out += [synthetic.make_sample("indset", i, scale=0.05) for i in range(3)]
    # a sample without candidate cuts and one without any edges: empty segments in the middle of a batch
    (cons, cons_edge, var, cut, cut_edge), imp = synthetic.make_sample("setcov", 7, scale=0.1)
    no_cut = dict(cut, values=cut["values"][:0])
    no_cut_edge = dict(cut_edge, indices=cut_edge["indices"][:, :0], values=cut_edge["values"][:0])
    out.insert(3, ((cons, cons_edge, var, no_cut, no_cut_edge), imp[:0]))
    no_edge = dict(cons_edge, indices=cons_edge["indices"][:, :0], values=cons_edge["values"][:0])
    out.insert(8, ((cons, no_edge, var, no_cut, no_cut_edge), imp[:0]))
    return out


def _model(seed, dev):
    from gcnn_cut_selector_amd.model import GCNN
    params = O.randomize_params(O.init_params(seed, np.float32), seed + 1)
    m = GCNN(device=dev)
    m.set_weights([params[n] for n in O.PARAM_NAMES])
    return m


def _assert_same_batch(got, want):
    for name in ("cons_feats", "var_feats", "cut_feats"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    for g in ("cons_graph", "cut_graph"):
        a, b = getattr(got, g), getattr(want, g)
        assert (a.n_edges, a.n_left, a.n_var) == (b.n_edges, b.n_left, b.n_var)
        for f in GRAPH_FIELDS:
            assert torch.equal(getattr(a, f), getattr(b, f)), f"{g}.{f}"
    assert [getattr(got.dims, f) for f, _ in got.dims._fields_] == [getattr(want.dims, f) for f, _ in want.dims._fields_]


@pytest.mark.parametrize("ids", [[0, 1, 2, 3, 4, 5], [13, 3, 8, 3, 0, 11, 11, 6], [3], [8, 3], [5, 8, 9, 3, 12, 1, 7, 2, 10, 4, 6, 0, 11, 13]])
def test_store_batch_equals_host_collate_plus_prepare(dev, ids):
    from gcnn_cut_selector_amd.store import SampleStore
    samples = _samples()
    m = _model(3, dev)
    store = SampleStore.from_samples(samples, dev, chunk=4)       # several ingestion chunks
    assert len(store) == len(samples) and store.nbytes > 0
    sb = store.batch(ids)
    host = utils.collate([samples[i] for i in ids])
    want = m.prepare(tuple(host[:7]) + (int(host[7].sum()), int(host[8].sum()), int(host[9].sum())))
    _assert_same_batch(sb.batch, want)
    np.testing.assert_array_equal(sb.n_cons, host[7])
    np.testing.assert_array_equal(sb.n_vars, host[8])
    np.testing.assert_array_equal(sb.n_cuts, host[9])
    np.testing.assert_array_equal(sb.improvements.cpu().numpy(), host[10])
    if sb.batch.dims.n_cuts:
        with torch.no_grad():
            assert torch.equal(m(sb.batch, False), m(want, False))


def test_store_from_files_and_process_match_the_host_loader(dev, tmp_path):
    from gcnn_cut_selector_amd.store import SampleStore
    from gcnn_cut_selector_amd.trainer import Adam, pretrain, process
    samples = _samples()
    files = []
    for i, (state, imp) in enumerate(samples):
        files.append(str(tmp_path / f"sample_{i}.pkl"))
        utils.save_sample(files[-1], state, imp)
    store = SampleStore.from_files(files, dev, chunk=5, workers=2)
    rng = np.random.default_rng(0)
    ids = rng.choice(len(files), 24, replace=True)            # an epoch drawn with replacement (model_trainer.py:147)
    fractions = np.array([0.25, 0.5, 0.75, 1.0])
    results = []
    for use_store in (False, True):
        m = _model(5, dev)
        if use_store:
            loader = lambda idx, bs: list(store.batches(idx, bs))
        else:
            loader = lambda idx, bs: [utils.load_batch([files[i] for i in idx[j:j + bs]]) for j in range(0, len(idx), bs)]
        n_layers = pretrain(m, loader(np.arange(len(files)), 4))
        opt = Adam(learning_rate=lambda: 1e-3)
        train = process(m, loader(ids, 6), fractions, None, opt)
        valid = process(m, loader(np.arange(len(files)), 5), fractions)
        results.append((n_layers, train, valid, m.flat_parameters.detach().cpu().numpy().copy()))
    (n0, t0, v0, w0), (n1, t1, v1, w1) = results
    assert n0 == n1
    np.testing.assert_array_equal(w0, w1)              # same batches, same kernels: bitwise equal weights
    assert t0[0] == t1[0] and v0[0] == v1[0]
    np.testing.assert_array_equal(t0[1], t1[1])
    np.testing.assert_array_equal(v0[1], v1[1])


def test_store_rejects_bad_ids_and_bad_samples(dev):
    from gcnn_cut_selector_amd.store import SampleStore
    samples = _samples()[:3]
    store = SampleStore.from_samples(samples, dev)
    with pytest.raises(IndexError):
        store.batch([0, 3])
    empty = store.batch([])          # legal: the share of a data-parallel rank that drew no sample
    assert empty.batch.dims.n_cuts == 0 and empty.batch.n_edges == 0 and empty.improvements.numel() == 0
    (cons, cons_edge, var, cut, cut_edge), imp = samples[0]
    bad = dict(cons_edge, indices=cons_edge["indices"].copy())
    bad["indices"][1, 0] = var["values"].shape[0]              # variable id out of range
    with pytest.raises(ValueError):
        SampleStore.from_samples([((cons, bad, var, cut, cut_edge), imp)], dev)


def test_full_size_store_epoch_runs_and_matches_prepared_batch(dev):
    """BASELINE size (setcov-500 x 32): the store's batch equals the host-built one and a step on it gives the same loss."""
    from gcnn_cut_selector_amd.store import SampleStore
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    samples = [synthetic.make_sample("setcov", i) for i in range(32)]
    store = SampleStore.from_samples(samples, dev, chunk=16)
    m = _model(9, dev)
    sb = store.batch(np.arange(32))
    state, y, _ = synthetic.make_batch("setcov", 32)
    want = m.prepare(state)
    _assert_same_batch(sb.batch, want)
    np.testing.assert_array_equal(sb.improvements.cpu().numpy(), np.asarray(y, np.float32))
    ts = TrainState(m)
    l0, s0 = train_step(m, want, torch.as_tensor(y).to(dev), None, ts)
    g0 = ts.grads.clone()
    l1, s1 = train_step(m, sb.batch, sb.improvements, None, ts)
    assert float(l0) == float(l1) and torch.equal(s0, s1) and torch.equal(g0, ts.grads)
