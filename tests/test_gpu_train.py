"""Training-path parity (GPU): fused train step, optimizer, reference-style process()/pretrain(), and size-independent
properties at the BASELINE size (setcov-500 x 32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gcnn_cut_selector_amd import synthetic, utils  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _model(seed, dev):
    from gcnn_cut_selector_amd.model import GCNN
    params = O.randomize_params(O.init_params(seed, np.float32), seed + 1)
    m = GCNN(device=dev)
    m.set_weights([params[n] for n in O.PARAM_NAMES])
    return m, params


def test_train_step_matches_oracle_and_autograd_path(dev):
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
    m, params = _model(20, dev)
    state, y, _ = synthetic.make_batch("combauc", 3)
    batch = m.prepare(state)
    yt = torch.as_tensor(y).to(dev)
    ts = TrainState(m)
    loss, scores = train_step(m, batch, yt, None, ts)
    # the same numbers through torch autograd (the GradientTape-like path)
    pred = m(batch, True)
    l2 = ((pred - yt) ** 2).mean()
    m.flat_parameters.grad = None
    l2.backward()
    assert torch.equal(pred.detach(), scores)
    np.testing.assert_allclose(float(loss), float(l2.detach()), rtol=1e-6)
    np.testing.assert_allclose(ts.grads.cpu().numpy(), m.flat_parameters.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # one optimizer step vs the oracle's Keras-form Adam on the oracle's gradients
    _, _, grads = O.loss_and_grads({k: v.astype(np.float64) for k, v in params.items()}, state, y, torch.float64)
    opt = Adam(learning_rate=lambda: 1e-3)
    train_step(m, batch, yt, opt, ts)
    for name, w in zip(O.PARAM_NAMES, m.get_weights()):
        if name in grads:
            want, _, _ = O.keras_adam_step(params[name].astype(np.float64), grads[name], 0 * grads[name], 0 * grads[name], 1, 1e-3)
            # first Adam step moves every weight by ~lr*sign(g): compare the update, tolerate sign flips of ~0 gradients
            upd, wupd = w - params[name], want - params[name]
            big = np.abs(grads[name]) > 1e-3 * np.abs(grads[name]).max()
            np.testing.assert_allclose(upd[big], wupd[big], rtol=2e-3, atol=2e-6, err_msg=name)
        else:
            np.testing.assert_array_equal(w, params[name])


def test_data_parallel_loss_scaling_semantics(dev):
    """train_step's DP branch back-propagates the local SUM; scaled by 1/count it must equal the single-GPU gradient."""
    from gcnn_cut_selector_amd.trainer import TrainState, mse_loss
    m, _ = _model(21, dev)
    state, y, _ = synthetic.make_batch("indset", 2)
    batch = m.prepare(state)
    yt = torch.as_tensor(y).to(dev)
    scores = m(batch, False)
    _, d_mean = mse_loss(scores, yt)
    _, d_sum = mse_loss(scores, yt, 1.0)
    np.testing.assert_allclose((d_sum / batch.dims.n_cuts).cpu().numpy(), d_mean.cpu().numpy(), rtol=1e-6)


def test_process_and_pretrain_mirror_reference_flow(dev, tmp_path):
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.trainer import Adam, pretrain, process
    files = []
    for i in range(6):
        state, imp = synthetic.make_sample("setcov", 100 + i, scale=0.1)
        f = str(tmp_path / f"sample_{i}.pkl"); utils.save_sample(f, state, imp); files.append(f)
    loader = [utils.load_batch(files[i:i + 2]) for i in range(0, 6, 2)]
    m = GCNN(device=dev, seed=5)
    assert pretrain(m, loader) == 11
    # oracle pretraining on the same batches, same initial weights
    p0 = dict(zip(O.PARAM_NAMES, GCNN(device=dev, seed=5).get_weights()))
    batches = [b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum())) for b in loader]
    fitted, _ = O.pretrain({k: v.astype(np.float64) for k, v in p0.items()}, batches, torch.float64)
    for shift, scale, _ in O.PRENORM_LAYERS:
        for name in (shift, scale):
            if name:
                np.testing.assert_allclose(m.get_variable(name).cpu().numpy(), fitted[name], rtol=1e-4, atol=1e-6, err_msg=name)
    fractions = np.array([0.25, 0.5, 0.75, 1])
    loss0, acc0 = process(m, loader, fractions)
    opt = Adam(learning_rate=lambda: 1e-3)
    for _ in range(15):
        train_loss, _ = process(m, loader, fractions, lambda y_true, y_pred: None, opt)   # the reference's positional order
    loss1, acc1 = process(m, loader, fractions)
    assert np.isfinite(loss0) and loss1 < loss0 and acc1.shape == (4,)
    # validation loss = cut-weighted mean of per-batch MSE (model_trainer.py:304,313), checked against the oracle
    params = dict(zip(O.PARAM_NAMES, m.get_weights()))
    tot = cnt = 0.0
    for b, st in zip(loader, batches):
        pred = O.scores(params, st, torch.float32)
        tot += float(((pred - b[10]) ** 2).sum()); cnt += len(pred)
    np.testing.assert_allclose(loss1, tot / cnt, rtol=1e-3)


def test_concrete_function_and_inference_mode(dev):
    m, params = _model(22, dev)
    state, _, _ = synthetic.make_batch("setcov", 1, scale=0.3)
    f = m.get_concrete_function()
    q = f(state, False).numpy()                     # the SCIP plugin's call shape (model_evaluator.py:103)
    assert q.dtype == np.float32 and q.shape == (state[9],)
    with torch.enable_grad():
        t = m(state, True)                          # training-mode forward (saves activations)
    np.testing.assert_allclose(q, t.numpy(), rtol=1e-5, atol=1e-6)   # same arithmetic, possibly another summation order per segment
    ranks = sorted(range(len(q)), key=lambda i: q[i], reverse=True)
    assert len(set(ranks)) == len(q)


# ---- BASELINE-size properties (setcov-500 x 32: ~1M edges) -----------------------------------------------------------
def test_full_size_batching_invariance_determinism_and_finite_grads(dev):
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    m, _ = _model(23, dev)
    samples = [synthetic.make_sample("setcov", i) for i in range(32)]
    full = synthetic.stack_samples(samples)
    state = full[:7] + (int(full[7].sum()), int(full[8].sum()), int(full[9].sum()))
    batch = m.prepare(state)
    a = m(batch, False).numpy()
    assert np.array_equal(a, m(batch, False).numpy())                     # deterministic (no float atomics)
    nk = full[9]
    first = synthetic.stack_samples(samples[:1]); last = synthetic.stack_samples(samples[-1:])
    s0 = m(first[:7] + (int(first[7][0]), int(first[8][0]), int(first[9][0])), False).numpy()
    s1 = m(last[:7] + (int(last[7][0]), int(last[8][0]), int(last[9][0])), False).numpy()
    np.testing.assert_allclose(a[:nk[0]], s0, rtol=1e-5, atol=1e-5)       # disjoint-union batching invariance
    np.testing.assert_allclose(a[-nk[-1]:], s1, rtol=1e-5, atol=1e-5)
    ts = TrainState(m)
    yt = torch.as_tensor(full[10]).to(dev)
    l1, _ = train_step(m, batch, yt, None, ts); g1 = ts.grads.clone()
    l2, _ = train_step(m, batch, yt, None, ts)
    assert torch.equal(g1, ts.grads) and torch.equal(l1, l2)              # bitwise reproducible gradients
    assert bool(torch.isfinite(g1).all())
    # linearity of the backward pass in d_scores: grads(2*d) == 2*grads(d)
    ws = m._take_workspace(batch)
    flat = m.flat_parameters.detach()
    scores = m._forward_into(flat, batch, ws)
    d = torch.randn_like(scores)
    ga, gb = torch.zeros_like(flat), torch.zeros_like(flat)   # non-trainable / padding slots are never written
    m._backward_into(flat, batch, ws, d, ga)
    m._forward_into(flat, batch, ws)
    m._backward_into(flat, batch, ws, 2 * d, gb)
    np.testing.assert_allclose(gb.cpu().numpy(), 2 * ga.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(ga.abs().max()))


def test_graphed_step_matches_eager_steps(dev):
    """A captured hipGraph of the whole step (3 streams, device-resident Adam state) replayed == the same steps issued
    eagerly: bitwise, because the kernels, their order of summation and the device-side Adam arithmetic are identical."""
    from gcnn_cut_selector_amd.trainer import Adam, GraphedTrainStep, TrainState, train_step
    state, y, _ = synthetic.make_batch("combauc", 4)
    finals = []
    for graphed in (False, True):
        m, _ = _model(30, dev)
        batch = m.prepare(state)
        yt = torch.as_tensor(y).to(dev)
        opt, ts = Adam(1e-3), TrainState(m)
        if graphed:
            step = GraphedTrainStep(m, batch, yt, opt, ts, warmup=2)     # two eager warm-up steps, then capture (no step)
            for _ in range(3):
                loss, _ = step()
        else:
            for _ in range(5):
                loss, _ = train_step(m, batch, yt, opt, ts, device_optimizer=True)
        torch.cuda.synchronize()
        opt.sync_from_device()
        assert opt.iterations == 5
        finals.append((float(loss), np.concatenate([w.reshape(-1) for w in m.get_weights()])))
    assert finals[0][0] == finals[1][0]
    assert np.array_equal(finals[0][1], finals[1][1])


def test_device_resident_adam_matches_host_parameterised_adam(dev):
    from gcnn_cut_selector_amd.trainer import Adam
    m1, _ = _model(31, dev)
    m2, _ = _model(31, dev)
    g = torch.randn_like(m1.flat_parameters.detach()) * m1._trainable_mask
    o1, o2 = Adam(3e-4), Adam(3e-4)
    for _ in range(4):
        o1.apply_flat(m1, g)
        o2.apply_flat_dev(m2, g)
    np.testing.assert_allclose(m1.flat_parameters.detach().cpu().numpy(), m2.flat_parameters.detach().cpu().numpy(),
                               rtol=2e-6, atol=1e-8)


def test_device_ranking_metric_matches_reference_semantics(dev):
    """gcnn_ranking_metric vs the oracle's restatement of model_trainer.py:288-301, with many ties and ragged sample sizes."""
    from gcnn_cut_selector_amd.trainer import ranking_metric
    rng = np.random.default_rng(0)
    n_cuts = np.array([1, 2, 3, 17, 64, 65, 100, 255, 256, 257, 1000, 4096, 5])   # <= 256: rank by counting; above: the sorting network
    pred = [rng.integers(0, 6, n).astype(np.float32) for n in n_cuts]
    true = [p.copy() for p in pred]
    for p, t in zip(pred, true):             # perturb a suffix so prefixes of varying length agree
        k = int(rng.integers(0, len(p) + 1))
        t[k:] = rng.integers(0, 6, len(p) - k)
    fractions = np.array([0.25, 0.5, 0.75, 1.0], np.float32)
    acc = torch.zeros(4, device=dev)
    loss, loss_acc = torch.tensor([0.5], device=dev), torch.zeros(1, device=dev)
    frac = ranking_metric(torch.from_numpy(np.concatenate(pred)).to(dev), torch.from_numpy(np.concatenate(true)).to(dev),
                          n_cuts, torch.from_numpy(fractions).to(dev), acc, loss, loss_acc)
    want = np.array([O.ranking_fraction(p, t) for p, t in zip(pred, true)])
    np.testing.assert_allclose(frac.cpu().numpy(), want, rtol=1e-6)
    np.testing.assert_array_equal(acc.cpu().numpy(), (want[:, None] >= fractions[None, :]).sum(0))
    assert float(loss_acc) == 0.5 * n_cuts.sum()


def test_dp_step_world1_equals_single_gpu_step(dev):
    """The data-parallel branch (SUM loss, count slot stored by backward, Adam dividing by the reduced count) with the
    collective replaced by the identity (world size 1) must update the weights exactly like the single-GPU branch."""
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, mse_loss
    state, y, _ = synthetic.make_batch("indset", 2)
    outs = []
    for dp in (False, True):
        m, _ = _model(40, dev)
        batch = m.prepare(state)
        yt = torch.as_tensor(y).to(dev)
        opt, ts = Adam(1e-3), TrainState(m)
        flat = m.flat_parameters.detach()
        ws = m._take_workspace(batch)
        scores = m._forward_into(flat, batch, ws)
        if dp:
            _, d = mse_loss(scores, yt, 1.0)
            m._backward_into(flat, batch, ws, d, ts.grads, count_slot=ts.count)
            assert float(ts.count) == batch.dims.n_cuts
            opt.apply_flat(m, ts.grads, grad_scale=ts.count, divide=True)
        else:
            _, d = mse_loss(scores, yt)
            m._backward_into(flat, batch, ws, d, ts.grads)
            opt.apply_flat(m, ts.grads)
        outs.append(m.flat_parameters.detach().cpu().numpy().copy())
    # the two branches scale the same gradient at different points (1/n inside the loss head vs a division in Adam): equal up to
    # rounding, which Adam's first step g / (|g| + eps) magnifies for gradient entries around eps = 1e-7 -- hence the absolute
    # term of one thousandth of a step (lr = 1e-3)
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=1e-6)


def test_train_step_on_degenerate_batches(dev):
    """Fused loss head + fused Adam on batches that leave the fused fast paths: no cuts at all (every gradient zero, Adam still
    decays its moments), no constraints (the reduction does not cover every gradient, so Adam runs as its own launch), no
    edges.  The step must equal the unfused sequence forward -> mse -> backward -> Adam."""
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, mse_loss, train_step
    rng = np.random.default_rng(5)
    z2 = np.zeros((2, 0), np.int32)
    ei = lambda n_left, n_var, n: np.stack([np.sort(rng.integers(0, n_left, n)), rng.integers(0, n_var, n)]).astype(np.int32)
    f32 = np.float32
    cases = {
        "no cuts": (rng.standard_normal((4, 4)).astype(f32), ei(4, 3, 6), rng.standard_normal((6, 1)).astype(f32),
                    rng.standard_normal((3, 14)).astype(f32), np.zeros((0, 6), f32), z2, np.zeros((0, 1), f32), 4, 3, 0),
        "no constraints": (np.zeros((0, 4), f32), z2, np.zeros((0, 1), f32), rng.standard_normal((3, 14)).astype(f32),
                           rng.standard_normal((5, 6)).astype(f32), ei(5, 3, 7), rng.standard_normal((7, 1)).astype(f32), 0, 3, 5),
        "no edges": (rng.standard_normal((4, 4)).astype(f32), z2, np.zeros((0, 1), f32), rng.standard_normal((3, 14)).astype(f32),
                     rng.standard_normal((5, 6)).astype(f32), z2, np.zeros((0, 1), f32), 4, 3, 5),
    }
    for name, state in cases.items():
        y = torch.as_tensor(rng.uniform(0, 0.1, state[9]).astype(f32)).to(dev)
        outs = []
        for fused in (True, False):
            m, _ = _model(60, dev)
            batch = m.prepare(state)
            opt, ts = Adam(1e-3), TrainState(m)
            for _ in range(2):   # second step: the moments are non-zero
                if fused:
                    loss, _ = train_step(m, batch, y, opt, ts)
                else:
                    flat = m.flat_parameters.detach()
                    ws = m._take_workspace(batch)
                    scores = m._forward_into(flat, batch, ws)
                    loss, d = mse_loss(scores, y)
                    m._backward_into(flat, batch, ws, d, ts.grads)
                    m._give_workspace(ws)
                    opt.apply_flat(m, ts.grads)
            outs.append((float(loss), ts.grads.cpu().numpy().copy(), m.flat_parameters.detach().cpu().numpy().copy()))
        (l0, g0, w0), (l1, g1, w1) = outs
        np.testing.assert_allclose(l0, l1, rtol=1e-5, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(g0, g1, rtol=1e-5, atol=1e-8, err_msg=name)
        np.testing.assert_allclose(w0, w1, rtol=1e-6, atol=1e-8, err_msg=name)
        assert np.isfinite(w0).all(), name


def test_tester_process_matches_reference_semantics(dev, tmp_path):
    """tester.process vs model_tester.py:173-237 restated with the oracle: cut-weighted MSE and the MEAN ranking fraction."""
    from gcnn_cut_selector_amd import tester
    from gcnn_cut_selector_amd.store import SampleStore
    m, params = _model(70, dev)
    samples = [synthetic.make_sample("combauc", 300 + i) for i in range(5)] + [synthetic.make_sample("setcov", 300, scale=0.2)]
    files = []
    for i, (state, imp) in enumerate(samples):
        f = str(tmp_path / f"sample_{i}.pkl"); utils.save_sample(f, state, imp); files.append(f)
    loader = [utils.load_batch(files[i:i + 4]) for i in range(0, len(files), 4)]     # a full and a short batch
    loss, mean_acc = tester.process(m, loader)
    tot = cnt = fr = 0.0
    for b in loader:
        st = b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum()))
        pred = O.scores(params, st, torch.float32)
        tot += float(((pred - b[10]) ** 2).sum()); cnt += len(pred)
        start = 0
        for nk in b[9]:
            fr += O.ranking_fraction(pred[start:start + nk], b[10][start:start + nk]); start += nk
    np.testing.assert_allclose(loss, tot / cnt, rtol=1e-4)
    np.testing.assert_allclose(mean_acc, fr / len(samples), rtol=1e-6)
    # the same through a device-resident store
    store = SampleStore.from_samples(samples, dev)
    loss2, acc2 = tester.process(m, store.batches(np.arange(len(samples)), 4))
    np.testing.assert_allclose([loss2, acc2], [loss, mean_acc], rtol=1e-6)


def test_adam_with_zero_global_cut_count_is_no_step(dev):
    """Data parallel: Adam divides by the all-reduced cut count; a global batch without cuts must leave weights, moments and
    the device-side step counter untouched (not 0 * inf = NaN)."""
    from gcnn_cut_selector_amd.trainer import Adam
    m, _ = _model(71, dev)
    before = m.flat_parameters.detach().clone()
    g = torch.zeros_like(before)
    zero = torch.zeros(1, device=dev)
    opt = Adam(1e-3)
    opt.apply_flat(m, torch.randn_like(before) * m._trainable_mask)          # a real step first: moments are non-zero
    w1, m1, v1 = m.flat_parameters.detach().clone(), opt.m.clone(), opt.v.clone()
    opt.apply_flat(m, g, grad_scale=zero, divide=True)
    opt.apply_flat_dev(m, g, grad_scale=zero, divide=True)
    torch.cuda.synchronize()
    assert torch.equal(m.flat_parameters.detach(), w1) and torch.equal(opt.m, m1) and torch.equal(opt.v, v1)
    assert bool(torch.isfinite(m.flat_parameters.detach()).all())
    t_dev = float(opt._dev[4])
    opt.apply_flat_dev(m, g, grad_scale=torch.ones(1, device=dev), divide=True)
    assert float(opt._dev[4]) == t_dev + 1


_BITS_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from gcnn_cut_selector_amd import synthetic, utils
from gcnn_cut_selector_amd.model import GCNN
from gcnn_cut_selector_amd.trainer import TrainState, train_step
from oracle import gcnn_oracle as O
dev = torch.device("cuda", 0)
params = O.randomize_params(O.init_params(31, np.float32), 32)
m = GCNN(device=dev); m.set_weights([params[n] for n in O.PARAM_NAMES])
state, y, _ = synthetic.make_batch("setcov", 6)
batch = m.prepare(state); ts = TrainState(m)
loss, scores = train_step(m, batch, torch.as_tensor(y).to(dev), None, ts)
h = hashlib.sha256()
h.update(ts.grads.cpu().numpy().tobytes()); h.update(scores.cpu().numpy().tobytes()); h.update(np.float32(float(loss)).tobytes())
s1, _ = synthetic.make_sample("combauc", 2)
h.update(m.score_state(utils.state_to_inputs(s1), rank=True).numpy().tobytes())
print("BITS", h.hexdigest())
"""


def test_few_tile_programs_give_the_same_bits_as_the_one_wave_programs(dev):
    """k_rows_split.hpp (four waves per tile; launches of <= 256 tiles: the training turnaround, single-state inference) keeps
    the MFMA order per output element, so gradients, scores and loss must not change by one bit when it is switched off
    (GCNN_SPLIT_MAX_TILES=0, read once per process -> two child processes)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for knob in ("256", "0"):
        env = dict(os.environ, GCNN_SPLIT_MAX_TILES=knob)
        r = subprocess.run([sys.executable, "-c", _BITS_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("BITS")][-1])
    assert out[0] == out[1]


def test_fused_adam_steps_are_bitwise_reproducible_and_equal_the_unfused_update(dev):
    """The blocks that unfold the folded layers' gradients (fold_block, k_wgrad.hpp) ride in the launch whose Adam updates
    overwrite Wf, W1a and bf: they must read the forward pass's copies of those parameters.  Three fused steps from the same
    state, twice: the same bits; and one fused step equals gradients first, Adam afterwards (model_trainer.py:271-276)."""
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
    state, y, _ = synthetic.make_batch("setcov", 8)
    outs = []
    for rep in range(3):
        m, _ = _model(31, dev)
        batch = m.prepare(state)
        yt = torch.as_tensor(y).to(dev)
        opt, ts = Adam(learning_rate=lambda: 1e-3), TrainState(m)
        if rep < 2:
            for _ in range(3):
                train_step(m, batch, yt, opt, ts)
            outs.append(m.flat_parameters.detach().cpu().numpy().copy())
        else:   # one step: fused against gradients-then-Adam on a second copy of the model
            m2, _ = _model(31, dev)
            opt2, ts2 = Adam(learning_rate=lambda: 1e-3), TrainState(m2)
            train_step(m, batch, yt, opt, ts)
            train_step(m2, m2.prepare(state), yt, None, ts2)
            opt2.apply_flat(m2, ts2.grads)
            np.testing.assert_allclose(m.flat_parameters.detach().cpu().numpy(), m2.flat_parameters.detach().cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert np.array_equal(outs[0], outs[1])
