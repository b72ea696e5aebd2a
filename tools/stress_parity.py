"""Randomised parity sweep (GPU): random bipartite states of random sizes -- isolated nodes, duplicate entries, unsorted COO,
one-row sets, cut counts around the 16-row tile size and the four-waves-per-tile threshold -- forward, inference and backward
against the fp64 oracle.  Scores: rtol = atol = 1e-4.  Gradients: 1e-4 of each tensor's largest entry, like the test suite; tensors
beyond that are reported and accepted under the single-column rule of tests/test_gpu_stress.py -- on random data about one case
in twenty has a ReLU pre-activation so close to zero that THIS fp32 evaluation takes the other branch than fp64 (and than
torch's fp32 evaluation), which moves one column of the weight gradients of that unit's layer by one row's share and every
tensor of the layers before it over all columns (seed 2, case 5: a unit of the constraint embedding; seed 21, case 17: a unit
of var_conv_out_1).
python tools/stress_parity.py [cases] [seed] [only: evaluate this case alone and print every gradient tensor's error]"""
import os, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)
from test_gpu_model import _model  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
m, params = _model(11, dev)
p64 = {k: v.astype(np.float64) for k, v in params.items()}
worst = 0.0
for case in range(cases):
    C = int(rng.choice([1, 2, 15, 16, 17, 100, 1000, 5000]) if rng.random() < .5 else rng.integers(1, 6000))
    V = int(rng.choice([1, 3, 16, 33, 500, 4097]) if rng.random() < .5 else rng.integers(1, 9000))
    K = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 255, 256, 257, 4095, 4096, 4097, 4200]) if rng.random() < .6 else rng.integers(1, 5000))
    e1 = int(rng.integers(0, 20 * max(C, V))); e2 = int(rng.integers(0, 12 * max(K, 8)))
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    # skewed degrees: a few hub rows / hub variables
    def ends(n, e):
        a = rng.integers(0, n, e)
        if e and rng.random() < .5: a[: e // 3] = rng.integers(0, max(1, n // 50 + 1), e // 3)
        return a
    cei = np.stack([ends(C, e1), ends(V, e1)]).astype(np.int32)
    kei = np.stack([ends(K, e2), ends(V, e2)]).astype(np.int32)
    if rng.random() < .5:   # (row, col)-sorted like get_state, else arbitrary order
        o = np.lexsort((cei[1], cei[0])); cei = cei[:, o]
        o = np.lexsort((kei[1], kei[0])); kei = kei[:, o]
    state = (f(C, 4), cei, f(e1, 1), f(V, 14), f(K, 6), kei, f(e2, 1), C, V, K)
    print(f"case {case:3d}  C={C:5d} V={V:5d} K={K:5d} E1={e1:6d} E2={e2:6d}", end="  ", flush=True)
    y = rng.uniform(0, 0.2, K)
    if only is not None and case != only:
        print("(skipped)")
        continue
    want = O.scores(p64, state, torch.float64)
    with torch.no_grad():
        got = m(state, False).numpy()
    err = float(np.abs(got - want).max()) if K else 0.0
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4), (case, C, V, K, e1, e2, err)
    pred = m(state, True)
    loss = ((pred - torch.as_tensor(y, device=pred.device)) ** 2).mean()
    m.flat_parameters.grad = None
    loss.backward()
    _, want_loss, wg = O.loss_and_grads(p64, state, y, torch.float64)
    _, _, wg32 = O.loss_and_grads(params, state, y, torch.float32)   # how far ANY fp32 evaluation sits from fp64 (cancellation in d w_edge)
    assert abs(float(loss.detach()) - want_loss) <= 1e-4 * max(1.0, abs(want_loss)), (case, "loss")
    flips, wide = [], []
    for name, g in zip([n for n, _, t in O.PARAM_SPEC if t], m.gradients()):
        g = g.cpu().numpy().astype(np.float64)
        ref = max(np.abs(wg[name]).max(), 1e-6)
        rel, gap32 = np.abs(g - wg[name]).max() / ref, np.abs(wg32[name].astype(np.float64) - wg[name]).max() / ref
        if only is not None:
            e_ = (np.abs(g - wg[name]) / ref).reshape(-1, g.shape[-1]).max(0)
            print(f"\n   {name:32s} rel {rel:.2e}  fp32-oracle gap {gap32:.2e}  worst columns {np.round(np.sort(e_)[-4:], 6)} at {np.argsort(e_)[-4:]}", end="")
            continue
        if rel > max(1e-4, 3 * gap32):
            # A flipped unit moves ONE output column of the weight gradients of ITS layer; the layers before it (in forward order)
            # inherit the difference spread over all columns.  So: somewhere there must be a tensor whose error is confined to one
            # column, and nothing may exceed 5e-3 (seed 21, case 17: unit (1851, 40) of var_conv_out_1 has the fp64 pre-activation
            # -2.3e-4 against summands of magnitude 265, i.e. 8.7e-7 relative -- a few fp32 roundings; every tensor before it is off
            # by 2-14e-4, everything behind it agrees to 1e-7).
            e = np.abs(g - wg[name]) / ref
            cols = e.reshape(-1, e.shape[-1]).max(0)
            one = cols.size > 1 and np.sort(cols)[-2] <= max(1e-4, 3 * gap32)
            wide.append((name, rel, one, int(cols.argmax())))
            flips.append(f"{name} {rel:.1e}" + (f" (one column: {int(cols.argmax())})" if one else ""))
    if wide:
        assert max(w[1] for w in wide) <= 5e-3, (case, "gradient error beyond a flipped unit's reach", wide)
        assert any(w[2] for w in wide) or max(w[1] for w in wide) <= 5e-4, (case, "wide gradient error without a single-column origin", wide)
    # the fused training step (forward + MSE head + cut-row turnaround in one launch, backward from there) against the autograd path
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    batch = m.prepare(state)
    ts = TrainState(m)
    loss2, scores2 = train_step(m, batch, torch.as_tensor(y, dtype=torch.float32).to(dev), None, ts)
    ga, gf = m.flat_parameters.grad.cpu().numpy(), ts.grads.cpu().numpy()
    assert np.allclose(gf, ga, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(ga).max()))), (case, "fused step vs autograd", float(np.abs(gf - ga).max()))
    assert abs(float(loss2) - want_loss) <= 1e-4 * max(1.0, abs(want_loss)), (case, "fused loss")
    q = m.score_state(state, rank=True)   # the one-call inference path (falls back to the general path for unsorted lists)
    assert np.allclose(q.numpy(), want, rtol=1e-4, atol=1e-4), (case, "score_state")
    assert list(q.rankings) == sorted(range(K), key=lambda i: q[i], reverse=True), (case, "ranking")
    worst = max(worst, err)
    print(f"max|score err| {err:.2e}" + ("   gradients beyond 1e-4 (ReLU branch): " + ", ".join(flips) if flips else ""), flush=True)
print(f"{cases} cases ok, worst score error {worst:.2e}")
