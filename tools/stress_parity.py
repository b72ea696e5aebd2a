"""Randomised parity sweep (GPU): random bipartite states of random sizes -- isolated nodes, duplicate entries, unsorted COO,
one-row sets, cut counts around the 16-row tile size and the four-waves-per-tile threshold -- forward, inference and backward
against the fp64 oracle.  Scores: rtol = atol = 1e-4.  Gradients: 1e-4 of each tensor's largest entry, like the test suite; a tensor
beyond that is accepted up to 5e-4 and reported -- on random data about one case in twenty has a ReLU pre-activation so close
to zero that THIS fp32 evaluation takes the other branch than fp64 (and than torch's fp32 evaluation), which moves one column of
one weight gradient by one row's share (seed 2, case 5: unit of the constraint embedding, error confined to one column).
python tools/stress_parity.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)
from test_gpu_model import _model  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
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
    flips = []
    for name, g in zip([n for n, _, t in O.PARAM_SPEC if t], m.gradients()):
        g = g.cpu().numpy().astype(np.float64)
        ref = max(np.abs(wg[name]).max(), 1e-6)
        rel, gap32 = np.abs(g - wg[name]).max() / ref, np.abs(wg32[name].astype(np.float64) - wg[name]).max() / ref
        if rel > max(5e-4, 3 * gap32):   # a single flipped unit moves ONE output column; anything wider is a defect
            e = np.abs(g - wg[name]) / ref
            cols = e.reshape(-1, e.shape[-1]).max(0)
            assert np.sort(cols)[-2] <= max(1e-4, 3 * gap32), (case, name, rel, gap32, np.sort(cols)[-4:])
            flips.append(f"{name} {rel:.1e} (one column: {int(cols.argmax())})")
            continue
        if rel > max(1e-4, 3 * gap32): flips.append(f"{name} {rel:.1e}")
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
