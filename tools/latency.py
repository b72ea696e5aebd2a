"""Single-state inference latency through GCNN.get_concrete_function() -- the SCIP plugin's call shape
(model_evaluator.py:82-111).  End to end = host arrays in, scores (and ranking) back on the host."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from gcnn_cut_selector_amd import _lib, synthetic, utils
from gcnn_cut_selector_amd.model import GCNN
dev = torch.device('cuda', 0)
m = GCNN(device=dev, seed=0)
f = m.get_concrete_function()
med = lambda xs: float(np.median(xs)) * 1e3
for prob in ["setcov", "combauc", "capfac", "indset"]:
    state, _ = synthetic.make_sample(prob, 7)
    rng = np.random.default_rng(0)
    inp = utils.state_to_inputs(state)
    for _ in range(10): f(inp, False).numpy()
    t_fast, t_rank, t_gen = [], [], []
    for _ in range(50):
        t0 = time.perf_counter(); q = f(inp, False).numpy(); t_fast.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); q = f(inp, False, rank=True); q.numpy(); t_rank.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        with torch.no_grad(): g = m(inp, False).numpy()
        t_gen.append(time.perf_counter() - t0)
    tm = {"pack": [], "enqueue": [], "wait": []}
    for _ in range(30):
        d = {}; m._session.run(inp, False, d)
        for k_ in tm: tm[k_].append(d[k_])
    with _lib.launch_profile() as prof:
        f(inp, False)
    dev_us = sum(ms for _, ms in prof.launches) * 1e3
    print(f"{prob:8s} cuts={inp[9]:4d} E={inp[1].shape[1] + inp[5].shape[1]:6d}  gcnn_infer end-to-end {med(t_fast):.3f} ms  (+ranking {med(t_rank):.3f} ms)  "
          f"general path (prepare + forward) {med(t_gen):.3f} ms   kernels: {len(prof.launches)} launches, {dev_us:.0f} us under event brackets")
    print(f"          host phases: pack {med(tm['pack']) * 1e3:.0f} us, enqueue (copies + {len(prof.launches)} launches) {med(tm['enqueue']) * 1e3:.0f} us, wait for the stream {med(tm['wait']) * 1e3:.0f} us")
    print("          " + "  ".join(f"{n}:{ms * 1e3:.1f}" for n, ms in prof.launches))
    # the same state with its edge lists in random order: sorted by row on the host while packing, then the same single call
    p1, p2 = rng.permutation(inp[1].shape[1]), rng.permutation(inp[5].shape[1])
    shuf = (inp[0], inp[1][:, p1], inp[2][p1], inp[3], inp[4], inp[5][:, p2], inp[6][p2]) + inp[7:]
    for _ in range(5): f(shuf, False).numpy()
    t_un = []
    for _ in range(30):
        t0 = time.perf_counter(); f(shuf, False).numpy(); t_un.append(time.perf_counter() - t0)
    print(f"          unsorted edge lists (host sort while packing + gcnn_infer): {med(t_un):.3f} ms")
