"""Per-launch times of one training step (HIP events inside the library): python tools/step_kernels.py [problem] [batch]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gcnn_cut_selector_amd import _lib, synthetic
from gcnn_cut_selector_amd.model import GCNN
from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
problem = sys.argv[1] if len(sys.argv) > 1 else "setcov"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
m = GCNN(device=dev, seed=0)
state, y, _ = synthetic.make_batch(problem, batch, 0)
b = m.prepare(state); t = torch.as_tensor(y).to(dev)
opt, ts = Adam(1e-4), TrainState(m)
for _ in range(5): train_step(m, b, t, opt, ts)
torch.cuda.synchronize()
acc = {}
order = []
for _ in range(30):
    with _lib.launch_profile() as prof:
        train_step(m, b, t, opt, ts)
    seen = {}
    for name, ms in prof.launches:
        k = (name, seen.get(name, 0)); seen[name] = k[1] + 1
        if k not in acc: acc[k] = []; order.append(k)
        acc[k].append(ms * 1e3)
tot = 0
for k in order:
    us = float(np.median(acc[k])); tot += us
    print(f"{k[0][:60]:60s} #{k[1]} {us:7.2f} us")
print(f"{problem} x{batch}: {len(order)} launches, {tot:.1f} us under event brackets; edges {b.n_edges}")
