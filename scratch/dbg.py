import sys; sys.path.insert(0,'.')
import numpy as np, torch
from gcnn_cut_selector_amd import synthetic
from gcnn_cut_selector_amd.model import GCNN
from oracle import gcnn_oracle as O
dev=torch.device('cuda',0)
params = O.randomize_params(O.init_params(7, np.float32), 8)
m = GCNN(device=dev); m.set_weights([params[n] for n in O.PARAM_NAMES])
state, y, _ = synthetic.make_batch("setcov", 2, scale=0.1)
pred = m(state, True)
loss = ((pred - torch.as_tensor(y, device=dev)) ** 2).mean(); loss.backward()
_, wl, want = O.loss_and_grads({k: v.astype(np.float64) for k, v in params.items()}, state, y, torch.float64)
names = [n for n, _, t in O.PARAM_SPEC if t]
for name, g in zip(names, m.gradients()):
    g = g.cpu().numpy().astype(np.float64); w = want[name]
    print(f"{name:32s} err={np.abs(g-w).max():.3e} ref={np.abs(w).max():.3e}")
