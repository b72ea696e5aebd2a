import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from gcnn_cut_selector_amd import synthetic, utils
from gcnn_cut_selector_amd.model import GCNN
dev=torch.device('cuda',0)
m=GCNN(device=dev, seed=0)
f=m.get_concrete_function()
for prob in ["setcov","combauc","capfac","indset"]:
    state,_=synthetic.make_sample(prob, 7)
    inp=utils.state_to_inputs(state)
    for _ in range(5): f(inp, False).numpy()
    ts=[]; tp=[]; tf=[]
    for _ in range(30):
        t0=time.perf_counter(); b=m.prepare(inp); torch.cuda.synchronize(); t1=time.perf_counter()
        q=f(b, False); q=q.numpy(); t2=time.perf_counter()
        tp.append(t1-t0); tf.append(t2-t1)
        t0=time.perf_counter(); q=f(inp, False).numpy(); ts.append(time.perf_counter()-t0)
    print(f"{prob:8s} cuts={inp[9]:4d} E={inp[1].shape[1]+inp[5].shape[1]:6d} end-to-end {np.median(ts)*1e3:.3f} ms  (prepare {np.median(tp)*1e3:.3f} ms, forward+D2H {np.median(tf)*1e3:.3f} ms)")
