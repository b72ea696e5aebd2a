import sys, os, time; sys.path.insert(0,'.')
import numpy as np, torch
from gcnn_cut_selector_amd import synthetic
from gcnn_cut_selector_amd.model import GCNN
from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
dev=torch.device('cuda',0)
for prob,bs,scale in [("setcov",32,1.0),("setcov",2,0.1)]:
    m=GCNN(device=dev, seed=0)
    state,y,_=synthetic.make_batch(prob,bs,scale=scale)
    b=m.prepare(state); t=torch.as_tensor(y).to(dev)
    opt,ts=Adam(1e-4),TrainState(m)
    for _ in range(30): train_step(m,b,t,opt,ts)
    torch.cuda.synchronize()
    n=300
    t0=time.perf_counter()
    for _ in range(n): train_step(m,b,t,opt,ts)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(prob,bs,scale,"edges",b.n_edges,"cpu issue ms/step",(t1-t0)/n*1e3,"total ms/step",(t2-t0)/n*1e3, "streams", os.environ.get("GCNN_STREAMS","1"))
