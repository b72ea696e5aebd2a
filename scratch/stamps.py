import sys, os, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
from gcnn_cut_selector_amd import _lib
_lib.LIB_PATH = os.path.abspath('scratch/libgcnn_stamps.so')
from gcnn_cut_selector_amd import synthetic
from gcnn_cut_selector_amd.model import GCNN
from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
dev=torch.device('cuda',0)
m=GCNN(device=dev, seed=0)
state,y,_=synthetic.make_batch("setcov",32)
b=m.prepare(state); t=torch.as_tensor(y).to(dev)
opt,ts=Adam(1e-4),TrainState(m)
for _ in range(5): train_step(m,b,t,opt,ts)
dbg=torch.zeros(16*16,dtype=torch.int64,device=dev)
lib=_lib.lib(); lib.gcnn_debug_set_stamps.argtypes=[C.c_void_p]; lib.gcnn_debug_set_stamps(C.c_void_p(dbg.data_ptr()))
train_step(m,b,t,opt,ts); torch.cuda.synchronize()
d=dbg.cpu().numpy().reshape(16,16)
names=["EMBc","EMBv","EMBk","UPDc","UPDv","UPDk","B0","KCH","VCH","CCH","VFIN","CFIN"]
for i,n in enumerate(names):
    r=d[i]; t0=r[0]
    segs=[]
    segs.append(f"wstage={r[1]-r[0]}")
    for s in range(6):
        if r[2+2*s]==0: break
        nxt = r[4+2*s] if (s<5 and r[4+2*s]!=0) else r[14]
        segs.append(f"s{s}: gemm={r[3+2*s]-r[2+2*s]} rowpass={nxt-r[3+2*s]}")
    print(n, "total", r[14]-r[0], " | ".join(segs))
