import sys, os, ctypes as C; sys.path.insert(0,'.')
os.environ['GCNN_LIB']=os.path.abspath('scratch/ab/lib_stamps.so'); os.environ['GCNN_STREAMS']='0'
import numpy as np, torch
from gcnn_cut_selector_amd import _lib, synthetic
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
names=["EMBc","EMBv","EMBk","UPDc","UPDv","UPDk","B0","KCH","VCH","CCH","CFIN","VFIN"]
for i,n in enumerate(names):
    r=d[i]
    st=[int(r[3+s]) for s in range(7) if r[3+s]!=0]+[int(r[10])]
    print(f"{n:5s} total={r[10]-r[0]:6d} par={r[1]-r[0]:5d} wstage={r[2]-r[1]:5d} stages=", [st[k+1]-st[k] for k in range(len(st)-1)])
