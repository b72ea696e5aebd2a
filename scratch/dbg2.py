import sys; sys.path.insert(0,'.')
import numpy as np, torch
from gcnn_cut_selector_amd import ops
from gcnn_cut_selector_amd.graph import BipartiteGraph
dev=torch.device('cuda',0)
g=torch.Generator().manual_seed(1)
for (n_left,n_var,n_edges) in [(3,50,2000),(40,30,300)]:
    left=torch.randint(0,n_left,(n_edges,),generator=g); var=torch.randint(0,n_var,(n_edges,),generator=g)
    order=torch.argsort(left*n_var+var,stable=True); left,var=left[order],var[order]
    ei=torch.stack([left,var]).to(torch.int32); coef=torch.randn(n_edges,generator=g,dtype=torch.float64)
    graph=BipartiteGraph(ei.to(dev),coef.float().to(dev),n_left,n_var)
    pl,pr,w=torch.randn(n_left,64,generator=g,dtype=torch.float64),torch.randn(n_var,64,generator=g,dtype=torch.float64),torch.randn(64,generator=g,dtype=torch.float64)
    f=lambda t:t.float().to(dev)
    one=torch.ones(1,device=dev); zero=torch.zeros(1,device=dev)
    s,mask=ops.conv_edge_fwd(graph,True,f(pl),f(pr),f(w),zero,one,one,want_mask=True)
    torch.cuda.synchronize()
    j=pl.float()[ei[0].long()]+coef.float()[:,None]*w.float()[None,:]+pr.float()[ei[1].long()]
    bits=(j>0)
    m=mask.cpu().numpy().astype(np.uint64)
    got=np.zeros((n_edges,64),bool)
    for k in range(4):
        for c in range(16):
            got[:,4*c+k]=((m>>np.uint64(16*k+c))&np.uint64(1)).astype(bool)
    bad=(got!=bits.numpy())
    print(n_left,n_var,n_edges,"bad bits",bad.sum(),"bad edges",bad.any(1).sum(), "first bad edges", np.flatnonzero(bad.any(1))[:20])
    if bad.any():
        e=np.flatnonzero(bad.any(1))[0]; print(" edge",e,"bad channels",np.flatnonzero(bad[e])[:16], "seg", int(left[e]), "segstart", int((left<left[e]).sum()))
