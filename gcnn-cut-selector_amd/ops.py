"""Thin torch-facing wrappers over the per-op C ABI (include/gcnn_hip.h).  Each wrapper only checks shapes, allocates
outputs and forwards device pointers; all arithmetic is in libgcnn_hip.so.  These are the unfused building blocks of
`PartialGraphConvolution.call` (/root/reference/model.py:533-575) and double as the unit-test surface."""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .graph import BipartiteGraph, _ptr, _stream

EMB = 64


def _chk(t, shape_tail=(EMB,), dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name}: expected contiguous {dtype} device tensor [N,{','.join(map(str, shape_tail))}], "
                         f"got {t.dtype} {tuple(t.shape)} cuda={t.is_cuda}")


def linear_fwd(xa, wa, bias=None, relu=False, xb=None, wb=None, sa=None, bd=None, seg_ptr=None):
    """y = act((sa*xa) @ wa [+ xb @ wb] [+ bias] [+ deg (x) bd]) on the fp32 MFMA (gcnn_linear_fwd)."""
    _chk(xa, name="xa"); _chk(xb, name="xb")
    y = torch.empty_like(xa)
    with torch.cuda.device(xa.device):
        _lib.check(_lib.lib().gcnn_linear_fwd(_ptr(xa), _ptr(sa), _ptr(wa), _ptr(xb), _ptr(wb), _ptr(bias), _ptr(bd),
                                              _ptr(seg_ptr), int(relu), _ptr(y), xa.shape[0], _stream(xa.device)),
                   "gcnn_linear_fwd")
    return y


def linear_bwd(dy, wa, ymask=None, so=None, dx=None, beta=0, wb=None, dx2=None, beta2=0):
    """dy <- dy*(ymask>0) in place; dx (=|+=) so*(dy @ wa^T); optionally dx2 (=|+=) dy @ wb^T (gcnn_linear_bwd)."""
    _chk(dy, name="dy"); _chk(ymask, name="ymask")
    if dx is None:
        dx = torch.empty_like(dy)
        beta = 0
    if wb is not None and dx2 is None:
        dx2 = torch.empty_like(dy)
        beta2 = 0
    with torch.cuda.device(dy.device):
        _lib.check(_lib.lib().gcnn_linear_bwd(_ptr(dy), _ptr(ymask), _ptr(wa), _ptr(so), _ptr(dx), int(beta), _ptr(wb),
                                              _ptr(dx2), int(beta2), dy.shape[0], _stream(dy.device)), "gcnn_linear_bwd")
    return dx, dx2


def _edge_side(graph: BipartiteGraph, by_left: bool):
    return (graph.l_ptr, graph.l_oth, graph.l_coef, graph.n_left) if by_left else \
           (graph.v_ptr, graph.v_oth, graph.v_coef, graph.n_var)


def conv_edge_fwd(graph: BipartiteGraph, recv_is_left: bool, pl, pr, w_edge, e_shift, e_scale, s1, save=False):
    """S[r] = sum_e relu(s1*(PL[l_e] + c_e*w + PR[v_e])) over the receiver's segment (gcnn_conv_edge_fwd).
    With save=True returns (S, N) where N[R,64] (active edges per receiver and channel) is what conv_edge_bwd consumes."""
    ptr, oth, coef, n_recv = _edge_side(graph, recv_is_left)
    p_recv, p_oth = (pl, pr) if recv_is_left else (pr, pl)
    dev = pl.device
    out = torch.empty((n_recv, EMB), dtype=torch.float32, device=dev)
    nrows = torch.empty((n_recv, EMB), dtype=torch.float32, device=dev) if save else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().gcnn_conv_edge_fwd(_ptr(ptr), _ptr(oth), _ptr(coef), n_recv, graph.n_edges, _ptr(p_recv),
                                                 _ptr(p_oth), _ptr(w_edge), _ptr(e_shift), _ptr(e_scale), _ptr(s1),
                                                 _ptr(out), _ptr(nrows), graph.l_max_deg if recv_is_left else graph.v_max_deg,
                                                 _stream(dev)), "gcnn_conv_edge_fwd")
    return (out, nrows) if save else out


def conv_edge_bwd(graph: BipartiteGraph, recv_is_left: bool, nrows, pl, pr, w_edge, e_shift, e_scale, s1, d_s):
    """Gradients of the edge pass: (d_PL, d_PR, d_w_edge[64]).  The receiver side uses the N rows of the forward; the sender
    side recomputes the ReLU pattern from the projected tables PL / PR (nothing per edge was stored)."""
    lib = _lib.lib()
    dev = d_s.device
    n_recv = graph.n_left if recv_is_left else graph.n_var
    d_recv = torch.empty((n_recv, EMB), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gcnn_conv_edge_bwd_recv(_ptr(d_s), _ptr(nrows), _ptr(s1), n_recv, _ptr(d_recv), _stream(dev)),
                   "gcnn_conv_edge_bwd_recv")
        sptr, soth, scoef, n_send = _edge_side(graph, not recv_is_left)
        p_send, p_recv = (pr, pl) if recv_is_left else (pl, pr)
        d_send = torch.empty((n_send, EMB), dtype=torch.float32, device=dev)
        rows = torch.empty((16384, EMB), dtype=torch.float32, device=dev)  # per-block partials of d w_edge (GCNN_EDGE_DW_PARTS)
        n_parts = C.c_int32(0)
        _lib.check(lib.gcnn_conv_edge_bwd_send(_ptr(sptr), _ptr(soth), _ptr(scoef), n_send, graph.n_edges, _ptr(p_send),
                                               _ptr(p_recv), _ptr(w_edge), _ptr(e_shift), _ptr(e_scale), _ptr(s1), _ptr(d_s),
                                               _ptr(d_send), _ptr(rows), C.byref(n_parts),
                                               graph.v_max_deg if recv_is_left else graph.l_max_deg, _stream(dev)),
                   "gcnn_conv_edge_bwd_send")
    d_pl, d_pr = (d_recv, d_send) if recv_is_left else (d_send, d_recv)
    return d_pl, d_pr, rows[:n_parts.value].sum(0)


class SegmentPlan:
    """Receiver-sorted view of an index vector: the plan behind the standalone scatter-sum pass."""

    def __init__(self, index: torch.Tensor, out_size: int, validate=True):
        index = index.to(torch.int32).contiguous()
        e = index.numel()
        g = BipartiteGraph(torch.stack([index, torch.zeros_like(index)]), torch.zeros(e, dtype=torch.float32,
                           device=index.device), out_size, 1, validate, keep_perm=True)
        self.seg_ptr, self.perm, self.n_recv, self.n_edges = g.l_ptr, g.l_perm, int(out_size), e
        srt = bool((index[1:] >= index[:-1]).all()) if e > 1 else True
        self.sorted = srt  # messages already receiver-sorted: stream them without the permutation


class _ScatterSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, messages, plan):
        out = torch.empty((plan.n_recv, EMB), dtype=torch.float32, device=messages.device)
        with torch.cuda.device(messages.device):
            _lib.check(_lib.lib().gcnn_seg_sum_f32(_ptr(messages), _ptr(plan.seg_ptr), None if plan.sorted else _ptr(plan.perm),
                                                   plan.n_recv, _ptr(out), _stream(messages.device)), "gcnn_seg_sum_f32")
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, d_out):
        plan = ctx.plan
        d_out = d_out.contiguous()
        d_msg = torch.empty((plan.n_edges, EMB), dtype=torch.float32, device=d_out.device)
        with torch.cuda.device(d_out.device):
            _lib.check(_lib.lib().gcnn_seg_bcast_f32(_ptr(d_out), _ptr(plan.seg_ptr), None if plan.sorted else _ptr(plan.perm),
                                                     plan.n_recv, _ptr(d_msg), _stream(d_out.device)), "gcnn_seg_bcast_f32")
        return d_msg, None


def scatter_sum(messages: torch.Tensor, index, out_size: int):
    """`tf.scatter_nd(indices=index[:,None], updates=messages, shape=[out_size,64])` (model.py:568-569): duplicates are
    summed, untouched rows are zero.  `index` may be a tensor or a prebuilt SegmentPlan.  Atomic-free and deterministic."""
    _chk(messages, name="messages")
    plan = index if isinstance(index, SegmentPlan) else SegmentPlan(index, out_size)
    if plan.n_edges != messages.shape[0]:
        raise ValueError("one index per message row expected")
    return _ScatterSum.apply(messages, plan)
