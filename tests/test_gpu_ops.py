"""Per-kernel parity (GPU): each C-ABI op against a plain fp64 torch/NumPy computation of the same op.
Tolerances: fp32 kernels vs fp64 reference, rtol 2e-5 on O(1..100) sums unless stated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 3e-5, 3e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=what)


def _rand(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float64)


@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 128, 129, 1000, 4097])
def test_linear_fwd_plain(dev, n):
    from gcnn_cut_selector_amd import ops
    g = torch.Generator().manual_seed(n)
    x, w, b = _rand(g, n, 64), _rand(g, 64, 64), _rand(g, 64)
    # asymmetric weights on purpose: a swapped row/col map in the MFMA layout cannot pass
    y = ops.linear_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), relu=True)
    _close(y, torch.relu(x @ w + b))
    y = ops.linear_fwd(x.float().to(dev), w.float().to(dev))
    _close(y, x @ w)


def test_linear_fwd_identity_layout(dev):
    """A = I rows against an asymmetric W: out row r must equal W row r (catches transposed C/D maps)."""
    from gcnn_cut_selector_amd import ops
    w = torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64)
    x = torch.eye(64)
    y = ops.linear_fwd(x.to(dev), w.to(dev))
    assert torch.equal(y.cpu(), w)


def test_linear_fwd_full(dev):
    from gcnn_cut_selector_amd import ops
    g = torch.Generator().manual_seed(7)
    n = 777
    xa, xb, wa, wb, b, bd = _rand(g, n, 64), _rand(g, n, 64), _rand(g, 64, 64), _rand(g, 64, 64), _rand(g, 64), _rand(g, 64)
    deg = torch.randint(0, 9, (n,), generator=g)
    seg = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).to(torch.int32)
    sa = torch.tensor([0.37], dtype=torch.float64)
    f = lambda t: t.float().to(dev)
    y = ops.linear_fwd(f(xa), f(wa), f(b), relu=True, xb=f(xb), wb=f(wb), sa=f(sa), bd=f(bd), seg_ptr=seg.to(dev))
    want = torch.relu((xa * sa) @ wa + xb @ wb + b + deg.double()[:, None] * bd)
    _close(y, want)


@pytest.mark.parametrize("n", [1, 32, 100, 1025])
def test_linear_bwd(dev, n):
    from gcnn_cut_selector_amd import ops
    g = torch.Generator().manual_seed(100 + n)
    dy, y, wa, wb = _rand(g, n, 64), _rand(g, n, 64), _rand(g, 64, 64), _rand(g, 64, 64)
    so = torch.tensor([1.7], dtype=torch.float64)
    f = lambda t: t.float().to(dev)
    dy_d = f(dy)
    prev = _rand(g, n, 64)
    dx2_d = f(prev)
    dx, dx2 = ops.linear_bwd(dy_d, f(wa), ymask=f(y), so=f(so), wb=f(wb), dx2=dx2_d, beta2=1)
    dpre = dy * (y > 0)
    _close(dy_d, dpre, what="in-place masked dy")
    _close(dx, so * (dpre @ wa.T), what="dx")
    _close(dx2, prev + dpre @ wb.T, what="dx2 accumulate")
    dx, _ = ops.linear_bwd(f(dy), f(wa))
    _close(dx, dy @ wa.T, what="dx plain")


def _random_graph(gen, n_left, n_var, n_edges, dev, sort=True, hub=False):
    from gcnn_cut_selector_amd.graph import BipartiteGraph
    left = torch.randint(0, n_left, (n_edges,), generator=gen)
    if hub and n_edges > 4:
        left[: n_edges // 2] = 0  # one very long segment
    var = torch.randint(0, n_var, (n_edges,), generator=gen)
    if sort:
        order = torch.argsort(left * n_var + var, stable=True)
        left, var = left[order], var[order]
    ei = torch.stack([left, var]).to(torch.int32)
    coef = torch.randn(n_edges, generator=gen, dtype=torch.float64)
    graph = BipartiteGraph(ei.to(dev), coef.float().to(dev).reshape(-1, 1), n_left, n_var)
    return graph, ei.long(), coef


@pytest.mark.parametrize("shape", [(5, 7, 0), (5, 7, 1), (40, 30, 300), (300, 50, 3000), (7, 900, 1500), (3, 5, 2000)])
@pytest.mark.parametrize("sort", [True, False])
def test_graph_build(dev, shape, sort):
    n_left, n_var, n_edges = shape
    g = torch.Generator().manual_seed(sum(shape))
    graph, ei, coef = _random_graph(g, n_left, n_var, n_edges, dev, sort=sort)
    for side, (ptr, oth, cf, n) in enumerate([(graph.l_ptr, graph.l_oth, graph.l_coef, n_left),
                                               (graph.v_ptr, graph.v_oth, graph.v_coef, n_var)]):
        order = torch.argsort(ei[side], stable=True)
        want_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.bincount(ei[side], minlength=n).cumsum(0)])
        assert torch.equal(ptr.cpu().long(), want_ptr)
        assert torch.equal(oth.cpu().long(), ei[1 - side][order])       # bit-exact index work, stable order
        assert torch.equal(cf.cpu(), coef.float()[order])


def test_graph_build_rejects_bad_indices(dev):
    from gcnn_cut_selector_amd.graph import BipartiteGraph
    ei = torch.tensor([[0, 5], [0, 1]], dtype=torch.int32, device=dev)
    with pytest.raises(ValueError):
        BipartiteGraph(ei, torch.zeros(2, device=dev), 5, 2)
    with pytest.raises(ValueError):
        BipartiteGraph(ei.long(), torch.zeros(2, device=dev), 6, 2)


def _edge_reference(ei, coef, pl, pr, w, esh, esc, s1, n_recv, side):
    c = (coef + esh) * esc
    j = pl[ei[0]] + c[:, None] * w[None, :] + pr[ei[1]]
    h = torch.relu(s1 * j)
    s = torch.zeros(n_recv, 64, dtype=torch.float64).index_add_(0, ei[side], h)
    return s, j, c


@pytest.mark.parametrize("shape", [(40, 30, 300), (300, 50, 6000), (700, 900, 1500), (3, 50, 2000), (64, 64, 0)])
@pytest.mark.parametrize("recv_is_left", [True, False])
@pytest.mark.parametrize("hub", [False, True])
def test_conv_edge_fwd_bwd(dev, shape, recv_is_left, hub):
    from gcnn_cut_selector_amd import ops
    n_left, n_var, n_edges = shape
    g = torch.Generator().manual_seed(sum(shape) + recv_is_left)
    graph, ei, coef = _random_graph(g, n_left, n_var, n_edges, dev, hub=hub)
    pl, pr, w = _rand(g, n_left, 64), _rand(g, n_var, 64), _rand(g, 64)
    esh, esc = torch.tensor(0.1, dtype=torch.float64), torch.tensor(1.3, dtype=torch.float64)
    s1 = torch.tensor(-0.8 if n_edges == 1500 else 0.8, dtype=torch.float64)  # one shape exercises a negative PreNorm scale
    side = 0 if recv_is_left else 1
    n_recv = n_left if recv_is_left else n_var
    f = lambda t: t.float().reshape(-1).to(dev) if t.dim() == 0 else t.float().to(dev)
    args = (f(pl), f(pr), f(w), f(esh), f(esc), f(s1))
    s, saved = ops.conv_edge_fwd(graph, recv_is_left, *args, save=True)
    # the inference variant may give a segment more lanes (small graphs: a wave or a block per segment): another summation order
    _close(ops.conv_edge_fwd(graph, recv_is_left, *args), s.double().cpu(), rtol=1e-5, atol=1e-5, what="inference variant")
    pl_, pr_, w_ = pl.clone().requires_grad_(), pr.clone().requires_grad_(), w.clone().requires_grad_()
    want, _, _ = _edge_reference(ei, coef, pl_, pr_, w_, esh, esc, s1, n_recv, side)
    _close(s, want, what="edge forward")
    ds = _rand(g, n_recv, 64)
    if n_edges:
        want.backward(ds)
        d_pl, d_pr, d_w = ops.conv_edge_bwd(graph, recv_is_left, saved, *args, f(ds))
        _close(d_pl, pl_.grad, rtol=1e-4, atol=1e-4, what="d PL")
        _close(d_pr, pr_.grad, rtol=1e-4, atol=1e-4, what="d PR")
        _close(d_w, w_.grad, rtol=1e-4, atol=1e-4, what="d w_edge")


@pytest.mark.parametrize("n_recv,n_edges", [(1, 0), (10, 5), (100, 5000), (4000, 30000), (3, 9000)])
@pytest.mark.parametrize("sorted_index", [True, False])
def test_scatter_sum_and_transpose(dev, n_recv, n_edges, sorted_index):
    from gcnn_cut_selector_amd import ops
    g = torch.Generator().manual_seed(n_recv + n_edges)
    idx = torch.randint(0, n_recv, (n_edges,), generator=g)
    if sorted_index:
        idx = idx.sort().values
    msg = _rand(g, n_edges, 64)
    m = msg.float().to(dev).requires_grad_()
    out = ops.scatter_sum(m, idx.to(dev), n_recv)
    want = torch.zeros(n_recv, 64, dtype=torch.float64).index_add_(0, idx, msg)
    _close(out, want)
    dout = _rand(g, n_recv, 64)
    out.backward(dout.float().to(dev))
    assert torch.equal(m.grad.cpu(), dout.float()[idx])  # a row gather: bit-exact


def test_scatter_sum_is_deterministic(dev):
    from gcnn_cut_selector_amd import ops
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 500, (40000,), generator=g).to(dev)
    msg = torch.randn(40000, 64, generator=g).to(dev)
    plan = ops.SegmentPlan(idx, 500)
    a = ops.scatter_sum(msg, plan, 500)
    b = ops.scatter_sum(msg, plan, 500)
    assert torch.equal(a, b)
