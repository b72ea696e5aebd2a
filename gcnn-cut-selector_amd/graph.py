"""Device-side index structures for one bipartite edge set.

The reference feeds `tf.gather` / `tf.scatter_nd` a raw COO list (/root/reference/model.py:564-569).  Here the COO
list is turned ONCE per batch into receiver-sorted CSR in both orders (by left node and by variable), so that the
scatter-sum and every gradient of the gathers run as atomic-free segmented sums (gcnn_graph_build in
include/gcnn_hip.h).  The two orders are independent lists: no pass needs to find an edge of one order in the other."""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


class _MaxDegreeMailbox:
    """Longest segment of a freshly built plan without a host wait: the two maxima are reduced on the device, copied into a
    pinned slot behind the build, and picked up the first time the graph is used AFTER the copy has landed (until then the plan
    says "unknown", which only means the long-segment finder launch always runs).  A small ring of slots; a slot taken over by a
    newer graph before its owner looked simply leaves that owner at "unknown"."""
    SLOTS = 32

    def __init__(self):
        self.host, self.events, self.gen, self.pos = None, [None] * self.SLOTS, [0] * self.SLOTS, 0

    def post(self, values_dev):
        if self.host is None:
            self.host = torch.zeros(self.SLOTS, 2, dtype=torch.int32).pin_memory()
        i = self.pos
        self.pos = (i + 1) % self.SLOTS
        self.gen[i] += 1
        if self.events[i] is not None:
            self.events[i].synchronize()      # an older copy into this slot (32 graphs ago) must have landed before the next one starts
        self.host[i].copy_(values_dev, non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record(torch.cuda.current_stream(values_dev.device))
        return i, self.gen[i]

    def poll(self, ticket):
        """(l_max_deg, v_max_deg) once the copy is done, None while it is in flight, False if the slot was taken over."""
        i, gen = ticket
        if self.gen[i] != gen:
            return False
        if not self.events[i].query():
            return None
        return tuple(int(x) for x in self.host[i].tolist())


_mailbox = _MaxDegreeMailbox()


class BipartiteGraph:
    """One edge set (constraint<->variable or cut<->variable) in by-left and by-variable CSR form, on the GPU.

    edge_inds: [2,E] int32, row 0 = left (constraint/cut) id, row 1 = variable id (utils.py:110,234).
    edge_feats: [E,1] or [E] fp32 raw coefficients.  Any edge order is accepted; ties keep input order."""

    def __init__(self, edge_inds: torch.Tensor, edge_feats: torch.Tensor, n_left: int, n_var: int, validate=True,
                 keep_perm=False, sync_max_degree=False):
        if edge_inds.dim() != 2 or edge_inds.shape[0] != 2:
            raise ValueError(f"edge index tensor must have shape [2,E], got {tuple(edge_inds.shape)}")
        if edge_inds.dtype != torch.int32:
            raise ValueError(f"edge index tensor must be int32, got {edge_inds.dtype}")
        n_edges = edge_inds.shape[1]
        edge_feats = edge_feats.reshape(-1)
        if edge_feats.numel() != n_edges or edge_feats.dtype != torch.float32:
            raise ValueError("edge features must be fp32 with one value per edge")
        if not edge_inds.is_cuda:
            raise ValueError("BipartiteGraph expects device tensors (GCNN.prepare moves host inputs)")
        dev = edge_inds.device
        edge_inds = edge_inds.contiguous()
        edge_feats = edge_feats.contiguous()
        lib = _lib.lib()
        left_sorted = 0
        if validate and n_edges > 0:   # one kernel + one 8-byte read: range check and "already sorted by left id?"
            flags = torch.empty(2, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.gcnn_graph_check(_ptr(edge_inds), n_edges, n_left, n_var, _ptr(flags), _stream(dev)),
                           "gcnn_graph_check")
            bad, unsorted = flags.tolist()
            if bad:
                raise ValueError(f"edge index out of range (left ids must be in [0,{n_left}), variable ids in [0,{n_var}))")
            left_sorted = int(not unsorted)
        self.n_edges, self.n_left, self.n_var, self.device = n_edges, int(n_left), int(n_var), dev
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.l_ptr = torch.empty(n_left + 1, **i32)
        self.v_ptr = torch.empty(n_var + 1, **i32)
        self.l_oth = torch.empty(n_edges, **i32)
        self.v_oth = torch.empty(n_edges, **i32)
        self.l_coef = torch.empty(n_edges, **f32)
        self.v_coef = torch.empty(n_edges, **f32)
        self.l_perm = torch.empty(n_edges, **i32) if keep_perm else None
        temp_bytes = lib.gcnn_graph_temp_bytes(n_edges)
        temp = torch.empty(temp_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.gcnn_graph_build(_ptr(edge_inds), _ptr(edge_feats), n_edges, n_left, n_var, left_sorted,
                                            _ptr(self.l_ptr), _ptr(self.l_oth), _ptr(self.l_coef), _ptr(self.v_ptr),
                                            _ptr(self.v_oth), _ptr(self.v_coef), _ptr(self.l_perm), _ptr(temp),
                                            temp_bytes, _stream(dev)), "gcnn_graph_build")
        # keep the temp alive until the stream has consumed it
        temp.record_stream(torch.cuda.current_stream(dev))
        # longest segment of either order: lets the edge passes skip their long-segment launch when the list has no segment
        # that is long for its mean degree.  Not waited for (`sync_max_degree=False`, the default: GCNN.prepare stays free of host
        # synchronisation beyond the optional validation read): the values arrive through a pinned mailbox and are adopted by
        # the first use after the copy has landed; 0 = unknown until then.
        self.l_max_deg = self.v_max_deg = 0
        self._md_ticket = None
        if n_edges > 0:
            md = torch.stack([(self.l_ptr[1:] - self.l_ptr[:-1]).max() if n_left else self.l_ptr.new_zeros(()),
                              (self.v_ptr[1:] - self.v_ptr[:-1]).max() if n_var else self.v_ptr.new_zeros(())])
            if sync_max_degree:
                self.l_max_deg, self.v_max_deg = (int(x) for x in md.tolist())
            else:
                self._md_ticket = _mailbox.post(md)
        self._bind()

    @classmethod
    def from_plan(cls, n_left, n_var, l_ptr, l_oth, l_coef, v_ptr, v_oth, v_coef, l_max_deg=0, v_max_deg=0):
        """Wrap CSR arrays that already exist on the device (SampleStore.batch collates them; no sort runs).
        l_max_deg / v_max_deg: longest segment of each order if known (0 = unknown)."""
        g = cls.__new__(cls)
        g.n_edges, g.n_left, g.n_var, g.device = int(l_oth.numel()), int(n_left), int(n_var), l_ptr.device
        g.l_ptr, g.l_oth, g.l_coef, g.v_ptr, g.v_oth, g.v_coef = l_ptr, l_oth, l_coef, v_ptr, v_oth, v_coef
        g.l_perm = None
        g.l_max_deg, g.v_max_deg = int(l_max_deg), int(v_max_deg)
        g._md_ticket = None
        g._bind()
        return g

    @property
    def c(self):
        """The `gcnn_graph` struct the C ABI takes; adopts the longest-segment values once their copy has landed."""
        if self._md_ticket is not None:
            got = _mailbox.poll(self._md_ticket)
            if got is not None:
                self._md_ticket = None
                if got:
                    self.l_max_deg, self.v_max_deg = got
                    self._bind()
        return self._c

    def _bind(self):
        n_edges = self.n_edges
        self._c = _lib.Graph(self.l_ptr.data_ptr(), self.l_oth.data_ptr() if n_edges else 0,
                            self.l_coef.data_ptr() if n_edges else 0, self.v_ptr.data_ptr(),
                            self.v_oth.data_ptr() if n_edges else 0, self.v_coef.data_ptr() if n_edges else 0,
                            self.l_max_deg, self.v_max_deg)
