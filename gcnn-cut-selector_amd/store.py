"""Device-resident sample store with on-device batch collation.

The reference re-reads its training set from disk for every batch: `tf.data` maps `utils.load_batch` over file names
(/root/reference/model_trainer.py:115-125,146-153), i.e. gunzip + unpickle + NumPy concatenation under the GIL per batch
(utils.py:339-426).  At ~3.5 ms of zlib per setcov sample that is ~200x the time this build needs for the training step
on the same batch.  An MI355X has 288 GB of HBM: a 10,000-sample setcov training set is ~8 GB in the layout below, so
the store decodes every file ONCE, keeps all samples in HBM -- features, targets and both CSR orders of both edge sets,
with sample-local index values -- and forms a mini-batch with ONE kernel launch (gcnn_collate): a disjoint union
(utils.py:401-407) of already-sorted samples is already sorted, so no per-batch sort, validation or host copy remains.
Epochs that draw samples with replacement (model_trainer.py:147) just pass the drawn ids."""

from __future__ import annotations

import ctypes as C
from collections import namedtuple

import numpy as np
import torch

from . import _lib, utils
from .graph import BipartiteGraph, _stream
from .model import Batch

# what `SampleStore.batch` returns; `trainer.process` / `trainer.pretrain` accept it in place of a load_batch 11-tuple
StoreBatch = namedtuple("StoreBatch", "batch n_cons n_vars n_cuts improvements")

_K_CONS, _K_VAR, _K_CUT, _K_E1, _K_E2 = range(5)   # unit kinds: which offset table indexes an array
_GRAPH_FIELDS = ("l_ptr", "l_oth", "l_coef", "v_ptr", "v_oth", "v_coef")


def _localise(graph: BipartiteGraph, n_left, n_var, n_edge, dev):
    """CSR arrays of a chunk built as one disjoint union -> sample-local values (what gcnn_collate re-shifts)."""
    as_dev = lambda a: torch.from_numpy(np.asarray(a, np.int64)).to(dev)
    n_left, n_var, n_edge = as_dev(n_left), as_dev(n_var), as_dev(n_edge)
    first = lambda n: torch.cumsum(n, 0) - n
    of_left = torch.repeat_interleave(first(n_edge), n_left)    # edge offset of the sample each left row belongs to
    of_var = torch.repeat_interleave(first(n_edge), n_var)
    edge_l = torch.repeat_interleave(first(n_left), n_edge)
    edge_v = torch.repeat_interleave(first(n_var), n_edge)
    i32 = torch.int32
    return dict(l_ptr=(graph.l_ptr[:-1] - of_left).to(i32), l_oth=(graph.l_oth - edge_v).to(i32), l_coef=graph.l_coef,
                v_ptr=(graph.v_ptr[:-1] - of_var).to(i32), v_oth=(graph.v_oth - edge_l).to(i32), v_coef=graph.v_coef)


class SampleStore:
    """All samples of a data set resident on one GPU; `batch(ids)` collates a mini-batch on the device.

    Build with `from_files` (the reference's sample_*.pkl files, data_collector.py:135-140) or `from_samples`
    ((state, improvements) pairs as `utils.load_sample` returns them).  Edge lists are validated once, at ingestion."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._parts = {k: [] for k in ("cons_feats", "var_feats", "cut_feats", "improvements")}
        self._gparts = [{f: [] for f in _GRAPH_FIELDS} for _ in range(2)]
        self._sizes = [[] for _ in range(5)]
        self._maxdeg = [[[], []] for _ in range(2)]    # per edge set: longest by-left / by-variable segment of every sample
        self._final = False
        self._ring, self._ring_pos = [], 0
        self._jobs = None                              # the collate jobs (sources, kinds, widths), built by the first batch

    # ---- ingestion ---------------------------------------------------------------------------------------------
    @classmethod
    def from_samples(cls, samples, device=None, chunk=64):
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        store = cls(device)
        samples = list(samples)
        for i in range(0, len(samples), chunk):
            store._add_chunk(samples[i:i + chunk])
        return store._finalise()

    @classmethod
    def from_files(cls, files, device=None, chunk=64, workers=8, process_group=None):
        """Decode every file once and move the samples to the device.  The gunzip work runs in `workers` child processes that are
        fresh, torch-free interpreters (`utils.decode_files`; `workers=0` decodes in-process); the calling script needs no
        `if __name__ == "__main__":` guard.

        Data parallel (`process_group` given; every rank passes the SAME file list): rank r decodes only the r-th contiguous share
        of the files -- 1/N of the zlib work and of the host memory per rank instead of N processes re-decoding everything on one
        host -- and the ranks then exchange their device arrays ONCE (one padded all-gather per array over RCCL / xGMI), so every
        rank ends up holding the whole store, array for array identical to a single-process ingestion: `batches(ids, b, rank,
        world)` keeps drawing any sample on any rank, as the reference's epoch sampling with replacement needs
        (model_trainer.py:147), with shards balanced by edge count."""
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        store = cls(device)
        files = list(files)
        rank, world = 0, 1
        if process_group is not None:
            import torch.distributed as dist
            rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        lo, hi = (len(files) * rank) // world, (len(files) * (rank + 1)) // world
        pending = []
        for sample in utils.decode_files(files[lo:hi], workers):
            pending.append(sample)
            if len(pending) == chunk:
                store._add_chunk(pending)
                pending = []
        if pending:
            store._add_chunk(pending)
        store._finalise()
        if world > 1:
            store._exchange(process_group)
        return store

    def _exchange(self, process_group):
        """After a sharded ingestion: every rank contributes its samples, every rank receives all of them, in rank order (= file
        order).  All stored index values are sample-local, so concatenation is all it takes."""
        from .parallel import allgather_concat
        ag = lambda t: allgather_concat(t, process_group)
        self.cons_feats, self.var_feats, self.cut_feats, self.improvements = (ag(t) for t in (
            self.cons_feats, self.var_feats, self.cut_feats, self.improvements))
        self.graphs = [{f: ag(g[f]) for f in _GRAPH_FIELDS} for g in self.graphs]
        meta = torch.from_numpy(np.concatenate([self.sizes, np.stack([m for per_set in self.max_deg for m in per_set])]).T.copy())  # [n, 9]
        meta = ag(meta.to(self.device)).cpu().numpy().T
        self.sizes = np.ascontiguousarray(meta[:5])
        self.max_deg = [[meta[5], meta[6]], [meta[7], meta[8]]]
        self.offsets = np.concatenate([np.zeros((5, 1), np.int64), np.cumsum(self.sizes, axis=1)], axis=1)
        self._jobs = None

    def _add_chunk(self, samples):
        if self._final:
            raise RuntimeError("the store is already finalised")
        dev = self.device
        c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts, imp = utils.collate(samples)
        n_e1 = np.asarray([s[0][1]["indices"].shape[1] for s in samples], np.int64)
        n_e2 = np.asarray([s[0][4]["indices"].shape[1] for s in samples], np.int64)
        if len(imp) != int(n_cuts.sum()):
            raise ValueError("one improvement per candidate cut expected (data_collector.py:135)")
        for name, t, f in (("cons_feats", c, 4), ("var_feats", v, 14), ("cut_feats", k, 6)):
            if t.ndim != 2 or t.shape[1] != f:
                raise ValueError(f"{name} must have shape [N,{f}], got {tuple(t.shape)}")
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        g1 = BipartiteGraph(up(cei), up(cef), c.shape[0], v.shape[0], validate=True, sync_max_degree=True)   # raises on out-of-range ids
        g2 = BipartiteGraph(up(kei), up(kef), k.shape[0], v.shape[0], validate=True, sync_max_degree=True)
        for name, a in (("cons_feats", c), ("var_feats", v), ("cut_feats", k), ("improvements", imp)):
            self._parts[name].append(up(a))
        for slot, (g, nl, ne) in enumerate(((g1, n_cons, n_e1), (g2, n_cuts, n_e2))):
            for f, t in _localise(g, nl, n_vars, ne, dev).items():
                self._gparts[slot][f].append(t)
            for side, (ptr, n) in enumerate(((g.l_ptr, nl), (g.v_ptr, n_vars))):   # longest segment per sample and order
                deg = (ptr[1:] - ptr[:-1]).to(torch.int64)
                sid = torch.repeat_interleave(torch.arange(len(samples), device=dev), torch.from_numpy(np.asarray(n, np.int64)).to(dev))
                self._maxdeg[slot][side].append(torch.zeros(len(samples), dtype=torch.int64, device=dev)
                                                .scatter_reduce_(0, sid, deg, "amax").cpu().numpy())
        for kind, n in enumerate((n_cons, n_vars, n_cuts, n_e1, n_e2)):
            self._sizes[kind].append(np.asarray(n, np.int64))

    def _finalise(self):
        dev = self.device
        cat = lambda parts, like: torch.cat(parts) if parts else like
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        self.cons_feats = cat(self._parts["cons_feats"], f32(0, 4))
        self.var_feats = cat(self._parts["var_feats"], f32(0, 14))
        self.cut_feats = cat(self._parts["cut_feats"], f32(0, 6))
        self.improvements = cat(self._parts["improvements"], f32(0))
        self.graphs = [{f: cat(g[f], f32(0) if f.endswith("coef") else i32(0)) for f in _GRAPH_FIELDS}
                       for g in self._gparts]
        self.sizes = np.stack([np.concatenate(s) if s else np.zeros(0, np.int64) for s in self._sizes])   # [5, n]
        self.offsets = np.concatenate([np.zeros((5, 1), np.int64), np.cumsum(self.sizes, axis=1)], axis=1)
        self.max_deg = [[np.concatenate(m) if m else np.zeros(0, np.int64) for m in per_set] for per_set in self._maxdeg]
        self._parts = self._gparts = self._sizes = self._maxdeg = None
        self._final = True
        return self

    def __len__(self):
        return self.sizes.shape[1]

    @property
    def nbytes(self):
        ts = [self.cons_feats, self.var_feats, self.cut_feats, self.improvements] + [t for g in self.graphs for t in g.values()]
        return sum(t.numel() * t.element_size() for t in ts)

    # ---- collation ---------------------------------------------------------------------------------------------
    def _table(self, ids):
        """[5][B] store offsets + [5][B+1] batch offsets, through a small ring of pinned buffers (the copy is async)."""
        b = len(ids)
        dst = np.zeros((5, b + 1), np.int64)
        np.cumsum(self.sizes[:, ids], axis=1, out=dst[:, 1:])
        n = 5 * b + 5 * (b + 1)
        if not self._ring or self._ring[0][0].numel() < n:
            self._ring = [[torch.empty(max(n, 1024), dtype=torch.int64).pin_memory(), None] for _ in range(4)]
        slot = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        if slot[1] is not None:
            slot[1].synchronize()       # the copy that last used this slot has left it (4 batches ago)
        host = slot[0].numpy()
        host[:5 * b] = self.offsets[:, ids].reshape(-1)
        host[5 * b:n] = dst.reshape(-1)
        tab = torch.empty(n, dtype=torch.int64, device=self.device)
        tab.copy_(slot[0][:n], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        return tab, dst

    def batch(self, ids) -> StoreBatch:
        """Mini-batch of the samples `ids` (any order, repeats allowed), equal array for array to
        `GCNN.prepare(utils.collate([samples[i] for i in ids]))`.  No ids: an empty batch."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if len(ids) == 0:   # a data-parallel rank may draw no sample of a short last batch: an empty batch still takes part
            return self._empty_batch()
        if ids.min() < 0 or ids.max() >= len(self):
            raise IndexError("sample id out of range")
        dev, b = self.device, len(ids)
        tab, dst = self._table(ids)
        n_c, n_v, n_k, n_e1, n_e2 = (int(x) for x in dst[:, -1])
        if max(n_c, n_v, n_k, n_e1, n_e2) >= 2 ** 31 - 1:
            raise ValueError("batch too large for int32 indices")
        # one allocation carved into the 16 arrays of the batch (64-word aligned)
        spec = [("cons_feats", n_c * 4), ("var_feats", n_v * 14), ("cut_feats", n_k * 6), ("improvements", n_k)]
        for slot, (nl, ne) in enumerate(((n_c, n_e1), (n_k, n_e2))):
            spec += [(f"{slot}.l_ptr", nl + 1), (f"{slot}.l_oth", ne), (f"{slot}.l_coef", ne), (f"{slot}.v_ptr", n_v + 1),
                     (f"{slot}.v_oth", ne), (f"{slot}.v_coef", ne)]
        pos, total = {}, 0
        for name, n in spec:
            pos[name] = (total, n)
            total += (n + 63) & ~63
        buf = torch.empty(max(total, 64), dtype=torch.int32, device=dev)
        view = lambda name: buf[pos[name][0]:pos[name][0] + pos[name][1]]
        # the 16 copy jobs: source, kinds and widths never change (built once); only the destinations move with the batch
        if self._jobs is None:
            names, jobs = [], (_lib.CollateJob * 16)()

            def job(src, name, kind, width, add=-1, is_ptr=0):
                jobs[len(names)] = _lib.CollateJob(src.data_ptr() if src.numel() else 0, 0, kind, width, add, is_ptr)
                names.append(name)

            job(self.cons_feats, "cons_feats", _K_CONS, 4)
            job(self.var_feats, "var_feats", _K_VAR, 14)
            job(self.cut_feats, "cut_feats", _K_CUT, 6)
            job(self.improvements, "improvements", _K_CUT, 1)
            for slot, (kl, ke) in enumerate(((_K_CONS, _K_E1), (_K_CUT, _K_E2))):
                g = self.graphs[slot]
                job(g["l_ptr"], f"{slot}.l_ptr", kl, 1, ke, 1)
                job(g["l_oth"], f"{slot}.l_oth", ke, 1, _K_VAR)
                job(g["l_coef"], f"{slot}.l_coef", ke, 1)
                job(g["v_ptr"], f"{slot}.v_ptr", _K_VAR, 1, ke, 1)
                job(g["v_oth"], f"{slot}.v_oth", ke, 1, kl)
                job(g["v_coef"], f"{slot}.v_coef", ke, 1)
            self._jobs = (jobs, names)
        jobs, names = self._jobs
        base = buf.data_ptr()
        for i, name in enumerate(names):
            jobs[i].dst = base + 4 * pos[name][0]
        nj = len(names)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().gcnn_collate(jobs, nj, C.c_void_p(tab.data_ptr()), C.c_void_p(tab.data_ptr() + 8 * 5 * b),
                                               b, max(n for _, n in spec), _stream(dev)), "gcnn_collate")
        tab.record_stream(torch.cuda.current_stream(dev))
        f32 = lambda name, *shape: view(name).view(torch.float32).view(*shape)
        graphs = []
        for slot, nl in enumerate((n_c, n_k)):
            graphs.append(BipartiteGraph.from_plan(
                nl, n_v, view(f"{slot}.l_ptr"), view(f"{slot}.l_oth"), f32(f"{slot}.l_coef", -1), view(f"{slot}.v_ptr"),
                view(f"{slot}.v_oth"), f32(f"{slot}.v_coef", -1),
                l_max_deg=int(self.max_deg[slot][0][ids].max()), v_max_deg=int(self.max_deg[slot][1][ids].max())))
        batch = Batch(f32("cons_feats", n_c, 4), f32("var_feats", n_v, 14), f32("cut_feats", n_k, 6), graphs[0], graphs[1])
        sizes = self.sizes[:, ids]
        return StoreBatch(batch, sizes[_K_CONS].astype(np.int32), sizes[_K_VAR].astype(np.int32),
                          sizes[_K_CUT].astype(np.int32), f32("improvements", n_k))

    def _empty_batch(self) -> StoreBatch:
        dev = self.device
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        graph = lambda: BipartiteGraph.from_plan(0, 0, i32(1), i32(0), f32(0), i32(1), i32(0), f32(0))
        z = np.zeros(0, np.int32)
        return StoreBatch(Batch(f32(0, 4), f32(0, 14), f32(0, 6), graph(), graph()), z, z, z, f32(0))

    def batches(self, ids, batch_size, rank=0, world_size=1):
        """Counterpart of `Dataset.from_tensor_slices(files).batch(batch_size).map(load_batch)`
        (model_trainer.py:115-125,150-153): consecutive groups of `batch_size` ids, the last one possibly short.
        Data parallel (`world_size` > 1; every rank holds the store and passes the SAME ids): each global batch is split
        by edge count (`parallel.shard_samples`) and this rank's share is collated -- possibly an empty batch, which still
        has to go through `train_step` so that the all-reduce matches up."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        for i in range(0, len(ids), batch_size):
            group = ids[i:i + batch_size]
            if world_size > 1:
                from .parallel import shard_samples
                edges = self.sizes[_K_E1, group] + self.sizes[_K_E2, group]
                group = group[shard_samples(edges, world_size)[rank]]
            yield self.batch(group)
