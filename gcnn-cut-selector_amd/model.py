"""`GCNN` with the call surface of the reference's model (/root/reference/model.py), running on hand-written HIP.

Surface mirrored (file:line in the reference):
  GCNN()                                   model.py:164-226   no-arg constructor, attrs emb_size/cons_feats/...
  model(inputs10, training) / model.call   model.py:257-300   -> flat fp32 scores, `.numpy()` works (model_evaluator.py:103)
  save_state / restore_state               model.py:47-67     62 consecutive pickle.dump(np.ndarray) records
  pretrain_init / pretrain / pretrain_next model.py:69-133    PreNorm fitting hooks (model_trainer.py:207-230)
  variables / trainable_variables          model.py:215, model_trainer.py:272-273
  input_signature                          model.py:218-226   kept as a description of dtypes/shapes
All arithmetic happens in libgcnn_hip.so (include/gcnn_hip.h); torch only stores tensors and hosts the autograd node.
"""

from __future__ import annotations

import ctypes as C
import os
import pickle

import numpy as np
import torch

from . import _lib, _safe_pickle
from .graph import BipartiteGraph, _ptr, _stream

EMB = 64


def _emb_spec(prefix, f):
    return [(f"{prefix}_prenorm/shift", (f,), False), (f"{prefix}_prenorm/scale", (f,), False),
            (f"{prefix}_emb_1/kernel", (f, EMB), True), (f"{prefix}_emb_1/bias", (EMB,), True),
            (f"{prefix}_emb_2/kernel", (EMB, EMB), True), (f"{prefix}_emb_2/bias", (EMB,), True)]


def _conv_spec(name):
    return [(f"{name}_feat_left/kernel", (EMB, EMB), True), (f"{name}_feat_left/bias", (EMB,), True),
            (f"{name}_feat_edge/kernel", (1, EMB), True), (f"{name}_feat_right/kernel", (EMB, EMB), True),
            (f"{name}_final_prenorm/scale", (1,), False),
            (f"{name}_feat_final/kernel", (EMB, EMB), True), (f"{name}_feat_final/bias", (EMB,), True),
            (f"{name}_post_prenorm/scale", (1,), False),
            (f"{name}_out_1/kernel", (2 * EMB, EMB), True), (f"{name}_out_1/bias", (EMB,), True),
            (f"{name}_out_2/kernel", (EMB, EMB), True), (f"{name}_out_2/bias", (EMB,), True)]


# Checkpoint order = Keras `Model.variables` of the reference (model.py:53-56, 215): tracked sub-layers in attribute
# order, each Dense [kernel, bias], each PreNormLayer [shift, scale].  Inferred from Keras 2.7 semantics; not verifiable
# here without TensorFlow (shapes make a wrong order fail loudly in restore_state).
VARIABLE_SPEC = (_emb_spec("cons", 4)
                 + [("cons_edge_prenorm/shift", (1,), False), ("cons_edge_prenorm/scale", (1,), False)]
                 + _emb_spec("var", 14) + _emb_spec("cut", 6)
                 + [("cut_edge_prenorm/shift", (1,), False), ("cut_edge_prenorm/scale", (1,), False)]
                 + _conv_spec("cons_conv") + _conv_spec("var_conv") + _conv_spec("cut_conv")
                 + [("out_1/kernel", (EMB, EMB), True), ("out_1/bias", (EMB,), True),
                    ("out_2/kernel", (EMB, 1), True), ("out_2/bias", (1,), True)])


# The 11 PreNorm layers in CALL order (model.py:287-296, 563-570): (shift variable | None, scale variable, units).
PRENORM_LAYERS = [("cons_prenorm/shift", "cons_prenorm/scale", 4), ("cons_edge_prenorm/shift", "cons_edge_prenorm/scale", 1),
                  ("var_prenorm/shift", "var_prenorm/scale", 14), ("cut_prenorm/shift", "cut_prenorm/scale", 6),
                  ("cut_edge_prenorm/shift", "cut_edge_prenorm/scale", 1),
                  (None, "cons_conv_final_prenorm/scale", 1), (None, "cons_conv_post_prenorm/scale", 1),
                  (None, "var_conv_final_prenorm/scale", 1), (None, "var_conv_post_prenorm/scale", 1),
                  (None, "cut_conv_final_prenorm/scale", 1), (None, "cut_conv_post_prenorm/scale", 1)]


class ScoreTensor(torch.Tensor):
    """Device tensor whose `.numpy()` copies to the host, so the reference's `model(...).numpy()` call sites work."""

    def numpy(self, *args, **kwargs):
        return torch.Tensor.numpy(self.detach().cpu().as_subclass(torch.Tensor), *args, **kwargs)


class ScoreArray(np.ndarray):
    """Host-side scores of the single-state inference path: an ndarray that also answers `.numpy()` (the reference's call sites do
    `get_improvements(state, False).numpy()`, model_evaluator.py:103).  `rankings` (optional): indices in descending score
    order, equal scores in index order -- `sorted(range(n), key=lambda x: quality[x], reverse=True)` of model_evaluator.py:110."""
    rankings = None

    def numpy(self):
        return np.asarray(self)


class _UseGeneralPath(Exception):
    """The specialised single-state path declined (unsorted edge list, very long segment, too many variables)."""


class _InferenceSession:
    """Host side of gcnn_infer (include/gcnn_hip.h): persistent pinned staging buffers and a device arena, one C call per state."""

    def __init__(self, model):
        self.model = model
        self.pin_in = self.pin_out = self.arena = None
        self.in_np = self.out_np = None
        self.sort_scratch = None
        self.layouts = {}

    def _layout(self, key):
        lay = self.layouts.get(key)
        if lay is None:
            dims, lay = _lib.Dims(*key), _lib.InferLayout()
            rc = _lib.lib().gcnn_infer_layout_for(C.byref(dims), C.byref(lay))
            if rc == -4:
                lay = False
            else:
                _lib.check(rc, "gcnn_infer_layout_for")
                lay = (dims, lay, list(lay.in_off), list(lay.out_off))
            if len(self.layouts) >= 256:
                self.layouts.pop(next(iter(self.layouts)))   # evict the oldest entry only
            self.layouts[key] = lay
        return lay

    def run(self, inputs, want_order, timings=None):
        """`timings` (optional dict): filled with the host-side phases in seconds (tools/latency.py)."""
        import time
        t0 = time.perf_counter()
        c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts = inputs
        c, v, k = np.asarray(c), np.asarray(v), np.asarray(k)
        cei, kei, cef, kef = np.asarray(cei), np.asarray(kei), np.asarray(cef), np.asarray(kef)
        for name, t, f in (("cons_feats", c, 4), ("var_feats", v, 14), ("cut_feats", k, 6)):
            if t.ndim != 2 or t.shape[1] != f:
                raise ValueError(f"{name} must have shape [N,{f}], got {tuple(t.shape)}")
        for name, total, t in (("n_cons", n_cons, c), ("n_vars", n_vars, v), ("n_cuts", n_cuts, k)):
            if int(total) != t.shape[0]:
                raise ValueError(f"{name}={int(total)} does not match the {t.shape[0]} feature rows")
        for name, ei, ef in (("cons_edge_inds", cei, cef), ("cut_edge_inds", kei, kef)):
            if ei.ndim != 2 or ei.shape[0] != 2 or ei.dtype.kind not in "iu":
                raise ValueError(f"{name} must be an integer array of shape [2,E], got {ei.dtype} {tuple(ei.shape)}")
            if ef.size != ei.shape[1]:
                raise ValueError("edge features must hold one value per edge")
            # the upload packs indices as int32 with an unchecked cast: an int64 / uint index of 2**31 or more would wrap, possibly
            # into range, and score another graph -- such lists are rejected here (int32 input cannot overflow; the device
            # flags catch everything that is out of range but representable)
            if ei.dtype != np.int32 and ei.size and (int(ei.max()) > 2 ** 31 - 1 or int(ei.min()) < -2 ** 31):
                raise ValueError("edge index out of range (left ids must be in [0,n_left), variable ids in [0,n_vars))")
        # The specialised plan wants lists sorted by row, which is what get_state emits (utils.py:102-104).  Packing is one native
        # pass per list (gcnn_host_pack_edges): copy into the staging buffer, look at the order on the way, and only a list in
        # another order goes through a stable counting sort on the host (tens of microseconds for a few 10^4 entries; NumPy's
        # stable argsort alone would take longer than the whole general path).
        key = (c.shape[0], v.shape[0], k.shape[0], cei.shape[1], kei.shape[1])
        lay = self._layout(key)
        if lay is False or (want_order and key[2] > 4096):
            raise _UseGeneralPath()
        dims, L, in_off, out_off = lay
        dev = self.model.device
        if self.pin_in is None or self.pin_in.numel() < L.in_bytes:
            self.pin_in = torch.empty(max(2 * L.in_bytes, 1 << 20), dtype=torch.uint8).pin_memory()
            self.in_np = self.pin_in.numpy()
        if self.pin_out is None or self.pin_out.numel() < L.out_bytes:
            self.pin_out = torch.empty(max(2 * L.out_bytes, 1 << 16), dtype=torch.uint8).pin_memory()
            self.out_np = self.pin_out.numpy()
        if self.arena is None or self.arena.numel() < L.arena_bytes:
            self.arena = None
            self.arena = torch.empty(max(2 * L.arena_bytes, 1 << 24), dtype=torch.uint8, device=dev)
        buf = self.in_np
        buf[in_off[0]:in_off[1]] = 0      # the plan's counters and flags travel zeroed inside the upload
        base = self.pin_in.data_ptr()
        for off, a, dt in ((in_off[1], c, np.float32), (in_off[4], v, np.float32), (in_off[5], k, np.float32)):
            if a.size:
                np.copyto(buf[off:off + 4 * a.size].view(dt).reshape(a.shape), a, casting="unsafe")
        for io, fo, ei, ef, n_left in ((in_off[2], in_off[3], cei, cef, key[0]), (in_off[6], in_off[7], kei, kef, key[2])):
            if not ei.size:
                continue
            ei32 = np.ascontiguousarray(ei, dtype=np.int32)                  # no copies for what get_state hands over
            ef32 = np.ascontiguousarray(ef, dtype=np.float32).reshape(-1)
            if self.sort_scratch is None or self.sort_scratch.size < n_left + 1:
                self.sort_scratch = np.empty(2 * (n_left + 1), np.int32)
            rc = _lib.lib().gcnn_host_pack_edges(ei32.ctypes.data, ei32.ctypes.data + 4 * ei32.shape[1], ef32.ctypes.data,
                                                 ei32.shape[1], n_left, base + io, base + fo, self.sort_scratch.ctypes.data)
            if rc < 0:      # (0 / 1 / 2: packed -- as it was, sorted here, or as it was with a row id the device check reports)
                _lib.check(rc, "gcnn_host_pack_edges")
        t1 = time.perf_counter()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev)
            _lib.check(_lib.lib().gcnn_infer(C.byref(dims), C.c_void_p(self.model._flat.data_ptr()),
                                            C.c_void_p(self.pin_in.data_ptr()), C.c_void_p(self.pin_out.data_ptr()),
                                            C.c_void_p(self.arena.data_ptr()), self.arena.numel(), int(want_order),
                                            C.c_void_p(stream.cuda_stream)), "gcnn_infer")
            t2 = time.perf_counter()
            stream.synchronize()
        t3 = time.perf_counter()
        if timings is not None:
            timings.update(pack=t1 - t0, enqueue=t2 - t1, wait=t3 - t2)
        out = self.out_np
        flags = out[out_off[2]:out_off[2] + 16].view(np.int32)
        if flags[0]:
            raise ValueError("edge index out of range (left ids must be in [0,n_left), variable ids in [0,n_vars))")
        if flags[1] or flags[2] or flags[3]:
            raise _UseGeneralPath()
        n = key[2]
        scores = out[out_off[0]:out_off[0] + 4 * n].view(np.float32).copy().view(ScoreArray)
        if want_order:
            scores.rankings = out[out_off[1]:out_off[1] + 4 * n].view(np.int32).copy()
        return scores


class Batch:
    """A stacked mini-batch resident on the GPU: features + both CSR orders of both edge sets (GCNN.prepare)."""

    def __init__(self, cons_feats, var_feats, cut_feats, cons_graph, cut_graph):
        self.cons_feats, self.var_feats, self.cut_feats = cons_feats, var_feats, cut_feats
        self.cons_graph, self.cut_graph = cons_graph, cut_graph
        self.dims = _lib.Dims(cons_feats.shape[0], var_feats.shape[0], cut_feats.shape[0], cons_graph.n_edges,
                              cut_graph.n_edges)
        self.n_edges = cons_graph.n_edges + cut_graph.n_edges
        self.device = cons_feats.device


def _as_device(x, dtype, device):
    if isinstance(x, torch.Tensor):
        t = x.detach()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype != dtype:
        if dtype == torch.int32 and t.dtype in (torch.int64, torch.int16, torch.uint8, torch.int8):
            t = t.to(torch.int32)
        elif dtype == torch.float32 and t.dtype in (torch.float64, torch.float16, torch.bfloat16, torch.bool,
                                                    torch.int64, torch.int32):
            t = t.to(torch.float32)
        else:
            raise ValueError(f"expected {dtype}, got {t.dtype}")
    return t.to(device, non_blocking=True).contiguous()


class _GCNNFunction(torch.autograd.Function):
    """Autograd node around gcnn_forward / gcnn_backward (the role tf.GradientTape plays in model_trainer.py:269-272)."""

    @staticmethod
    def forward(ctx, flat, model, batch):
        ws = model._take_workspace(batch)
        scores = model._forward_into(flat, batch, ws)
        ctx.model, ctx.batch, ctx.ws = model, batch, ws
        ctx.save_for_backward(flat)
        return scores

    @staticmethod
    def backward(ctx, d_scores):
        model, batch, ws = ctx.model, ctx.batch, ctx.ws
        if ws is None:
            raise RuntimeError("GCNN backward called twice on the same forward pass")
        (flat,) = ctx.saved_tensors
        grads = torch.zeros_like(flat)  # non-trainable / padding slots are never written by gcnn_backward
        model._backward_into(flat, batch, ws, d_scores.contiguous().to(torch.float32), grads)
        model._give_workspace(ws)
        ctx.ws = None
        return grads, None, None


class GCNN:
    """Bipartite GCNN cut scorer (the reference's `GCNN(BaseModel)`, model.py:136-300) on MI355X."""

    def __init__(self, device=None, seed=None):
        self.emb_size, self.cons_feats, self.edge_feats, self.var_feats, self.cut_feats = EMB, 4, 1, 14, 6
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.GcnnError("GCNN needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))
        self.device = torch.device(device)
        layout, total = _lib.param_layout()
        if len(layout) != len(VARIABLE_SPEC):
            raise _lib.GcnnError("library / binding disagree on the number of model variables")
        for (off, rows, cols, tr), (name, shape, trainable) in zip(layout, VARIABLE_SPEC):
            if rows * cols != int(np.prod(shape)) or tr != trainable:
                raise _lib.GcnnError(f"library / binding disagree on variable {name}")
        self._layout, self._total = layout, total
        self._flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._flat.requires_grad_(True)
        mask = torch.zeros(total, dtype=torch.bool)
        for (off, rows, cols, tr) in layout:
            if tr:
                mask[off:off + rows * cols] = True
        self._trainable_mask = mask.to(self.device)
        self.variables_topological_order = [name for name, _, _ in VARIABLE_SPEC]
        self.input_signature = [(("float32", (None, 4)), ("int32", (2, None)), ("float32", (None, 1)),
                                 ("float32", (None, 14)), ("float32", (None, 6)), ("int32", (2, None)),
                                 ("float32", (None, 1)), ("int32", ()), ("int32", ()), ("int32", ())), ("bool", ())]
        self._ws_pool = []
        self._session = None      # single-state inference (gcnn_infer): pinned staging + device arena, created on first use
        self._pin = None          # pinned host staging buffer for prepare()
        self._pin_event = None
        self._prenorm_state = None
        self._init_weights(np.random.default_rng(seed))

    # ---- variables ---------------------------------------------------------------------------------------------
    def _init_weights(self, rng):
        """Keras defaults of the reference (model.py:175, 334, 342): orthogonal kernels, zero biases, shift 0, scale 1."""
        host = np.zeros(self._total, np.float32)
        for (off, rows, cols, _), (name, shape, _) in zip(self._layout, VARIABLE_SPEC):
            n = rows * cols
            if name.endswith("/kernel"):
                r, c = shape
                q, tri = np.linalg.qr(rng.standard_normal((max(r, c), min(r, c))))
                q = q * np.sign(np.diag(tri))
                host[off:off + n] = (q if r >= c else q.T).astype(np.float32).reshape(-1)
            elif name.endswith("/scale"):
                host[off:off + n] = 1.0
        with torch.no_grad():
            self._flat.copy_(torch.from_numpy(host))

    @property
    def flat_parameters(self) -> torch.Tensor:
        """The single flat fp32 buffer holding all 62 variables (leaf tensor; `.grad` is filled by backward)."""
        return self._flat

    @property
    def variables(self):
        out = []
        for (off, rows, cols, _), (_, shape, _) in zip(self._layout, VARIABLE_SPEC):
            out.append(self._flat.detach()[off:off + rows * cols].view(shape))
        return out

    @property
    def trainable_variables(self):
        return [v for v, (_, _, tr) in zip(self.variables, VARIABLE_SPEC) if tr]

    def gradients(self, flat_grad=None):
        """The 46 gradient views matching `trainable_variables` (model_trainer.py:272), from a flat gradient buffer."""
        g = self._flat.grad if flat_grad is None else flat_grad
        if g is None:
            raise RuntimeError("no gradient has been computed yet")
        return [g[off:off + rows * cols].view(shape) for (off, rows, cols, tr), (_, shape, _) in
                zip(self._layout, VARIABLE_SPEC) if tr]

    def get_variable(self, name):
        return self.variables[self.variables_topological_order.index(name)]

    def set_weights(self, arrays):
        """Assign the 62 variables from arrays in checkpoint order (dict name->array also accepted)."""
        if isinstance(arrays, dict):
            arrays = [arrays[n] for n in self.variables_topological_order]
        if len(arrays) != len(VARIABLE_SPEC):
            raise ValueError(f"expected {len(VARIABLE_SPEC)} arrays, got {len(arrays)}")
        host = self._flat.detach().cpu().numpy().copy()
        for a, (off, rows, cols, _), (name, shape, _) in zip(arrays, self._layout, VARIABLE_SPEC):
            a = np.asarray(a, dtype=np.float32)
            if a.shape != tuple(shape):
                raise ValueError(f"variable {name}: expected shape {tuple(shape)}, got {a.shape}")
            host[off:off + rows * cols] = a.reshape(-1)
        with torch.no_grad():
            self._flat.copy_(torch.from_numpy(host))

    def get_weights(self):
        host = self._flat.detach().cpu().numpy()
        return [host[off:off + rows * cols].reshape(shape).copy() for (off, rows, cols, _), (_, shape, _) in
                zip(self._layout, VARIABLE_SPEC)]

    def save_state(self, path: str):
        """model.py:47-56: one pickle.dump(np.ndarray) per variable, no header."""
        with open(path, "wb") as file:
            for a in self.get_weights():
                pickle.dump(a, file)

    def restore_state(self, path: str):
        """model.py:58-67.  The records are read with an unpickler that admits NumPy arrays only (`_safe_pickle`)."""
        arrays = []
        with open(path, "rb") as file:
            for _ in VARIABLE_SPEC:
                arrays.append(_safe_pickle.load(file))
        self.set_weights(arrays)

    # ---- inputs ------------------------------------------------------------------------------------------------
    def _stage_host_arrays(self, arrays):
        """ONE host->device copy for all seven input arrays: pack them (16-byte aligned) into a pinned staging buffer,
        copy once, and hand out typed views of the device buffer (which the views keep alive)."""
        specs, total = [], 0
        for a, dt in arrays:
            a = np.ascontiguousarray(np.asarray(a), dtype=dt)
            specs.append((a, total))
            total += (a.nbytes + 15) & ~15
        total = max(total, 16)
        if self._pin is None or self._pin.numel() < total:
            self._pin = torch.empty(max(total, 1 << 16), dtype=torch.uint8).pin_memory()
            self._pin_event = None
        if self._pin_event is not None:
            self._pin_event.synchronize()      # the previous batch's copy must have left the staging buffer
        host = self._pin.numpy()
        for a, off in specs:
            host[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        dev_buf = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev_buf.copy_(self._pin[:total], non_blocking=True)
        self._pin_event = torch.cuda.Event()
        self._pin_event.record(torch.cuda.current_stream(self.device))
        out = []
        for a, off in specs:
            tdt = torch.float32 if a.dtype == np.float32 else torch.int32
            out.append(dev_buf[off:off + a.nbytes].view(tdt).view(a.shape))
        return out

    def prepare(self, inputs, validate=True) -> Batch:
        """10-tuple of NumPy arrays / torch tensors (model.py:263-275) -> device-resident Batch with CSR plans."""
        if isinstance(inputs, Batch):
            return inputs
        if len(inputs) != 10:
            raise ValueError(f"expected the reference's 10-tuple input, got {len(inputs)} items")
        c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts = inputs
        dev = self.device
        if not any(isinstance(x, torch.Tensor) for x in (c, cei, cef, v, k, kei, kef)):
            for name, x, kind in (("cons_edge_inds", cei, "iu"), ("cut_edge_inds", kei, "iu")):
                if np.asarray(x).dtype.kind not in kind:
                    raise ValueError(f"{name} must be an integer array, got {np.asarray(x).dtype}")
            f32, i32 = np.float32, np.int32
            c, cei, cef, v, k, kei, kef = self._stage_host_arrays(
                [(c, f32), (cei, i32), (cef, f32), (v, f32), (k, f32), (kei, i32), (kef, f32)])
        else:
            c, v, k = (_as_device(x, torch.float32, dev) for x in (c, v, k))
            cei, kei = _as_device(cei, torch.int32, dev), _as_device(kei, torch.int32, dev)
            cef, kef = _as_device(cef, torch.float32, dev), _as_device(kef, torch.float32, dev)
        for name, t, f in (("cons_feats", c, 4), ("var_feats", v, 14), ("cut_feats", k, 6)):
            if t.dim() != 2 or t.shape[1] != f:
                raise ValueError(f"{name} must have shape [N,{f}], got {tuple(t.shape)}")
        for name, total, t in (("n_cons", n_cons, c), ("n_vars", n_vars, v), ("n_cuts", n_cuts, k)):
            if int(total) != t.shape[0]:
                raise ValueError(f"{name}={int(total)} does not match the {t.shape[0]} feature rows")
        return Batch(c, v, k, BipartiteGraph(cei, cef, c.shape[0], v.shape[0], validate),
                     BipartiteGraph(kei, kef, k.shape[0], v.shape[0], validate))

    # ---- workspaces --------------------------------------------------------------------------------------------
    def _take_workspace(self, batch):
        need = _lib.lib().gcnn_workspace_floats(C.byref(batch.dims))
        best = None
        for i, ws in enumerate(self._ws_pool):
            if ws.numel() >= need and (best is None or ws.numel() < self._ws_pool[best].numel()):
                best = i
        if best is not None:
            return self._ws_pool.pop(best)
        try:
            return torch.empty(max(need, 4), dtype=torch.float32, device=self.device)
        except torch.OutOfMemoryError:
            self._ws_pool.clear()
            torch.cuda.empty_cache()
            return torch.empty(max(need, 4), dtype=torch.float32, device=self.device)

    def _give_workspace(self, ws):
        self._ws_pool.append(ws)
        if len(self._ws_pool) > 2:
            self._ws_pool.sort(key=lambda t: t.numel())
            self._ws_pool.pop(0)

    # ---- forward / backward ------------------------------------------------------------------------------------
    def _forward_into(self, flat, batch, ws, save=True):
        scores = torch.empty(batch.dims.n_cuts, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gcnn_forward(C.byref(batch.dims), _ptr(flat), _ptr(batch.cons_feats),
                                               _ptr(batch.var_feats), _ptr(batch.cut_feats), C.byref(batch.cons_graph.c),
                                               C.byref(batch.cut_graph.c), _ptr(ws), ws.numel(), _ptr(scores),
                                               int(save), _stream(self.device)), "gcnn_forward")
        return scores

    def _forward_loss_into(self, flat, batch, ws, targets, loss_scale):
        """Forward with the MSE head fused into its last launch (gcnn_forward_loss); continue with
        `_backward_into(d_scores=None, ..., loss_out=...)`."""
        scores = torch.empty(batch.dims.n_cuts, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gcnn_forward_loss(C.byref(batch.dims), _ptr(flat), _ptr(batch.cons_feats),
                                                    _ptr(batch.var_feats), _ptr(batch.cut_feats), C.byref(batch.cons_graph.c),
                                                    C.byref(batch.cut_graph.c), _ptr(ws), ws.numel(), _ptr(scores),
                                                    _ptr(targets), float(loss_scale), _stream(self.device)), "gcnn_forward_loss")
        return scores

    def _backward_into(self, flat, batch, ws, d_scores, grads, count_slot=None, loss_out=None, adam=None):
        """`adam`: optional `_lib.AdamArgs` -- the optimizer step then rides in the backward's last launch."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gcnn_backward(C.byref(batch.dims), _ptr(flat), _ptr(batch.cons_feats),
                                                _ptr(batch.var_feats), _ptr(batch.cut_feats), C.byref(batch.cons_graph.c),
                                                C.byref(batch.cut_graph.c), _ptr(ws), ws.numel(), _ptr(d_scores),
                                                _ptr(grads), _ptr(count_slot), _ptr(loss_out),
                                                C.byref(adam) if adam is not None else None, _stream(self.device)),
                       "gcnn_backward")

    def call(self, inputs, training=False):
        """GCNN.call (model.py:257-300): flat fp32 scores, one per candidate cut.  `training` is accepted and ignored
        exactly like the reference (no dropout / batch-norm).  Differentiable w.r.t. `flat_parameters` under autograd."""
        batch = self.prepare(inputs)
        if torch.is_grad_enabled() and self._flat.requires_grad:
            scores = _GCNNFunction.apply(self._flat, self, batch)
        else:
            ws = self._take_workspace(batch)
            scores = self._forward_into(self._flat.detach(), batch, ws, save=False)
            self._give_workspace(ws)
        return scores.as_subclass(ScoreTensor)

    def __call__(self, inputs, training=False):
        return self.call(inputs, training)

    # ---- PreNorm pretraining hooks (model.py:69-133, 384-437) --------------------------------------------------------
    def pretrain_init(self):
        """BaseModel.pretrain_init (model.py:69-87): every PreNorm layer starts waiting for updates."""
        self._prenorm_state = [dict(waiting=True, received=False, mean=np.zeros(u, np.float32), var=np.zeros(u, np.float32),
                                    count=np.float32(0)) for _, _, u in PRENORM_LAYERS]

    def pretrain(self, inputs, training=True) -> bool:
        """BaseModel.pretrain (model.py:119-133): run the model; the first PreNorm layer (in call order) that is still
        waiting absorbs this batch's statistics (PreNormLayer.update_params, model.py:394-423) and the call stops there
        (the reference raises PreNormException).  Returns True when a layer absorbed the batch."""
        if self._prenorm_state is None:
            return False
        waiting = [i for i, st in enumerate(self._prenorm_state) if st["waiting"]]
        if not waiting:
            return False
        layer = waiting[0]
        st = self._prenorm_state[layer]
        units = PRENORM_LAYERS[layer][2]
        batch = self.prepare(inputs)
        sizes = [batch.dims.n_cons, batch.dims.n_cons_edges, batch.dims.n_vars, batch.dims.n_cuts, batch.dims.n_cut_edges]
        if layer <= 4:
            sample_count = sizes[layer]
        else:
            conv, post = (layer - 5) // 2, (layer - 5) % 2
            n_recv = [batch.dims.n_cons, batch.dims.n_vars, batch.dims.n_cuts][conv]
            n_edge = [batch.dims.n_cons_edges, batch.dims.n_cons_edges, batch.dims.n_cut_edges][conv]
            sample_count = (n_recv if post else n_edge) * EMB
        st["received"] = True
        if sample_count == 0:
            return True
        ws = self._take_workspace(batch)
        flat = self._flat.detach()
        if layer >= 5:
            self._forward_into(flat, batch, ws, save=2)   # 2: the two-layer form that materialises A (the post-conv PreNorm's input)
        out = torch.empty(2 * units, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gcnn_prenorm_stats(C.byref(batch.dims), _ptr(flat), _ptr(batch.cons_feats),
                                                     _ptr(batch.var_feats), _ptr(batch.cut_feats), C.byref(batch.cons_graph.c),
                                                     C.byref(batch.cut_graph.c), _ptr(ws), ws.numel(), layer, _ptr(out),
                                                     _stream(self.device)), "gcnn_prenorm_stats")
        host = out.cpu().numpy()
        self._give_workspace(ws)
        # streaming merge of Chan et al. in fp32, exactly the arithmetic of model.py:415-423
        f = np.float32
        sample_mean, sample_var, sample_count = host[:units].astype(f), host[units:].astype(f), f(sample_count)
        delta = sample_mean - st["mean"]
        m2 = st["var"] * st["count"] + sample_var * sample_count + delta ** 2 * st["count"] * sample_count / (st["count"] + sample_count)
        st["count"] = f(st["count"] + sample_count)
        st["mean"] = (st["mean"] + delta * sample_count / st["count"]).astype(f)
        st["var"] = (m2 / st["count"]).astype(f)
        return True

    def pretrain_sync(self, process_group):
        """Data-parallel fitting: merge the statistics of the layer that is absorbing updates across the ranks of
        `process_group` (every rank saw only its shard of the pretraining batches).  Call once per pass, on every rank,
        before `pretrain_next`.  The merge is the same Chan update as between batches (model.py:415-423), in rank order."""
        if self._prenorm_state is None or process_group is None:
            return
        waiting = [i for i, st in enumerate(self._prenorm_state) if st["waiting"]]
        if not waiting:          # identical on every rank: the set of fitted layers only changes through this method's callers
            return
        from .parallel import allgather_prenorm
        st = self._prenorm_state[waiting[0]]
        st["count"], st["mean"], st["var"], st["received"] = allgather_prenorm(st["count"], st["mean"], st["var"], st["received"],
                                                                              process_group, self.device)

    def pretrain_next(self):
        """BaseModel.pretrain_next (model.py:89-117): freeze the layer that just received updates
        (PreNormLayer.stop_updates, model.py:425-437: shift = -mean, scale = 1/sqrt(var), var == 0 -> 1)."""
        if self._prenorm_state is None:
            return None
        for i, st in enumerate(self._prenorm_state):
            if st["waiting"] and st["received"]:
                shift_name, scale_name, units = PRENORM_LAYERS[i]
                var = np.where(st["var"] == 0, np.float32(1), st["var"]).astype(np.float32)
                with torch.no_grad():
                    if shift_name is not None:
                        self.get_variable(shift_name).copy_(torch.from_numpy((-st["mean"]).astype(np.float32)))
                    self.get_variable(scale_name).copy_(torch.from_numpy((1 / np.sqrt(var)).astype(np.float32)))
                st["waiting"] = False
                return i, scale_name.rsplit("/", 1)[0]
        return None

    # Below this many cuts the descending stable ranking is done on the host (a stable NumPy argsort of a few dozen floats takes
    # ~2 us; the device ranking kernel costs a launch plus a dependent kernel in the chain: ~10 us end to end, tools/latency.py)
    HOST_RANK_MAX = 1024

    def score_state(self, inputs, rank=False):
        """Scores of ONE sampled state given as host arrays (the SCIP plugins' call, model_evaluator.py:84-111), through the
        single-call path gcnn_infer: one upload, a three-launch graph plan, the inference forward pass, one download.
        Returns a `ScoreArray` (ndarray with `.numpy()`); with `rank=True` its `.rankings` holds the cut indices in descending
        score order (ties in index order): computed by the device ranking kernel for more than HOST_RANK_MAX cuts or with
        `rank="device"`, else by a stable argsort on the host (faster for a few dozen cuts).  States the specialised plan declines (edge lists not sorted by row, more than
        32,768 variables, ...) run through `prepare` + the general forward pass instead; results are identical."""
        if self._session is None:
            self._session = _InferenceSession(self)
        try:
            n_cuts = int(np.asarray(inputs[4]).shape[0])
            on_device = bool(rank) and (rank == "device" or n_cuts > self.HOST_RANK_MAX)
            scores = self._session.run(inputs, on_device)
            if rank and not on_device:
                scores.rankings = np.argsort(-np.asarray(scores), kind="stable").astype(np.int32)
            return scores
        except _UseGeneralPath:
            with torch.no_grad():
                scores = self.call(inputs, False).numpy().view(ScoreArray)
            if rank:
                scores.rankings = np.argsort(-np.asarray(scores), kind="stable").astype(np.int32)
            return scores

    def get_concrete_function(self):
        """Counterpart of `tf.function(model.call).get_concrete_function()` (model_evaluator.py:310-311): an inference
        callable `(state10, training) -> scores` with `.numpy()`.  Host arrays (what `get_state` produces) take the
        single-call path `score_state`; device tensors / prepared batches the general forward pass."""
        def get_improvements(state, training=False, rank=False):
            if isinstance(state, Batch) or any(isinstance(x, torch.Tensor) for x in state[:7]):
                with torch.no_grad():
                    return self.call(state, training)
            return self.score_state(state, rank)
        return get_improvements
