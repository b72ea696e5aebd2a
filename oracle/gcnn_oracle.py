"""CPU oracle for the bipartite GCNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference ships no tests, golden vectors, sample data or trained weights
 and its runtime (TensorFlow 2.7.1 / Keras) is absent from this image, so this
restatement cannot be checked against outputs of the reference itself.  It follows the reference's
source op for op under the documented TF/Keras op semantics and is cross-checked by (a) an independent
NumPy forward written from the same lines, (b) the oracle-free invariants of SURVEY.md section 4 and
(c) hand-computed known answers (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product path (the package next to it) never does.

What is restated (all cites are into /root/reference):
  * PreNormLayer.call                    model.py:365-382   -> prenorm()
  * PreNormLayer.update_params/stop      model.py:394-437   -> PreNormFit
  * PartialGraphConvolution.call         model.py:533-575   -> conv()
  * GCNN.call                            model.py:257-300   -> forward()
  * layer shapes / use_bias / activations model.py:174-208, 486-508 -> PARAM_SPEC
  * variable (checkpoint) order          model.py:53-56, 215 -> PARAM_SPEC order (Keras 2.7 Model.variables:
        tracked sub-layers in attribute order, each layer [kernel, bias] / [shift, scale])
  * loss + gradients                     model_trainer.py:266-273 -> loss_and_grads() (torch autograd plays
        the role of tf.GradientTape)
  * Keras Adam                           model_trainer.py:131,273 -> keras_adam_step()
  * pretraining loop                     model_trainer.py:194-236, model.py:69-133 -> pretrain()
  * ranking-prefix accuracy              model_trainer.py:280-302 -> ranking_fraction()
Op semantics honoured: Keras Dense = x @ W(in,out) + b; tf.gather(axis=0) = row select; tf.scatter_nd into
zeros SUMS duplicates and leaves untouched rows at 0; Keras MeanSquaredError on 1-D input = scalar mean;
Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), theta -= lr_t*m/(sqrt(v)+eps), eps = 1e-7.
"""

from __future__ import annotations

import numpy as np
import torch

EMB = 64
CONS_F, EDGE_F, VAR_F, CUT_F = 4, 1, 14, 6

# (name, shape, trainable) in the reference's checkpoint order (model.py:53-56, 215).
def _emb(prefix, f):
    return [(f"{prefix}_prenorm/shift", (f,), False), (f"{prefix}_prenorm/scale", (f,), False),
            (f"{prefix}_emb_1/kernel", (f, EMB), True), (f"{prefix}_emb_1/bias", (EMB,), True),
            (f"{prefix}_emb_2/kernel", (EMB, EMB), True), (f"{prefix}_emb_2/bias", (EMB,), True)]


def _conv(name):
    return [(f"{name}_feat_left/kernel", (EMB, EMB), True), (f"{name}_feat_left/bias", (EMB,), True),
            (f"{name}_feat_edge/kernel", (1, EMB), True),
            (f"{name}_feat_right/kernel", (EMB, EMB), True),
            (f"{name}_final_prenorm/scale", (1,), False),
            (f"{name}_feat_final/kernel", (EMB, EMB), True), (f"{name}_feat_final/bias", (EMB,), True),
            (f"{name}_post_prenorm/scale", (1,), False),
            (f"{name}_out_1/kernel", (2 * EMB, EMB), True), (f"{name}_out_1/bias", (EMB,), True),
            (f"{name}_out_2/kernel", (EMB, EMB), True), (f"{name}_out_2/bias", (EMB,), True)]


PARAM_SPEC = (_emb("cons", CONS_F)
              + [("cons_edge_prenorm/shift", (1,), False), ("cons_edge_prenorm/scale", (1,), False)]
              + _emb("var", VAR_F) + _emb("cut", CUT_F)
              + [("cut_edge_prenorm/shift", (1,), False), ("cut_edge_prenorm/scale", (1,), False)]
              + _conv("cons_conv") + _conv("var_conv") + _conv("cut_conv")
              + [("out_1/kernel", (EMB, EMB), True), ("out_1/bias", (EMB,), True),
                 ("out_2/kernel", (EMB, 1), True), ("out_2/bias", (1,), True)])
PARAM_NAMES = [n for n, _, _ in PARAM_SPEC]
assert len(PARAM_SPEC) == 62 and sum(t for _, _, t in PARAM_SPEC) == 46
assert sum(int(np.prod(s)) for _, s, t in PARAM_SPEC if t) == 93121

# The 11 PreNorm layers in CALL order (model.py:287-296 then 563-570): (shift name | None, scale name, n_units).
PRENORM_LAYERS = [("cons_prenorm/shift", "cons_prenorm/scale", CONS_F),
                  ("cons_edge_prenorm/shift", "cons_edge_prenorm/scale", 1),
                  ("var_prenorm/shift", "var_prenorm/scale", VAR_F),
                  ("cut_prenorm/shift", "cut_prenorm/scale", CUT_F),
                  ("cut_edge_prenorm/shift", "cut_edge_prenorm/scale", 1),
                  (None, "cons_conv_final_prenorm/scale", 1), (None, "cons_conv_post_prenorm/scale", 1),
                  (None, "var_conv_final_prenorm/scale", 1), (None, "var_conv_post_prenorm/scale", 1),
                  (None, "cut_conv_final_prenorm/scale", 1), (None, "cut_conv_post_prenorm/scale", 1)]


def init_params(seed: int, dtype=np.float32) -> dict:
    """Keras defaults (model.py:175): orthogonal kernels (gain 1), zero biases; PreNorm shift 0 / scale 1
    (model.py:334,342).  Initial weights cannot be bit-matched to TF's RNG; parity is always via loaded weights."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, _ in PARAM_SPEC:
        if name.endswith("/kernel"):
            rows, cols = shape
            a = rng.standard_normal((max(rows, cols), min(rows, cols)))
            q, r = np.linalg.qr(a)
            q = q * np.sign(np.diag(r))
            out[name] = (q if rows >= cols else q.T).astype(dtype).reshape(shape)
        elif name.endswith("/scale"):
            out[name] = np.ones(shape, dtype)
        else:
            out[name] = np.zeros(shape, dtype)
    return out


def randomize_params(params: dict, seed: int, bias_std=0.1) -> dict:
    """Non-degenerate weights for parity tests: random biases, positive PreNorm scales, random shifts."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, _ in PARAM_SPEC:
        p = params[name]
        if name.endswith("/bias"):
            out[name] = (bias_std * rng.standard_normal(shape)).astype(p.dtype)
        elif name.endswith("/scale"):
            out[name] = rng.uniform(0.5, 1.5, shape).astype(p.dtype)
        elif name.endswith("/shift"):
            out[name] = (0.2 * rng.standard_normal(shape)).astype(p.dtype)
        else:
            out[name] = p.copy()
    return out


def to_torch(params: dict, dtype=torch.float32, requires_grad=False) -> dict:
    out = {}
    for name, _, trainable in PARAM_SPEC:
        t = torch.tensor(np.asarray(params[name]), dtype=dtype)
        if requires_grad and trainable:
            t.requires_grad_(True)
        out[name] = t
    return out


class PreNormAbsorb(Exception):
    """Counterpart of PreNormException (model.py:440-443)."""


class PreNormFit:
    """Streaming population mean/variance of one PreNorm layer (model.py:384-437), in `dtype` arithmetic."""

    def __init__(self, n_units, dtype=torch.float32):
        self.n_units, self.dtype = n_units, dtype
        self.mean = torch.zeros((), dtype=dtype)
        self.var = torch.zeros((), dtype=dtype)
        self.count = torch.zeros((), dtype=dtype)
        self.received = False

    def update(self, x):  # model.py:394-423
        x = x.detach().reshape(-1, self.n_units).to(self.dtype)
        sample_mean = x.mean(0)
        sample_var = ((x - sample_mean) ** 2).mean(0)
        sample_count = torch.tensor(float(x.numel() / self.n_units), dtype=self.dtype)
        delta = sample_mean - self.mean
        m2 = (self.var * self.count + sample_var * sample_count
              + delta ** 2 * self.count * sample_count / (self.count + sample_count))
        self.count = self.count + sample_count
        self.mean = self.mean + delta * sample_count / self.count
        self.var = m2 / self.count if self.count > 0 else torch.ones((), dtype=self.dtype)
        self.received = True

    def finish(self):  # model.py:425-437 -> (shift, scale)
        var = torch.where(self.var == 0, torch.ones_like(self.var), self.var)
        shift = (-self.mean).reshape(-1) if self.mean.ndim else (-self.mean).reshape(1)
        return shift.expand(self.n_units).clone(), (1 / torch.sqrt(var)).reshape(-1).expand(self.n_units).clone()


def prenorm(x, shift, scale, hook=None, key=None):
    """model.py:365-382.  `hook` maps a scale-name to a PreNormFit that is still waiting for updates."""
    if hook is not None and key in hook:
        hook[key].update(x)
        raise PreNormAbsorb(key)
    if shift is not None:
        x = x + shift
    if scale is not None:
        x = x * scale
    return x


def dense(x, w, b=None, relu=False):
    y = x @ w
    if b is not None:
        y = y + b
    return torch.relu(y) if relu else y


def scatter_nd_sum(updates, index, out_size):
    """tf.scatter_nd(indices[:,None], updates, [out_size, d]) (model.py:568-569): sums duplicates into zeros."""
    out = torch.zeros((out_size, updates.shape[1]), dtype=updates.dtype)
    out.index_add_(0, index, updates)
    return out


def conv(p, name, left, ei, ef, var, out_size, from_v, hook=None):
    """PartialGraphConvolution.call, model.py:533-575, materialising every [E,64] tensor like the reference."""
    recv_idx, recv = (ei[0], left) if from_v else (ei[1], var)  # model.py:553-560
    joint = (dense(left, p[f"{name}_feat_left/kernel"], p[f"{name}_feat_left/bias"]).index_select(0, ei[0])
             + dense(ef, p[f"{name}_feat_edge/kernel"])
             + dense(var, p[f"{name}_feat_right/kernel"]).index_select(0, ei[1]))  # model.py:564-565
    joint = prenorm(joint, None, p[f"{name}_final_prenorm/scale"], hook, f"{name}_final_prenorm/scale")
    joint = dense(torch.relu(joint), p[f"{name}_feat_final/kernel"], p[f"{name}_feat_final/bias"])  # :498-500
    agg = scatter_nd_sum(joint, recv_idx, out_size)  # model.py:568-569
    agg = prenorm(agg, None, p[f"{name}_post_prenorm/scale"], hook, f"{name}_post_prenorm/scale")  # :570
    h = dense(torch.cat([agg, recv], dim=1), p[f"{name}_out_1/kernel"], p[f"{name}_out_1/bias"], relu=True)
    return dense(h, p[f"{name}_out_2/kernel"], p[f"{name}_out_2/bias"], relu=True)  # model.py:573


def _embed(p, prefix, x, hook):
    x = prenorm(x, p[f"{prefix}_prenorm/shift"], p[f"{prefix}_prenorm/scale"], hook, f"{prefix}_prenorm/scale")
    x = dense(x, p[f"{prefix}_emb_1/kernel"], p[f"{prefix}_emb_1/bias"], relu=True)
    return dense(x, p[f"{prefix}_emb_2/kernel"], p[f"{prefix}_emb_2/bias"], relu=True)


def as_inputs(state, dtype=torch.float32):
    """10-tuple (numpy or torch) -> torch CPU tensors in the oracle's dtype (indices int64)."""
    c, cei, cef, v, k, kei, kef, nc, nv, nk = state
    f = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    i = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int64)
    return f(c), i(cei), f(cef), f(v), f(k), i(kei), f(kef), int(nc), int(nv), int(nk)


def forward(p, inputs, hook=None):
    """GCNN.call, model.py:257-300.  `inputs` from as_inputs(); returns flat [n_cuts] scores (model.py:300)."""
    c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts = inputs
    c = _embed(p, "cons", c, hook)  # model.py:287
    cef = prenorm(cef, p["cons_edge_prenorm/shift"], p["cons_edge_prenorm/scale"], hook, "cons_edge_prenorm/scale")
    v = _embed(p, "var", v, hook)  # model.py:289
    k = _embed(p, "cut", k, hook)  # model.py:290
    kef = prenorm(kef, p["cut_edge_prenorm/shift"], p["cut_edge_prenorm/scale"], hook, "cut_edge_prenorm/scale")
    c = conv(p, "cons_conv", c, cei, cef, v, n_cons, True, hook)  # model.py:294
    v = conv(p, "var_conv", c, cei, cef, v, n_vars, False, hook)  # model.py:295
    k = conv(p, "cut_conv", k, kei, kef, v, n_cuts, True, hook)  # model.py:296
    out = dense(dense(k, p["out_1/kernel"], p["out_1/bias"], relu=True), p["out_2/kernel"], p["out_2/bias"])
    return out.reshape(-1)  # model.py:299-300


def loss_and_grads(params: dict, state, targets, dtype=torch.float32):
    """model_trainer.py:266-273: loss = mean((pred - y)^2) over ALL cuts; grads w.r.t. the 46 trainables."""
    p = to_torch(params, dtype, requires_grad=True)
    pred = forward(p, as_inputs(state, dtype))
    y = torch.as_tensor(np.asarray(targets), dtype=dtype)
    loss = ((pred - y) ** 2).mean()
    names = [n for n, _, t in PARAM_SPEC if t]
    grads = torch.autograd.grad(loss, [p[n] for n in names])
    return pred.detach().numpy(), float(loss.detach()), {n: g.numpy() for n, g in zip(names, grads)}


def scores(params: dict, state, dtype=torch.float32):
    with torch.no_grad():
        return forward(to_torch(params, dtype), as_inputs(state, dtype)).numpy()


def keras_adam_step(theta, grad, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """One Keras-2.7 Adam update (non-amsgrad) in the arrays' dtype; t is the 1-based step count.
    Note eps sits OUTSIDE the bias-corrected sqrt, unlike torch.optim.Adam."""
    dt = theta.dtype.type
    m = dt(beta1) * m + dt(1 - beta1) * grad
    v = dt(beta2) * v + dt(1 - beta2) * grad * grad
    lr_t = dt(lr) * dt(np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    theta = theta - lr_t * m / (np.sqrt(v) + dt(eps))
    return theta, m, v


def pretrain(params: dict, batches, dtype=torch.float32):
    """model_trainer.pretrain (model_trainer.py:194-236) + BaseModel.pretrain_* (model.py:69-133): fit the 11
    PreNorm layers one at a time, each over every batch with all earlier layers frozen.  Returns (params', n)."""
    params = {k: np.array(v, copy=True) for k, v in params.items()}
    waiting = {scale: PreNormFit(n, dtype) for _, scale, n in PRENORM_LAYERS}
    shift_of = {scale: shift for shift, scale, _ in PRENORM_LAYERS}
    n_done = 0
    while True:
        for state in batches:
            try:
                with torch.no_grad():
                    forward(to_torch(params, dtype), as_inputs(state, dtype), hook=waiting)
                break  # no layer absorbed anything any more (model_trainer.py:221-223)
            except PreNormAbsorb:
                pass
        done = [k for k, f in waiting.items() if f.received]  # pretrain_next, model.py:108-117
        if not done:
            break
        key = done[0]
        shift, scale = waiting.pop(key).finish()
        np_dtype = params[key].dtype
        params[key] = scale.numpy().astype(np_dtype)
        if shift_of[key] is not None:
            params[shift_of[key]] = shift.numpy().astype(np_dtype)
        n_done += 1
    return params, n_done


def ranking_fraction(pred, true):
    """model_trainer.py:288-301: fraction of the ranking prefix on which prediction and truth agree."""
    pr = np.array(sorted(range(len(pred)), key=lambda x: pred[x], reverse=True))
    tr = np.array(sorted(range(len(true)), key=lambda x: true[x], reverse=True))
    diff = pr != tr
    return (int(np.argmax(diff)) if np.any(diff) else len(pred)) / len(pred)


# ---------------------------------------------------------------------------------------------------------
# Independent NumPy forward (written separately from the torch one, explicit loops for gather/scatter on
# small inputs) -- used only to cross-check forward() in tests/test_oracle.py.
def numpy_forward(params: dict, state, dtype=np.float64, loop_scatter=False):
    P = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    c, cei, cef, v, k, kei, kef, n_cons, n_vars, n_cuts = state
    c, cef, v, k, kef = (np.asarray(a, dtype=dtype) for a in (c, cef, v, k, kef))
    cei, kei = np.asarray(cei, dtype=np.int64), np.asarray(kei, dtype=np.int64)
    relu = lambda a: np.maximum(a, 0)

    def emb(x, pre):
        x = (x + P[f"{pre}_prenorm/shift"]) * P[f"{pre}_prenorm/scale"]
        x = relu(x @ P[f"{pre}_emb_1/kernel"] + P[f"{pre}_emb_1/bias"])
        return relu(x @ P[f"{pre}_emb_2/kernel"] + P[f"{pre}_emb_2/bias"])

    def cv(name, left, ei, ef, var, out_size, side):
        pl = left @ P[f"{name}_feat_left/kernel"] + P[f"{name}_feat_left/bias"]
        pr = var @ P[f"{name}_feat_right/kernel"]
        j = pl[ei[0]] + ef @ P[f"{name}_feat_edge/kernel"] + pr[ei[1]]
        m = relu(j * P[f"{name}_final_prenorm/scale"]) @ P[f"{name}_feat_final/kernel"] + P[f"{name}_feat_final/bias"]
        agg = np.zeros((int(out_size), EMB), dtype)
        if loop_scatter:
            for e in range(ei.shape[1]):
                agg[ei[side, e]] += m[e]
        else:
            np.add.at(agg, ei[side], m)
        agg = agg * P[f"{name}_post_prenorm/scale"]
        recv = left if side == 0 else var
        h = relu(np.concatenate([agg, recv], 1) @ P[f"{name}_out_1/kernel"] + P[f"{name}_out_1/bias"])
        return relu(h @ P[f"{name}_out_2/kernel"] + P[f"{name}_out_2/bias"])

    c = emb(c, "cons")
    cef = (cef + P["cons_edge_prenorm/shift"]) * P["cons_edge_prenorm/scale"]
    v = emb(v, "var")
    k = emb(k, "cut")
    kef = (kef + P["cut_edge_prenorm/shift"]) * P["cut_edge_prenorm/scale"]
    c = cv("cons_conv", c, cei, cef, v, n_cons, 0)
    v = cv("var_conv", c, cei, cef, v, n_vars, 1)
    k = cv("cut_conv", k, kei, kef, v, n_cuts, 0)
    return (relu(k @ P["out_1/kernel"] + P["out_1/bias"]) @ P["out_2/kernel"] + P["out_2/bias"]).reshape(-1)
