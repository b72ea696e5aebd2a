#!/usr/bin/env python3
"""Headline benchmark: bipartite-graph edges/sec (fwd+bwd) per training step, setcov-500 batch=32 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + MSE + backward (+ one RCCL all-reduce of the flat gradient buffer when N > 1) + Keras-form Adam on
one stacked synthetic batch (32 setcov-500 samples PER GPU: weak scaling) already resident in HBM.  Rank 0 prints ONE
JSON line.  `roofline` is the standalone scatter-sum pass (K9, the pass BASELINE.json's 40%-of-HBM target names) timed
live with HIP events on its own stream-ordered loop; `roofline_fused` is the production fused edge kernel of the same
convolution; `cpu_baseline` is the op-for-op CPU restatement (oracle/, a stand-in for the reference's TF-CPU path,
which cannot be installed here) timed on this host's cores."""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming-copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--problem", default="setcov", choices=["setcov", "combauc", "capfac", "indset"])
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the step once and replay it as a hipGraph (measured: no faster than eager issue -- the GPU-side dependency chain is the limit, not the host)")
    return ap.parse_args()


def event_time_ms(fn, iters, warmup=3):
    """Average device time of `fn` (stream-ordered launches on torch's current stream) with HIP events."""
    for _ in range(warmup):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters


def roofline_scatter_sum(batch, dev):
    """K9 standalone on the v->c convolution's shape: messages [E1,64] fp32 + receiver segments -> [C,64].
    Three message buffers are rotated so the working set (3 x 205 MB at setcov x 32) exceeds the 256 MiB Infinity Cache."""
    from gcnn_cut_selector_amd import _lib
    from gcnn_cut_selector_amd.graph import _ptr, _stream
    g = batch.cons_graph
    e, r = g.n_edges, g.n_left
    if e == 0 or r == 0:
        return None
    nbuf = max(2, int(np.ceil(300e6 / (e * 256.0))) + 1)
    msgs = [torch.randn(e, 64, device=dev) for _ in range(nbuf)]
    out = torch.empty(r, 64, device=dev)
    lib, state = _lib.lib(), {"i": 0}

    def run():
        m = msgs[state["i"] % nbuf]
        state["i"] += 1
        _lib.check(lib.gcnn_seg_sum_f32(_ptr(m), _ptr(g.l_ptr), None, r, _ptr(out), _stream(dev)), "gcnn_seg_sum_f32")

    ms = event_time_ms(run, 30 * nbuf, warmup=nbuf)
    nbytes = 260.0 * e + 256.0 * r  # 256+4 B per edge in, 256 B per receiver out
    achieved = nbytes / (ms * 1e-3) / 1e9
    traffic = None  # HBM bytes per launch from the committed PMC passes (same shape only); see profiles/README.md
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath) and (e, r) == (800000, 16000):
        traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"])
    return {"kernel": "k_seg_sum (scatter-sum pass, conv v->c shape)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "bytes_per_launch": nbytes, "us_per_launch": round(ms * 1e3, 2), "edges": e, "receivers": r,
            "frac_of_measured_copy_peak": round(achieved / 6290.0, 4)}


def roofline_fused_edge(model, batch, dev):
    """The production fused edge kernel (gather + ReLU + segmented sum, K8 hoisted) of conv v->c, timed alone."""
    from gcnn_cut_selector_amd import ops
    g = batch.cons_graph
    if g.n_edges == 0:
        return None
    pl, pr = torch.randn(g.n_left, 64, device=dev), torch.randn(g.n_var, 64, device=dev)
    w = torch.randn(64, device=dev)
    one, zero = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    ms = event_time_ms(lambda: ops.conv_edge_fwd(g, True, pl, pr, w, zero, one, one), 50)
    e, l, v = g.n_edges, g.n_left, g.n_var
    nbytes = 12.0 * e + 256.0 * (l + v) + 256.0 * l  # compulsory HBM bytes, SURVEY.md section 8(d)
    gathered = 256.0 * e
    return {"kernel": "k_edge<fwd> (fused gather+relu+segmented sum, conv v->c)", "us_per_launch": round(ms * 1e3, 2),
            "edges_per_s": round(e / (ms * 1e-3), 1), "compulsory_hbm_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "row_gather_GBs": round(gathered / (ms * 1e-3) / 1e9, 1)}


def cpu_baseline(problem, batch_size, first_sample, budget_s=20.0):
    """The op-for-op CPU restatement (oracle/, torch CPU fp32, autograd) on the same synthetic batch: fwd+MSE+bwd."""
    from gcnn_cut_selector_amd import synthetic
    from oracle import gcnn_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    state, y, _ = synthetic.make_batch(problem, batch_size, first_sample)
    params = O.randomize_params(O.init_params(0), 1)
    p = O.to_torch(params, torch.float32, requires_grad=True)
    inputs = O.as_inputs(state, torch.float32)
    yt = torch.as_tensor(y)
    leaves = [p[n] for n, _, t in O.PARAM_SPEC if t]
    n_edges = state[1].shape[1] + state[5].shape[1]

    def step():
        loss = ((O.forward(p, inputs) - yt) ** 2).mean()
        torch.autograd.grad(loss, leaves)

    step()
    t0 = time.perf_counter(); step(); one = time.perf_counter() - t0
    n = int(max(3, min(30, budget_s / max(one, 1e-3))))
    times = []
    for _ in range(n):
        t0 = time.perf_counter(); step(); times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(n_edges / med, 1), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of fwd+mse+bwd on the same {problem} batch={batch_size} stacked batch "
                      f"({n_edges} edges), median {med * 1e3:.0f} ms/step, torch {torch.__version__} CPU fp32, "
                      f"os.cpu_count()={os.cpu_count()}; stand-in for the reference's TF-2.7 CPU path"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    group = None
    if world > 1 or os.environ.get("GCNN_FORCE_DP") == "1":  # GCNN_FORCE_DP: rehearse the collective path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI
        group = dist.group.WORLD

    from gcnn_cut_selector_amd import synthetic
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.trainer import Adam, GraphedTrainStep, TrainState, train_step

    model = GCNN(device=dev, seed=0)
    if group is not None:  # replicate rank 0's initial weights
        dist.broadcast(model.flat_parameters.detach(), src=0)
    # per-GPU work is fixed (weak scaling): rank r stacks samples [r*B, (r+1)*B)
    state, y, _ = synthetic.make_batch(args.problem, args.batch, first_sample=rank * args.batch)
    batch = model.prepare(state)
    targets = torch.as_tensor(y).to(dev)
    opt, ts = Adam(1e-4), TrainState(model)
    edges_local = batch.n_edges

    def step():
        return train_step(model, batch, targets, opt, ts, process_group=group)

    launch = "eager"
    if args.graph:
        try:  # the whole step (20 launches, the all-reduce) as one hipGraph replay
            step = GraphedTrainStep(model, batch, targets, opt, ts, process_group=group)
            launch = "hipGraph replay"
        except Exception as exc:  # capture not possible (e.g. a collective that cannot be captured): stay eager
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if group is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    if group is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    edges_total = float(edges_local)
    if group is not None:
        t = torch.tensor([elapsed, float(edges_local)], dtype=torch.float64, device=dev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, edges_total = float(tmax[0]), float(t[1])
    if rank != 0:
        dist.destroy_process_group()
        return
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "bipartite-graph edges/sec (fwd+bwd) per training step, setcov-500 batch=32",
        "value": round(edges_total * args.steps / elapsed, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.problem}-500 rows x batch {args.batch} per GPU (BASELINE configs[1])"
                   if args.problem == "setcov" else f"{args.problem} x batch {args.batch} per GPU",
                   "step": "fwd + mse + bwd" + (" + rccl all-reduce(flat grads)" if group is not None else "") + " + adam",
                   "launch": launch,
                   "global_batch": args.batch * world, "edges_per_step": edges_total, "n_cons": batch.dims.n_cons,
                   "n_vars": batch.dims.n_vars, "n_cuts": batch.dims.n_cuts, "parallelism": f"dp{world}",
                   "final_loss": float(loss)},
    }
    if not args.no_roofline:
        out["roofline"] = roofline_scatter_sum(batch, dev)
        out["roofline_fused"] = roofline_fused_edge(model, batch, dev)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.problem, args.batch, 0)
        out["config"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out))
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
