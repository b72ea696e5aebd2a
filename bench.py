#!/usr/bin/env python3
"""Headline benchmark: bipartite-graph edges/sec (fwd+bwd) per training step, setcov-500 batch=32 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks ITSELF: the parent never
touches the GPU, starts N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relays rank
0's JSON line and exits non-zero if any child fails or if the line does not show N ranks on N distinct devices.  Under
torch.distributed.run the ranks already exist; `--gpus` must then equal WORLD_SIZE.

A step = forward + MSE + backward (+ one RCCL all-reduce of the flat gradient buffer when N > 1) + Keras-form Adam on one
stacked synthetic batch already resident in HBM.  Weak scaling (default): 32 setcov-500 samples PER GPU.  Strong scaling:
the global batch of 32 is split over the ranks by edge count (`parallel.shard_samples`).  Rank 0 prints ONE JSON line.

Timing: a timed block is EXACTLY K steps bracketed by barrier + synchronize on both sides (max over ranks).  Blocks are
repeated until at least --min-seconds of timed region exist (so that a small K still gives a measurable run);
`ms_per_step` / `value` come from the MEDIAN block, HIP events record the same blocks on the compute stream.

`roofline`: the standalone scatter-sum pass (K9, the pass BASELINE.json's 40%-of-HBM target names), timed live with HIP
events in its own loop.  `roofline_step`: the kernels of the training step itself, launch by launch (HIP events inside the
library, gcnn_profile_begin/end), with their algorithmic bytes / FLOPs and the fraction of the HBM / fp32-MFMA peak.
`cpu_baseline`: the op-for-op CPU restatement (oracle/, a stand-in for the reference's TF-CPU path, which cannot be
installed here) timed on this host's cores at 1 thread and at all physical cores."""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy / torch are imported by the worker only: the launcher (`--gpus N` without WORLD_SIZE) must stay off the GPU

HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming-copy ceiling
MFMA_F32_PEAK_TF = 157.3    # dense fp32 MFMA = fp32 vector peak
METRIC = "bipartite-graph edges/sec (fwd+bwd) per training step, setcov-500 batch=32"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE if set, else 1")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--problem", default="setcov", choices=["setcov", "combauc", "capfac", "indset"])
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU (weak scaling) / in the global batch (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step block until this much time is measured")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the step once and replay it as a hipGraph (measured: no faster than eager issue -- the GPU-side dependency chain is the limit, not the host)")
    ap.add_argument("--stub-worker", action="store_true", help=argparse.SUPPRESS)   # tests/test_host.py: ranks on CPU over gloo, no GPU work
    return ap.parse_args(argv)


def _import_heavy():
    global np, torch
    import numpy as np
    import torch


# ---- launcher: `python bench.py --gpus N` starts the N ranks itself ---------------------------------------------------------
def check_line(line, n):
    """The JSON line of an N-rank run must show N ranks on N distinct devices; returns a reason string or None."""
    try:
        out = json.loads(line)
    except (TypeError, ValueError):
        return "rank 0 printed no JSON line"
    d = out.get("distributed") or {}
    ranks = d.get("ranks") or []
    if out.get("n_gpus") != n or d.get("world_size_observed") != n or len(ranks) != n:
        return f"asked for {n} ranks, the line reports n_gpus={out.get('n_gpus')}, world_size_observed={d.get('world_size_observed')}, {len(ranks)} rank records"
    if len({r.get("uuid") or r.get("device") for r in ranks}) != n:
        return f"{n} ranks share devices: " + ", ".join(f"rank {r.get('rank')} -> {r.get('uuid') or r.get('device')}" for r in ranks)
    return None


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run.  Makes no HIP / torch call (it does not even import torch): every rank is a fresh
    `python bench.py ...` child with the usual torch.distributed environment; no exec, nothing is re-launched."""
    import socket
    import subprocess
    with socket.socket() as s:   # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:   # one rank is gone: its peers would wait in a collective for ever
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print(f"[bench] rank {failed[0]} exited with code {failed[1]}; run aborted", file=sys.stderr)
        return 1
    reader.join(timeout=10)
    line = next((ln.strip() for ln in reversed(lines) if ln.lstrip().startswith("{")), None)
    why = check_line(line, n)
    if why:
        print(f"[bench] invalid {n}-rank run: {why}", file=sys.stderr)
        if line:
            print(line, file=sys.stderr)
        return 2
    print(line, flush=True)
    return 0


def stub_worker(args, world, rank):
    """Stand-in rank for the launcher test (no GPU): gloo rendezvous from the environment the launcher set, one all-gather of
    rank records, rank 0 prints a line of the real shape.  GCNN_BENCH_STUB_FAIL=<rank> makes that rank exit 3 instead;
    GCNN_BENCH_STUB_SAME_DEVICE=1 makes every rank report the same device."""
    import torch.distributed as dist
    if os.environ.get("GCNN_BENCH_STUB_FAIL") == str(rank):
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    me = {"rank": rank, "device": int(os.environ["LOCAL_RANK"]),
          "uuid": "stub-0" if os.environ.get("GCNN_BENCH_STUB_SAME_DEVICE") == "1" else f"stub-{os.environ['LOCAL_RANK']}"}
    got = [None] * world
    dist.all_gather_object(got, me)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "stub",
                          "distributed": {"world_size_env": world, "world_size_observed": dist.get_world_size(), "ranks": got}}), flush=True)
    dist.destroy_process_group()


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
    # HBM bytes per launch: NOT measured in this run -- PMC counters need rocprofv3 around the process.  The figure of the
    # committed FETCH_SIZE / WRITE_SIZE passes over this same kernel and shape is attached, with its source.
    traffic, source = None, None
    for name in ("r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath) and (e, r) == (800000, 16000):
            traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"])
            source = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel and shape, committed under profiles/: counters cannot be read inside this process, so tools/collect_round.sh takes them right before the bench line on the same box and any other run of bench.py re-uses them)"
            break
    return {"kernel": "k_seg_sum (standalone scatter-sum pass, conv v->c shape; not on the fused training path)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": source,
            "bytes_per_launch": nbytes, "us_per_launch": round(ms * 1e3, 2), "edges": e, "receivers": r,
            "frac_of_measured_copy_peak": round(achieved / 6290.0, 4)}


def roofline_step(model, batch, targets, dev, steps=20):
    """The training step's own kernels, launch by launch: live HIP-event time (events recorded by the library around every
    launch), algorithmic HBM bytes and fp32 FLOPs per launch, fraction of the peak that binds.  Bytes: every [N,64] fp32
    tensor a launch must read or write once (256 B per row), edge lists 8 B per edge, raw features; rows gathered through
    L2 are listed separately (`l2_gather_bytes`): they are not HBM traffic.  FLOPs: 2*64*64 per row and 64x64 product."""
    from gcnn_cut_selector_amd import _lib
    from gcnn_cut_selector_amd.trainer import Adam, TrainState, train_step
    d = batch.dims
    C, V, K, E1, E2 = d.n_cons, d.n_vars, d.n_cuts, d.n_cons_edges, d.n_cut_edges
    opt, ts = Adam(1e-4), TrainState(model)
    for _ in range(3):
        train_step(model, batch, targets, opt, ts)
    torch.cuda.synchronize()
    per = {}
    order = []
    for _ in range(steps):
        with _lib.launch_profile() as prof:
            train_step(model, batch, targets, opt, ts)
        seen = {}
        for name, ms in prof.launches:
            k = seen.get(name, 0)
            seen[name] = k + 1
            key = (name, k)
            if key not in per:
                per[key] = []
                order.append(key)
            per[key].append(ms * 1e3)
    row, mm = 256.0, 2.0 * 64 * 64
    # (kernel, occurrence) -> (what, HBM bytes, FLOPs, L2-gathered bytes).  HBM bytes = the MINIMUM a launch must move: every
    # DISTINCT [N,64] fp32 tensor it reads or writes counted once (256 B per row), edge lists 8 B per edge, raw features.
    # k_wgrad reads several tensors in more than one job (dZ1 feeds the folded layers' product and the lower half of W1, a raw
    # embedding X up to three products): `hbm_bytes` counts each once (10 / 11 / 10 distinct tensors per constraint / variable /
    # cut row: 4 X operands and 6 / 7 / 6 D operands -- the first embedding layer's output E1 is not among them: the forward
    # pass does not store it, its product's job recomputes it from the raw features) plus the raw features, the segment offsets
    # and the partial slabs it writes; `hbm_bytes_per_job` charges every job its own operands (12 / 14 / 12 per row: 6 / 7 / 6
    # products less the recomputed operand + the first embedding layer's dE1, whose ReLU pattern is 8 B per row; the raw features
    # twice) -- what the launch would move if no second read hit L2.  PMC traffic (FETCH_SIZE + WRITE_SIZE) lies between.
    wg_min = row * (10 * C + 11 * V + 10 * K) + 4.0 * (4 * C + 14 * V + 6 * K) + 12.0 * (C + V + K)
    wg_job = row * (12 * C + 14 * V + 12 * K) + 8.0 * (4 * C + 14 * V + 6 * K) + 12.0 * (C + V + K)
    wg_flops = mm * (6 * C + 7 * V + 6 * K) + 2.0 * 16 * 64 * (C + V + K) + 2.0 * 64 * (4 * C + 16 * V + 8 * K)
    edge_f = lambda e, own, oth: 8.0 * e + row * (own + oth) + 2 * row * own    # tables in, S + N out, (index, coef) per edge
    edge_b = lambda e, own, oth: 8.0 * e + row * (own + 2 * oth) + row * own    # P_send, P_recv + dS in, dP_send out
    model_of = {
        ("k_embed_fwd", 0): ("3 embeddings + 4 projections", V * (56 + 3 * row + 16) + C * (16 + 2 * row + 16) + K * (24 + 2 * row + 16),
                             V * (2 * 14 * 64 + 3 * mm) + C * (2 * 4 * 64 + 2 * mm) + K * (2 * 6 * 64 + 2 * mm), 0),
        ("k_edge_fwd<count>", 0): ("conv v->c edge pass", edge_f(E1, C, V), 14.0 * 64 * E1, row * E1),
        ("k_conv_fwd<proj>", 0): ("conv v->c receiver update (C rows)", C * (5 * row + 16), C * 4 * mm, 0),
        ("k_edge_fwd<count>", 1): ("conv c->v edge pass", edge_f(E1, V, C), 14.0 * 64 * E1, row * E1),
        ("k_conv_fwd<proj>", 1): ("conv c->v receiver update (V rows)", V * (5 * row + 16), V * 4 * mm, 0),
        ("k_edge_fwd<count>", 2): ("conv v->k edge pass", edge_f(E2, K, V), 14.0 * 64 * E2, row * E2),
        ("k_edge_fwd_block<count>", 0): ("conv v->k edge pass (a block per cut row)", edge_f(E2, K, V), 14.0 * 64 * E2, row * E2),
        ("k_conv_turn (readout + loss head + cut-row gradients)", 0):
            ("conv v->k receiver update + readout + MSE head + receiver gradients (K rows, one launch)", K * 11 * row, K * 8 * mm, 0),
        ("k_edge_bwd_send", 0): ("conv v->k sender gradients", edge_b(E2, V, K), 22.0 * 64 * E2, 2 * row * E2),
        ("k_conv_bwd", 0): ("conv c->v receiver gradients (V rows) + cut tail", V * (7 * row + 16) + K * (4 * row + 8), V * 4 * mm + K * 2 * mm, 0),
        ("k_edge_bwd_send", 1): ("conv c->v sender gradients", edge_b(E1, C, V), 22.0 * 64 * E1, 2 * row * E1),
        ("k_conv_bwd", 1): ("conv v->c receiver gradients (C rows)", C * (7 * row + 16), C * 4 * mm, 0),
        ("k_edge_bwd_send", 2): ("conv v->c sender gradients", edge_b(E1, V, C), 22.0 * 64 * E1, 2 * row * E1),
        ("k_tail_bwd", 0): ("embedding tails (V and C rows)", V * (5 * row + 8) + C * (4 * row + 8), V * 3 * mm + C * 2 * mm, 0),
        ("k_wgrad", 0): ("19 weight-gradient products (3 of them recomputing their operand E1) + 3 first layers", wg_min, wg_flops, 0),
    }
    pmc = step_traffic(d)
    out = []
    for key in order:
        us = float(np.median(per[key]))
        entry = {"kernel": key[0], "launch": key[1], "us": round(us, 2)}
        if key in model_of:
            what, nbytes, flops, l2 = model_of[key]
            gbs, tf = nbytes / us / 1e3, flops / us / 1e6
            fh, fm = gbs / HBM_PEAK_GBS, tf / MFMA_F32_PEAK_TF
            entry.update({"what": what, "hbm_bytes": round(nbytes), "flops": round(flops), "GBs": round(gbs, 1), "TFs": round(tf, 2),
                          "bound": "hbm" if fh >= fm else "mfma", "frac": round(max(fh, fm), 4)})
            if key[0] == "k_wgrad":
                entry["hbm_bytes_per_job"] = round(wg_job)
                entry["frac_per_job_bytes"] = round(max(wg_job / us / 1e3 / HBM_PEAK_GBS, fm), 4)
            if l2:
                entry["l2_gather_bytes"] = round(l2)
                entry["l2_gather_GBs"] = round(l2 / us / 1e3, 1)
        if pmc is not None and len(pmc["launches"]) == len(order):   # same launch sequence: attach by position
            rec = pmc["launches"][len(out)]
            entry["traffic"] = round(rec["fetch_bytes"] + rec["write_bytes"])
            entry["traffic_kernel"] = rec["kernel"]
        out.append(entry)
    res = {"steps_profiled": steps, "kernel_us_per_step": round(sum(e["us"] for e in out), 1), "launches": out,
           "hbm_bytes": "minimum per launch: every distinct tensor read or written once (k_wgrad: also hbm_bytes_per_job, every job "
                        "charged its own operands; PMC traffic lies between the two because shared operands hit L2)",
           "note": "HIP events recorded by the library around each launch (gcnn_profile_begin/end); medians over steps_profiled "
                   "steps; event brackets add a few us of launch gaps, so the sum exceeds the un-instrumented step"}
    if pmc is not None:
        res["traffic_source"] = pmc["source"] + (" (attached launch by launch)" if len(pmc["launches"]) == len(order) else
                                                 " (NOT attached: the committed record has another launch sequence)")
    return res


def step_traffic(d):
    """HBM bytes per launch of the training step from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, profiles/hbm_traffic.py --step-json), if the record is for this workload.  Not measured in this run:
    counters need rocprofv3 around the process."""
    for name in ("r03_step_hbm_traffic.json",):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            rec = json.load(open(path))
            if rec.get("dims") == [d.n_cons, d.n_vars, d.n_cuts, d.n_cons_edges, d.n_cut_edges]:
                rec["source"] = f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload committed under profiles/ (FETCH_SIZE doubled per MI355X_MICROARCH.md): counters cannot be read inside this process, so tools/collect_round.sh takes them right before the bench line on the same box and any other run of bench.py re-uses them"
                return rec
    return None


def physical_cores():
    """Physical cores this process may use: unique (package, core) pairs among the CPUs of the affinity mask."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    seen = set()
    try:
        for cpu in allowed:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        return max(1, len(seen))
    except OSError:
        return max(1, len(allowed))


def cpu_baseline(problem, batch_size, first_sample, budget_s=10.0):
    """The op-for-op CPU restatement (oracle/, torch CPU fp32, autograd) on the same synthetic workload: fwd+MSE+bwd, at one
    thread, at 16 threads and at all physical cores.  Bounded: each point gets `budget_s` seconds; the 1-thread point runs on the first
    eighth of the batch (same per-sample shapes; the step is linear in the number of samples)."""
    from gcnn_cut_selector_amd import synthetic
    from oracle import gcnn_oracle as O
    params = O.randomize_params(O.init_params(0), 1)
    p = O.to_torch(params, torch.float32, requires_grad=True)
    leaves = [p[n] for n, _, t in O.PARAM_SPEC if t]

    def point(threads, nsamples):
        torch.set_num_threads(threads)
        state, y, _ = synthetic.make_batch(problem, nsamples, first_sample)
        inputs = O.as_inputs(state, torch.float32)
        yt = torch.as_tensor(y)
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
        return {"value": round(n_edges / med, 1), "unit": "edges/s", "cores": threads,
                "sample": f"{n} steps of fwd+mse+bwd on a stacked {problem} batch of {nsamples} samples ({n_edges} edges), "
                          f"median {med * 1e3:.0f} ms/step"}

    cores = physical_cores()
    full = point(cores, batch_size)
    single = point(1, max(1, batch_size // 8))
    mid = point(min(16, cores), batch_size) if cores > 16 else None   # torch's CPU kernels stop scaling long before 128 threads
    out = dict(full)
    out.update({"kind": "port", "single_thread": single, "threads_16": mid,
                "best_value": max(p["value"] for p in (full, single, mid) if p),
                "host": f"os.cpu_count()={os.cpu_count()}, affinity={len(os.sched_getaffinity(0))}, physical cores used={cores}, "
                        f"torch {torch.__version__} CPU fp32",
                "note": "the oracle (CPU restatement of model.py, same unfused dataflow) is a stand-in for the reference's "
                        "TF-2.7 CPU path, which cannot be installed here"})
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus is not None and args.gpus > 1:   # no ranks exist yet: start them (before anything touches the GPU)
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    elif args.gpus is not None and args.gpus != int(os.environ["WORLD_SIZE"]):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to report a run of another size")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _import_heavy()
    if args.stub_worker:
        return stub_worker(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    group = None
    # RCCL prints a version banner to STDOUT when its communicator comes up; the contract is ONE JSON line on stdout, so
    # everything before that line goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or os.environ.get("GCNN_FORCE_DP") == "1":  # GCNN_FORCE_DP: rehearse the collective path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI
        group = dist.group.WORLD

    from gcnn_cut_selector_amd import synthetic
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.parallel import shard_samples
    from gcnn_cut_selector_amd.trainer import Adam, GraphedTrainStep, TrainState, train_step

    model = GCNN(device=dev, seed=0)
    if group is not None:  # replicate rank 0's initial weights
        dist.broadcast(model.flat_parameters.detach(), src=0)
    if args.scaling == "weak":      # per-GPU work is fixed: rank r stacks samples [r*B, (r+1)*B)
        samples = [synthetic.make_sample(args.problem, rank * args.batch + i) for i in range(args.batch)]
    else:                           # total work is fixed: the global batch [0, B) is split by edge count
        every = [synthetic.make_sample(args.problem, i) for i in range(args.batch)]
        sizes = [s[0][1]["indices"].shape[1] + s[0][4]["indices"].shape[1] for s in every]
        samples = [every[i] for i in shard_samples(sizes, world)[rank]]
    if samples:
        b = synthetic.stack_samples(samples)
        state, y = b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum())), b[10]
    else:                           # a rank without samples still takes part in the all-reduce
        z2, f32 = np.zeros((2, 0), np.int32), np.float32
        state = (np.zeros((0, 4), f32), z2, np.zeros((0, 1), f32), np.zeros((0, 14), f32), np.zeros((0, 6), f32), z2,
                 np.zeros((0, 1), f32), 0, 0, 0)
        y = np.zeros(0, f32)
    batch = model.prepare(state)
    targets = torch.as_tensor(y).to(dev)
    opt, ts = Adam(1e-4), TrainState(model)
    edges_local = batch.n_edges

    def step():
        return train_step(model, batch, targets, opt, ts, process_group=group)

    launch = "eager"
    if args.graph:
        try:  # the whole step (15 launches, the all-reduce) as one hipGraph replay
            step = GraphedTrainStep(model, batch, targets, opt, ts, process_group=group)
            launch = "hipGraph replay"
        except Exception as exc:  # capture not possible (e.g. a collective that cannot be captured): stay eager
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    blocks_wall, blocks_dev = [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    loss = None
    while True:
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):     # EXACTLY K steps per timed block
            loss, _ = step()
        ev1.record()
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        if group is not None:           # the block took as long as its slowest rank; every rank sees the same numbers
            t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, dev_ms = float(t[0]), float(t[1])
        blocks_wall.append(elapsed)
        blocks_dev.append(dev_ms)
        if sum(blocks_wall) >= args.min_seconds or len(blocks_wall) >= 1000:
            break
    from gcnn_cut_selector_amd import _lib
    with _lib.launch_profile() as prof:   # one more step on every rank, outside the timed region: how many launches a step is
        train_step(model, batch, targets, opt, ts, process_group=group)
    n_launches = len(prof.launches)
    edges_total = float(edges_local)
    rank_info = [{"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev),
                  "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")), "edges_per_step": edges_local,
                  "ms_per_step_local": round(float(np.median(blocks_dev)) / args.steps, 4)}]
    world_seen = 1
    if group is not None:
        t = torch.tensor([float(edges_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        edges_total = float(t[0])
        world_seen = dist.get_world_size()
        gathered = [None] * world_seen
        dist.all_gather_object(gathered, rank_info[0])
        rank_info = gathered
    if rank != 0:
        dist.destroy_process_group()
        return
    if world_seen != world or (world > 1 and len({r["uuid"] or r["device"] for r in rank_info}) != world
                               and os.environ.get("GCNN_ALLOW_SHARED_DEVICE") != "1"):
        print(f"[bench] {world} ranks asked for, {world_seen} seen on devices {[r['uuid'] or r['device'] for r in rank_info]}", file=sys.stderr)
        raise SystemExit(2)
    med = float(np.median(blocks_wall))
    ms_per_step = med / args.steps * 1e3
    local = [r["ms_per_step_local"] for r in rank_info]
    out = {
        "metric": METRIC,
        "value": round(edges_total * args.steps / med, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"blocks": len(blocks_wall), "steps_timed": len(blocks_wall) * args.steps,
                   "timed_seconds": round(sum(blocks_wall), 4), "statistic": "median over blocks of exactly `steps` steps, each bracketed by barrier + synchronize (max over ranks)",
                   "ms_per_step_hip_events": round(float(np.median(blocks_dev)) / args.steps, 4),
                   "ms_per_step_min_block": round(min(blocks_wall) / args.steps * 1e3, 4),
                   "ms_per_step_max_block": round(max(blocks_wall) / args.steps * 1e3, 4)},
        "config": {"workload": (f"setcov-500 rows x batch {args.batch} per GPU (BASELINE configs[1])" if args.scaling == "weak" else
                                f"setcov-500 rows x global batch {args.batch} split over {world} GPU(s) by edge count")
                   if args.problem == "setcov" else f"{args.problem} x batch {args.batch} ({args.scaling} scaling)",
                   "step": "fwd + mse + bwd" + (" + rccl all-reduce(flat grads)" if group is not None else "") + " + adam",
                   "launch": launch, "library_launches_per_step": n_launches, "problem": args.problem, "batch_per_gpu" if args.scaling == "weak" else "global_batch_samples": args.batch,
                   "global_batch": args.batch * world if args.scaling == "weak" else args.batch, "edges_per_step": edges_total,
                   "n_cons": batch.dims.n_cons, "n_vars": batch.dims.n_vars, "n_cuts": batch.dims.n_cuts,
                   "n_cons_edges": batch.dims.n_cons_edges, "n_cut_edges": batch.dims.n_cut_edges,
                   "parallelism": f"dp{world}", "final_loss": float(loss)},
        "distributed": {"world_size_env": world, "world_size_observed": world_seen,
                        "collective": ("rccl all_reduce (torch.distributed backend nccl), one flat fp32 buffer of "
                                       f"{ts.buf.numel()} elements per step") if group is not None else None,
                        "ranks": rank_info, "ms_per_step_local_min": min(local), "ms_per_step_local_max": max(local)},
    }
    if not args.no_roofline:
        out["roofline"] = roofline_scatter_sum(batch, dev)
        if group is None:
            out["roofline_step"] = roofline_step(model, batch, targets, dev)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.problem, args.batch, 0)
        out["config"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["best_value"], 1)   # against the fastest CPU point
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
