"""CPU-side checks: the C-ABI library loads and exports every symbol include/gcnn_hip.h declares (no compute calls),
the parameter layout agrees with the oracle's checkpoint spec, sample IO / collation mirror the reference's load_batch,
and the data-parallel host logic (sharding, flat-buffer reduction with global cut-count scaling) is exact."""
import os
import re

import numpy as np
import pytest
import torch

from gcnn_cut_selector_amd import _lib, parallel, synthetic, utils
from gcnn_cut_selector_amd.model import PRENORM_LAYERS, VARIABLE_SPEC
from gcnn_cut_selector_amd.trainer import ranking_fraction
from oracle import gcnn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gcnn_hip.h")).read()
    declared = set(re.findall(r"\b(gcnn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.lib()  # loads libgcnn_hip.so and binds every signature (raises if one is missing)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gcnn_abi_version() == _lib.ABI_VERSION


def test_parameter_layout_matches_checkpoint_spec():
    layout, total = _lib.param_layout()
    assert [(n, tuple(s), t) for n, s, t in VARIABLE_SPEC] == [(n, tuple(s), t) for n, s, t in O.PARAM_SPEC]
    assert PRENORM_LAYERS == O.PRENORM_LAYERS
    assert len(layout) == 62
    end = 0
    for (off, rows, cols, tr), (name, shape, trainable) in zip(layout, O.PARAM_SPEC):
        assert off % 4 == 0 and off >= end, name          # 16-byte aligned, non-overlapping, checkpoint order
        assert rows * cols == int(np.prod(shape)) and tr == trainable, name
        end = off + rows * cols
    assert total >= end and sum(r * c for _, r, c, t in layout if t) == 93121


def test_workspace_size_query_without_gpu():
    import ctypes as C
    d = _lib.Dims(16000, 32000, 1893, 800000, 199704)
    n = _lib.lib().gcnn_workspace_floats(C.byref(d))
    assert 2 * 64 * (8 * 16000 + 9 * 32000 + 8 * 1893) <= n < 4 * 64 * (9 * 16000 + 10 * 32000 + 9 * 1893) + 64 * 1000000


def test_product_fails_loudly_without_gpu():
    from gcnn_cut_selector_amd.model import GCNN
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.GcnnError):
        GCNN()


def test_sample_roundtrip_and_load_batch(tmp_path):
    samples = [synthetic.make_sample("combauc", i) for i in range(3)]
    files = []
    for i, (state, imp) in enumerate(samples):
        f = str(tmp_path / f"sample_{i}.pkl")
        utils.save_sample(f, state, imp)
        files.append(f)
    got = utils.load_batch(files)
    want = synthetic.stack_samples(samples)
    assert len(got) == 11
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    # edge indices are shifted per sample (utils.py:401-407): disjoint union
    assert got[1][0].max() == got[7].sum() - 1 or got[1][0].max() < got[7].sum()
    single = utils.state_to_inputs(samples[0][0])
    assert single[7:] == (samples[0][0][0]["values"].shape[0], samples[0][0][2]["values"].shape[0], samples[0][0][3]["values"].shape[0])


def test_ranking_fraction_matches_reference_semantics():
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = int(rng.integers(1, 30))
        pred, true = rng.integers(0, 5, n).astype(float), rng.integers(0, 5, n).astype(float)  # many ties
        assert ranking_fraction(pred, true) == O.ranking_fraction(pred, true)


def test_shard_samples_balances_edges_and_is_deterministic():
    sizes = [30825, 27007, 41000, 1000, 999, 35000, 28000, 30000]
    shards = parallel.shard_samples(sizes, 4)
    assert sorted(i for s in shards for i in s) == list(range(8))
    loads = [sum(sizes[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(sizes)
    assert shards == parallel.shard_samples(sizes, 4)
    assert parallel.shard_samples([5], 2) == [[0], []]


def _dp_worker(rank, world, port, tmpdir):
    """One data-parallel rank on CPU (gloo): local SUM-loss gradients from the oracle, ONE all-reduce of the packed buffer."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params = O.randomize_params(O.init_params(3, np.float64), 4)
    samples = [synthetic.make_sample("setcov", i, scale=0.05) for i in range(5)]
    sizes = [s[0][1]["indices"].shape[1] + s[0][4]["indices"].shape[1] for s in samples]
    mine = parallel.shard_samples(sizes, world)[rank]
    names = [n for n, _, t in O.PARAM_SPEC if t]
    flat = np.zeros(sum(int(np.prod(s)) for _, s, t in O.PARAM_SPEC if t))
    n_cuts = 0
    if mine:
        b = synthetic.stack_samples([samples[i] for i in mine])
        state = b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum()))
        n_cuts = int(b[9].sum())
        _, mean_loss, grads = O.loss_and_grads(params, state, b[10], torch.float64)
        flat = np.concatenate([grads[n].reshape(-1) for n in names]) * n_cuts   # gradient of the local SUM
    buf = torch.from_numpy(parallel.pack(flat, n_cuts).astype(np.float64))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    np.save(os.path.join(tmpdir, f"dp_{rank}.npy"), parallel.unpack_mean(buf.numpy()))
    dist.destroy_process_group()


def test_data_parallel_gradient_equals_global_batch_gradient(tmp_path):
    """world_size 2 over gloo: the reduced, count-scaled gradient equals the gradient of the mean loss over ALL cuts of the
    global batch (ranks hold different cut counts, so a mean of per-rank means would fail this)."""
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    params = O.randomize_params(O.init_params(3, np.float64), 4)
    samples = [synthetic.make_sample("setcov", i, scale=0.05) for i in range(5)]
    b = synthetic.stack_samples(samples)
    state = b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum()))
    _, _, grads = O.loss_and_grads(params, state, b[10], torch.float64)
    names = [n for n, _, t in O.PARAM_SPEC if t]
    want = np.concatenate([grads[n].reshape(-1) for n in names])
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"dp_{r}.npy"))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6 * np.abs(want).max())  # the packed buffer is fp32


def test_process_binds_the_reference_positional_order(monkeypatch):
    """model_trainer.py:156,161,182 call `process(model, data, fractions, loss_fn[, optimizer])` positionally: the fourth slot
    is the loss function, the fifth the optimizer.  Checked on the CPU with the device calls stubbed out."""
    from gcnn_cut_selector_amd import trainer

    class FakeModel:
        device = torch.device("cpu")
        flat_parameters = torch.zeros(8)

        def prepare(self, inputs):
            return inputs

        def __call__(self, batch, training):
            return torch.zeros(3)

    seen = []

    def fake_train_step(model, batch, y, optimizer, state, process_group=None):
        seen.append(("train", optimizer, process_group))
        return torch.ones(1), torch.zeros(3)

    monkeypatch.setattr(trainer, "train_step", fake_train_step)
    monkeypatch.setattr(trainer, "mse_loss", lambda pred, y, scale=None, want_grad=True: (seen.append(("eval", None, None)) or torch.ones(1), None))
    monkeypatch.setattr(trainer, "ranking_metric", lambda *a, **k: None)
    z = np.zeros
    batch = (z((1, 4), np.float32), z((2, 0), np.int32), z((0, 1), np.float32), z((1, 14), np.float32), z((3, 6), np.float32),
             z((2, 0), np.int32), z((0, 1), np.float32), np.array([1]), np.array([1]), np.array([3]), z(3, np.float32))
    loss_fn = lambda y_true, y_pred: None       # stands for tf.keras.losses.MeanSquaredError() (model_trainer.py:132)
    opt = trainer.Adam(learning_rate=lambda: 1e-3)
    fractions = np.array([0.25, 0.5, 0.75, 1])
    trainer.process(FakeModel(), [batch], fractions, loss_fn, opt)          # model_trainer.py:156
    trainer.process(FakeModel(), [batch], fractions, loss_fn)               # model_trainer.py:161,182
    assert seen == [("train", opt, None), ("eval", None, None)]
    with pytest.raises(TypeError):                                           # round-1 call order: optimizer in the loss_fn slot
        trainer.process(FakeModel(), [batch], fractions, opt)
    with pytest.raises(TypeError):
        trainer.process(FakeModel(), [batch], fractions, loss_fn, opt, "group")   # process_group is keyword only


def test_checkpoint_and_sample_files_are_unpickled_restrictively(tmp_path):
    """restore_state / load_sample read files from disk: only NumPy arrays in builtin containers may come out of them."""
    import gzip
    import io
    import pickle
    from gcnn_cut_selector_amd import _safe_pickle
    buf = io.BytesIO()
    arrays = [np.arange(12, dtype=np.float32).reshape(3, 4), np.float32(2.5) * np.ones(1, np.float32)]
    for a in arrays:                                        # the checkpoint format: consecutive bare records (model.py:53-56)
        pickle.dump(a, buf)
    buf.seek(0)
    for a in arrays:
        got = _safe_pickle.load(buf)
        assert got.dtype == a.dtype and np.array_equal(got, a)

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))

    with pytest.raises(pickle.UnpicklingError):
        _safe_pickle.load(io.BytesIO(pickle.dumps(Evil())))
    bad = str(tmp_path / "sample_evil.pkl")
    with gzip.open(bad, "wb") as f:
        pickle.dump({"data": [Evil(), np.zeros(1)]}, f)
    with pytest.raises(pickle.UnpicklingError):
        utils.load_sample(bad)


def _prenorm_worker(rank, world, port, tmpdir):
    """One rank of a data-parallel PreNorm fit on CPU (gloo): streaming statistics over ITS batches, then the rank-ordered merge."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    batches = [rng.standard_normal((int(n), 4)).astype(np.float32) * 3 + 1 for n in (50, 7, 0, 120, 33)]
    mine = batches[rank::world] if rank == 0 else []          # rank 1 holds nothing at all in the second case below
    for case, shard in (("split", batches[rank::world]), ("one-sided", mine if rank == 0 else [])):
        if case == "one-sided" and rank == 0:
            shard = batches
        local = parallel.chan_merge((len(b), b.mean(0) if len(b) else np.zeros(4), b.var(0) if len(b) else np.zeros(4)) for b in shard)
        count, mean, var = local if local[1] is not None else (np.float32(0), np.zeros(4, np.float32), np.zeros(4, np.float32))
        got = parallel.allgather_prenorm(count, mean, var, len(shard) > 0, dist.group.WORLD, torch.device("cpu"))
        np.savez(os.path.join(tmpdir, f"pn_{case}_{rank}.npz"), count=got[0], mean=got[1], var=got[2], received=got[3])
    dist.destroy_process_group()


def test_data_parallel_prenorm_fit_matches_single_process(tmp_path):
    """world_size 2 over gloo: merged statistics == the two-pass population statistics of ALL batches (SURVEY section 4
    invariant 6), identical on both ranks, also when one rank holds no data."""
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(_prenorm_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(5)
    allx = np.concatenate([rng.standard_normal((int(n), 4)).astype(np.float32) * 3 + 1 for n in (50, 7, 0, 120, 33)])
    for case in ("split", "one-sided"):
        r0, r1 = (np.load(os.path.join(str(tmp_path), f"pn_{case}_{r}.npz")) for r in range(2))
        for k in ("count", "mean", "var", "received"):
            assert np.array_equal(r0[k], r1[k]), (case, k)      # bit-identical on every rank
        assert float(r0["count"]) == len(allx) and bool(r0["received"])
        np.testing.assert_allclose(r0["mean"], allx.mean(0), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r0["var"], allx.var(0), rtol=1e-4)


def test_decode_files_children_are_torch_free_and_match_load_sample(tmp_path):
    """`utils.decode_files`: gunzip in child interpreters that import the standard library only (no torch, no re-import of the
    caller's __main__), results equal to `load_sample` file by file, errors raised in the parent, children gone afterwards."""
    import subprocess
    import sys
    samples = [synthetic.make_sample("combauc", i) for i in range(7)]
    files = []
    for i, (state, imp) in enumerate(samples):
        files.append(str(tmp_path / f"sample_{i}.pkl"))
        utils.save_sample(files[-1], state, imp)
    want = [utils.load_sample(f) for f in files]
    for workers in (0, 1, 3, 16):
        got = list(utils.decode_files(files, workers))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            for dg, dw in zip(g[0], w[0]):
                assert dg["features"] == dw["features"] and all(np.array_equal(dg[k], dw[k]) for k in dw if k != "features")
            assert np.array_equal(g[1], w[1])
    with pytest.raises(OSError):
        list(utils.decode_files(files[:2] + [str(tmp_path / "missing.pkl")] + files[2:], 2))
    worker = os.path.join(ROOT, "gcnn-cut-selector_amd", "_decode_worker.py")
    probe = ("import sys, runpy; sys.argv = ['w']; sys.stdin = open('/dev/null'); runpy.run_path(%r, run_name='__main__'); "
             "print(int(any(m == 'torch' or m == 'numpy' or m.startswith('gcnn') for m in sys.modules)))" % worker)
    out = subprocess.run([sys.executable, "-S", "-E", "-c", probe], capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "0"


def _gather_worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    rows = [5, 0][rank] if world == 2 else rank
    a = torch.from_numpy(rng.standard_normal((rows, 14)).astype(np.float32))
    b = torch.from_numpy(rng.integers(0, 1000, 3 + 2 * rank).astype(np.int32))
    e = torch.zeros(0, 4)                                       # nobody holds a row
    out = [parallel.allgather_concat(t, dist.group.WORLD) for t in (a, b, e)]
    np.savez(os.path.join(tmpdir, f"ag_{rank}.npz"), a=out[0].numpy(), b=out[1].numpy(), e=out[2].numpy(), mine_a=a.numpy(), mine_b=b.numpy())
    dist.destroy_process_group()


def test_allgather_concat_is_a_rank_ordered_ragged_concatenation(tmp_path):
    """The exchange step of a sharded `SampleStore.from_files` (world_size 2 over gloo): ranks hold different row counts (one of
    them none); every rank gets the rank-ordered concatenation, dtype and trailing shape kept."""
    import torch.multiprocessing as mp
    port = 33500 + os.getpid() % 2000
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), f"ag_{r}.npz")) for r in range(2))
    for k in ("a", "b"):
        want = np.concatenate([r0["mine_" + k], r1["mine_" + k]])
        assert r0[k].dtype == want.dtype and np.array_equal(r0[k], want) and np.array_equal(r1[k], want)
    assert r0["e"].shape == (0, 4) and r1["e"].shape == (0, 4)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` without WORLD_SIZE must start two ranks (fresh children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), relay rank 0's line and fail when a rank fails or the line does not show 2 ranks on 2 devices.  Run here with
    the stub worker (gloo on CPU, no GPU work); the launcher code is the one a GPU run uses."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = lambda extra, **kw: subprocess.run([sys.executable, bench, "--gpus", "2", "--stub-worker"], capture_output=True, text=True,
                                             env=dict(env, **extra), timeout=300, **kw)
    ok = run({})
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [ln for ln in ok.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                                  # ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["world_size_observed"] == 2
    assert sorted(r["rank"] for r in out["distributed"]["ranks"]) == [0, 1]
    assert len({r["uuid"] for r in out["distributed"]["ranks"]}) == 2
    failed = run({"GCNN_BENCH_STUB_FAIL": "1"})                             # a rank dies: non-zero, no line
    assert failed.returncode != 0 and not failed.stdout.strip()
    shared = run({"GCNN_BENCH_STUB_SAME_DEVICE": "1"})                      # two ranks on one device: refused
    assert shared.returncode != 0 and not shared.stdout.strip()
    wrong = run({"WORLD_SIZE": "1", "RANK": "0"})                           # ranks exist but not as many as --gpus says
    assert wrong.returncode != 0 and not wrong.stdout.strip()


def test_out_of_memory_skips_a_batch_single_gpu_and_propagates_data_parallel(monkeypatch, capsys):
    """The reference skips a batch that exhausts memory (model_trainer.py:308-311: `except ResourceExhaustedError: print(WARNING)`).
    Single GPU: the same.  Data parallel: the peers are already in this step's all-reduce, so a rank must not skip on its own --
    the error propagates (documented deviation).  Checked on the CPU with the device calls stubbed out."""
    from gcnn_cut_selector_amd import trainer

    class FakeModel:
        device = torch.device("cpu")
        flat_parameters = torch.zeros(8)

        def prepare(self, inputs):
            return inputs

    calls = []

    def oom_on_second(model, batch, y, optimizer, state, process_group=None):
        calls.append(process_group)
        if len(calls) == 2:
            raise torch.OutOfMemoryError("HIP out of memory (simulated)")
        return torch.ones(1), torch.zeros(3)

    monkeypatch.setattr(trainer, "train_step", oom_on_second)
    monkeypatch.setattr(trainer, "ranking_metric", lambda *a, **k: None)
    z = np.zeros
    batch = (z((1, 4), np.float32), z((2, 0), np.int32), z((0, 1), np.float32), z((1, 14), np.float32), z((3, 6), np.float32),
             z((2, 0), np.int32), z((0, 1), np.float32), np.array([1]), np.array([1]), np.array([3]), z(3, np.float32))
    opt = trainer.Adam(learning_rate=lambda: 1e-3)
    fractions = np.array([0.25, 0.5, 0.75, 1])
    loss, acc = trainer.process(FakeModel(), [batch, batch, batch], fractions, None, opt)      # the second batch is skipped
    assert len(calls) == 3 and "WARNING: batch skipped." in capsys.readouterr().out
    calls.clear()
    monkeypatch.setattr(torch.distributed, "all_reduce", lambda *a, **k: None)
    with pytest.raises(torch.OutOfMemoryError):
        trainer.process(FakeModel(), [batch, batch, batch], fractions, None, opt, process_group="group")
    assert calls == ["group", "group"]                                                            # stopped at the failing step


def test_host_pack_edges_copies_sorted_lists_and_sorts_the_others_stably():
    """gcnn_host_pack_edges (the host half of gcnn_infer: no device work): a (row, col)-sorted list as get_state emits it
    (utils.py:102-104) is copied, any other order comes out as NumPy's stable sort by row, an unsorted list with a row id out of
    range is copied as it is (the device check of gcnn_infer reports it)."""
    import ctypes as C
    from gcnn_cut_selector_amd import _lib
    f = _lib.lib().gcnn_host_pack_edges
    rng = np.random.default_rng(5)
    for n_left, n in ((1, 1), (7, 40), (300, 5000), (50, 0)):
        rows = np.sort(rng.integers(0, n_left, n)).astype(np.int32)
        cols = rng.integers(0, 1000, n).astype(np.int32)
        vals = rng.standard_normal(n).astype(np.float32)
        scratch = np.empty(n_left + 1, np.int32)
        for shuffled in (False, True):
            if shuffled:
                p = rng.permutation(n)
                rows, cols, vals = rows[p].copy(), cols[p].copy(), vals[p].copy()
            out_i, out_v = np.full(2 * n, -7, np.int32), np.full(n, np.nan, np.float32)
            rc = f(rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, n, n_left, out_i.ctypes.data, out_v.ctypes.data, scratch.ctypes.data)
            order = np.argsort(rows, kind="stable")
            was_sorted = bool((np.diff(rows) >= 0).all())
            assert rc == (0 if was_sorted else 1)
            assert np.array_equal(out_i[:n], rows[order]) and np.array_equal(out_i[n:], cols[order]) and np.array_equal(out_v, vals[order])
    rows, cols, vals = np.array([3, 9, 1], np.int32), np.array([0, 1, 2], np.int32), np.array([1, 2, 3], np.float32)
    out_i, out_v, scratch = np.zeros(6, np.int32), np.zeros(3, np.float32), np.empty(6, np.int32)
    assert f(rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, 3, 5, out_i.ctypes.data, out_v.ctypes.data, scratch.ctypes.data) == 2
    assert np.array_equal(out_i, [3, 9, 1, 0, 1, 2]) and np.array_equal(out_v, vals)
    assert f(None, None, None, 3, 5, None, None, None) == -1
