"""Data parallelism on the GPU with two real ranks: each rank takes its shard of the samples
(`parallel.shard_samples`), back-propagates the local SUM of squared errors, the flat [gradients | cut count] buffer is
all-reduced once, and gradient / count must equal the single-process gradient of the mean loss over ALL cuts of the global
batch (model_trainer.py:271) -- also when the ranks hold different numbers of cuts.  Both ranks share the one GPU of the
test box, so the collective runs over gloo (RCCL refuses two ranks on one device); the code path in `train_step` is the same."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gcnn_cut_selector_amd import synthetic  # noqa: E402
from oracle import gcnn_oracle as O  # noqa: E402  (checker only)

N_SAMPLES = 7


def _samples():
    return [synthetic.make_sample("combauc", i) for i in range(4)] + [synthetic.make_sample("setcov", i, scale=0.1) for i in range(3)]


def _weights():
    return O.randomize_params(O.init_params(11, np.float32), 12)


def _store_arrays(store):
    out = {"cons_feats": store.cons_feats, "var_feats": store.var_feats, "cut_feats": store.cut_feats, "improvements": store.improvements}
    for slot, g in enumerate(store.graphs):
        out.update({f"{slot}.{k}": v for k, v in g.items()})
    out = {k: v.cpu().numpy() for k, v in out.items()}
    out["sizes"], out["offsets"] = store.sizes, store.offsets
    for slot in range(2):
        for side in range(2):
            out[f"max_deg{slot}{side}"] = np.asarray(store.max_deg[slot][side])
    return out


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.parallel import shard_samples
    from gcnn_cut_selector_amd.store import SampleStore
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    samples = _samples()
    params = _weights()
    m = GCNN(device=dev)
    m.set_weights([params[n] for n in O.PARAM_NAMES])
    sizes = [s[0][1]["indices"].shape[1] + s[0][4]["indices"].shape[1] for s in samples]
    mine = shard_samples(sizes, world)[rank]
    store = SampleStore.from_samples(samples, dev)
    sb = store.batch(mine)
    ts = TrainState(m)
    train_step(m, sb.batch, sb.improvements, None, ts, process_group=dist.group.WORLD)
    torch.cuda.synchronize()
    # the reference-style epoch loop, data parallel: two training batches (the second one short), then an evaluation pass
    # with single-sample batches, where one of the two ranks always holds an empty shard
    from gcnn_cut_selector_amd.trainer import Adam, process
    fractions = np.array([0.25, 0.5, 0.75, 1.0])
    opt = Adam(learning_rate=lambda: 1e-3)
    ids = np.arange(N_SAMPLES)
    train = process(m, store.batches(ids, 4, rank, world), fractions, None, opt, process_group=dist.group.WORLD)
    valid = process(m, store.batches(ids[:3], 1, rank, world), fractions, process_group=dist.group.WORLD)
    torch.cuda.synchronize()
    # data-parallel PreNorm fitting: every rank fits on ITS shard of the pretraining batches (rank 1 gets a single one), the
    # per-layer statistics are merged across ranks (GCNN.pretrain_sync)
    from gcnn_cut_selector_amd.trainer import pretrain
    mp_ = GCNN(device=dev, seed=3)
    my_batches = list(store.batches(ids, 2))[0:3] if rank == 0 else list(store.batches(ids, 2))[3:]
    n_layers = pretrain(mp_, my_batches, process_group=dist.group.WORLD)
    # sharded ingestion: each rank decodes ITS share of the sample files (rank 0: files 0-2, rank 1: files 3-6), one exchange, and
    # every rank must hold the store a single process builds from all files, array for array
    from gcnn_cut_selector_amd import utils
    files = []
    for i, (state, imp) in enumerate(samples):
        files.append(os.path.join(os.path.dirname(out_path), f"sample_{i}.pkl"))
        if rank == 0:
            utils.save_sample(files[-1], state, imp)
    dist.barrier()
    sharded = _store_arrays(SampleStore.from_files(files, dev, chunk=2, workers=2, process_group=dist.group.WORLD))
    whole = _store_arrays(store)
    assert sharded.keys() == whole.keys()
    for k in whole:
        assert sharded[k].dtype == whole[k].dtype and np.array_equal(sharded[k], whole[k]), (rank, k)
    if rank == 0:
        np.savez(out_path, buf=ts.buf.cpu().numpy(), mine=np.asarray(mine), train_loss=train[0], train_acc=train[1],
                 valid_loss=valid[0], valid_acc=valid[1], weights=m.flat_parameters.detach().cpu().numpy(),
                 prenorm_layers=n_layers, prenorm_weights=mp_.flat_parameters.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_and_epoch_loop_equal_single_process(tmp_path):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import torch.multiprocessing as mp
    from gcnn_cut_selector_amd.model import GCNN
    from gcnn_cut_selector_amd.store import SampleStore
    from gcnn_cut_selector_amd.trainer import TrainState, train_step
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    # single process, whole batch, mean loss
    dev = torch.device("cuda", 0)
    samples, params = _samples(), _weights()
    m = GCNN(device=dev)
    m.set_weights([params[n] for n in O.PARAM_NAMES])
    sb = SampleStore.from_samples(samples, dev).batch(np.arange(N_SAMPLES))
    ts = TrainState(m)
    train_step(m, sb.batch, sb.improvements, None, ts)
    want = ts.grads.cpu().numpy()
    n = want.size
    count = got["buf"][n]
    assert count == sb.batch.dims.n_cuts                      # the reduced count slot = cuts of the GLOBAL batch
    assert 0 < len(got["mine"]) < N_SAMPLES                    # rank 0 really held only a shard
    np.testing.assert_allclose(got["buf"][:n] / count, want, rtol=2e-4, atol=2e-6 * np.abs(want).max())
    # the same epoch loop in one process on the global batches
    from gcnn_cut_selector_amd.trainer import Adam, process
    fractions = np.array([0.25, 0.5, 0.75, 1.0])
    store = SampleStore.from_samples(samples, dev)
    opt = Adam(learning_rate=lambda: 1e-3)
    ids = np.arange(N_SAMPLES)
    train = process(m, store.batches(ids, 4), fractions, None, opt)
    valid = process(m, store.batches(ids[:3], 1), fractions)
    np.testing.assert_allclose(got["train_loss"], train[0], rtol=1e-4)
    np.testing.assert_allclose(got["valid_loss"], valid[0], rtol=1e-4)
    np.testing.assert_array_equal(got["train_acc"], train[1])
    np.testing.assert_array_equal(got["valid_acc"], valid[1])
    w = m.flat_parameters.detach().cpu().numpy()
    np.testing.assert_allclose(got["weights"], w, rtol=1e-3, atol=1e-5)       # two Adam steps on (almost) equal gradients
    # PreNorm fitting: two ranks on disjoint shards == one process on all batches (the Chan merge is associative up to fp32 rounding)
    from gcnn_cut_selector_amd.trainer import pretrain
    m1 = GCNN(device=dev, seed=3)
    assert pretrain(m1, list(store.batches(ids, 2))) == 11 == int(got["prenorm_layers"])
    np.testing.assert_allclose(got["prenorm_weights"], m1.flat_parameters.detach().cpu().numpy(), rtol=2e-4, atol=1e-6)
