"""End-to-end training throughput over an epoch, three ways of feeding the same batches (setcov-500, batch 32):
  files : utils.load_batch per batch (gunzip + unpickle + NumPy stacking, as model_trainer.py:150-153 does) -> prepare
  host  : samples decoded once and kept in host memory; utils.collate + prepare per batch
  store : SampleStore (samples resident in HBM, one collation kernel per batch)
Usage: python tools/epoch_throughput.py [n_samples=64] [n_batches=40]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcnn_cut_selector_amd import synthetic, utils  # noqa: E402
from gcnn_cut_selector_amd.model import GCNN  # noqa: E402
from gcnn_cut_selector_amd.store import SampleStore  # noqa: E402
from gcnn_cut_selector_amd.trainer import Adam, process  # noqa: E402


def main():
    n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp()
    files = []
    samples = []
    for i in range(n_samples):
        s = synthetic.make_sample("setcov", i)
        samples.append(s)
        files.append(os.path.join(tmp, f"sample_{i}.pkl"))
        utils.save_sample(files[-1], *s)
    rng = np.random.default_rng(0)
    ids = rng.choice(n_samples, n_batches * 32, replace=True)
    fractions = np.array([0.25, 0.5, 0.75, 1.0])
    t = time.perf_counter()
    store = SampleStore.from_files(files, dev)
    torch.cuda.synchronize()
    print(f"store ingest: {n_samples} files in {time.perf_counter() - t:.2f} s, {store.nbytes / 2**20:.1f} MiB in HBM "
          f"({store.nbytes / n_samples / 1024:.0f} KiB/sample)")


    def feed(kind):
        for j in range(0, len(ids), 32):
            idx = ids[j:j + 32]
            if kind == "files":
                yield utils.load_batch([files[i] for i in idx])
            elif kind == "host":
                yield utils.collate([samples[i] for i in idx])
            else:
                yield store.batch(idx)


    edges = None
    for kind in ("files", "host", "store"):
        m = GCNN(device=dev)
        opt = Adam(learning_rate=lambda: 1e-3)
        process(m, list(feed(kind))[:2], fractions, None, opt)      # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        loss, acc = process(m, feed(kind), fractions, None, opt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{kind:6s}: {n_batches} batches in {dt * 1e3:8.1f} ms = {dt / n_batches * 1e3:7.3f} ms/batch, "
              f"{n_batches * 32 / dt:9.0f} samples/s, loss {loss:.6f}")
    # collation alone
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(0, len(ids), 32):
        store.batch(ids[j:j + 32])
    e1.record()
    torch.cuda.synchronize()
    print(f"store.batch alone: {e0.elapsed_time(e1) / n_batches * 1e3:.1f} us/batch (device timeline)")


if __name__ == "__main__":   # SampleStore.from_files spawns decoder processes that re-import this module
    main()
