"""Host-side profile (cProfile) of the store-fed epoch loop: where the Python time of one batch goes.
Usage: python tools/profile_host.py [n_batches=400]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcnn_cut_selector_amd import synthetic  # noqa: E402
from gcnn_cut_selector_amd.model import GCNN  # noqa: E402
from gcnn_cut_selector_amd.store import SampleStore  # noqa: E402
from gcnn_cut_selector_amd.trainer import Adam, process  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
samples = [synthetic.make_sample("setcov", i) for i in range(64)]
store = SampleStore.from_samples(samples, dev)
rng = np.random.default_rng(0)
ids = rng.choice(64, n_batches * 32, replace=True)
fractions = np.array([0.25, 0.5, 0.75, 1.0])
m = GCNN(device=dev)
opt = Adam(learning_rate=lambda: 1e-3)
process(m, store.batches(ids[:64], 32), fractions, None, opt)
torch.cuda.synchronize()
t = time.perf_counter()
for j in range(0, len(ids), 32):
    store.batch(ids[j:j + 32])
dt = time.perf_counter() - t
torch.cuda.synchronize()
print(f"store.batch host time: {dt / n_batches * 1e6:.1f} us/batch")
t = time.perf_counter()
process(m, store.batches(ids, 32), fractions, None, opt)
torch.cuda.synchronize()
print(f"epoch: {(time.perf_counter() - t) / n_batches * 1e3:.3f} ms/batch")
pr = cProfile.Profile()
pr.enable()
process(m, store.batches(ids, 32), fractions, None, opt)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
